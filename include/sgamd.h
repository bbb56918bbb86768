/* sgamd.h -- C ABI of libsgamd.so: the MI355X (gfx950) native kernels behind StudioGAN's GAN training step
 * and FID/IS feature-extraction hot path.
 *
 * Every entry point takes plain pointers/sizes (device pointers are what torch's tensor.data_ptr() returns),
 * enqueues work on the given hipStream_t (pass torch.cuda.current_stream().cuda_stream), never synchronises
 * the host, never allocates, and returns 0 on success or a negative code (message via sg_last_error()).
 * This is the same contract as the reference's only native plugin precedent
 * (reference src/utils/custom_ops.py:59-155, src/utils/style_ops/bias_act.cpp:35-97: validate -> RuntimeError,
 * borrowed dense inputs, caller-owned outputs, current stream, no host sync).
 *
 * Activations are NHWC ("pixel-major": [N][H][W][C], C contiguous) in fp32 or bf16; the Python host mirror
 * (pytorch-studiogan_amd/ops.py) converts at the reference's NCHW boundary. Master weights stay in the
 * reference's OIHW fp32 layout (state_dict compatible, reference src/utils/ckpt.py:38); the spectral-norm
 * kernels emit the [Cout][R][S][Cin] (forward) and [Cin][R'][S'][Cout] (data-gradient) operand images.
 *
 * Reference call sites replaced (all paths relative to /root/reference/src):
 *   sg_conv2d_fwd / sg_conv2d_wgrad   utils/ops.py:165-173,195-204 (nn.Conv2d fwd + autograd dgrad/wgrad),
 *                                     utils/ops.py:176-184 (ConvTranspose2d, via SG_PIX_TRANSPOSED)
 *   sg_gemm                           utils/ops.py:187-188,219-220 (nn.Linear), utils/ops.py:93,100 (torch.bmm)
 *   sg_sn_*                           utils/ops.py:195-224 (torch.nn.utils.spectral_norm hook)
 *   sg_bn_* / sg_cbn_*                utils/ops.py:14-28,227-228 (BatchNorm2d eps=1e-4, ConditionalBatchNorm2d)
 *   sg_softmax_* / sg_maxpool2_*      utils/ops.py:79-94 (SelfAttention)
 *   sg_avgpool2_*                     models/big_resnet.py:175,219 (nn.AvgPool2d(2))
 *   sg_relu_sum_hw_* / sg_pd_head_*   models/big_resnet.py:359-363,386-387 (D head, projection)
 *   sg_loss_*                         utils/losses.py:197-239 (vanilla / hinge / wasserstein)
 *   sg_embedding_*                    utils/ops.py:191-192
 *   sg_adam_ema / sg_ema_lerp         config.py:541-563 (torch.optim.Adam eps 1e-6), utils/ema.py:27-40
 *   sg_quantize_resize_normalize      utils/ops.py:251-263, utils/resize.py:72-93
 */
#ifndef SGAMD_H
#define SGAMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* sg_stream_t; /* hipStream_t */

#define SG_DTYPE_F32 0
#define SG_DTYPE_BF16 1
#define SG_DTYPE_F64 2 /* collectives only */

/* activation-view flags (pix_flags / x_flags / g_flags) */
#define SG_PIX_RELU 1
#define SG_PIX_UPSAMPLE 2
#define SG_PIX_QUAD 4
#define SG_PIX_TRANSPOSED 8
/* epilogue flags */
#define SG_EPI_OUT_F32 1
#define SG_EPI_ATOMIC 2
#define SG_EPI_POOL 4
#define SG_EPI_RELU 8
#define SG_EPI_RES_F32 16

const char* sg_last_error(void);
int sg_version(void);
/* launch profiler used by bench.py's roofline leg: hipEvent pairs around every contraction-engine launch, recorded on the
 * launch stream. collect: out[kind*3+{0,1,2}] = {launches, total ms, algorithmic FLOPs}; kind 0 = conv fwd/dgrad,
 * 1 = conv wgrad, 2 = gemm. Synchronise the device before collecting.
 * `on` is a bit set: 1 = the contraction engine (kinds 0-2), 2 = the HBM-bound families (kinds 3-6: spectral norm, batch norm, attention, Adam / EMA),
 * 4 = sg_conv2d_q's launches alone (bench.py's timed region: the dominant kernel, without the dispatch gaps event pairs around every launch would cost). */
int sg_prof_enable(int on);
int sg_prof_collect(double* out, int nkinds);

/* out[n,ho,wo,co] = beta*res + mask(alpha * pool2x2sum(conv(x', w)) + bias), x' = [relu][upsample x2](x).
 * Forward convolution and (with the data-gradient weight image) its data gradient. */
typedef struct {
  int dtype;
  int N, Hs, Ws, C, ldx;          /* stored input [N,Hs,Ws,C], element pitch between pixels */
  int Ho, Wo, Cout;               /* output grid (before the optional 2x2 pooling) and channels */
  int R, S, stride, pad_h, pad_w;
  int pix_flags, epi_flags;
  float alpha, beta;
  const void* x; const void* w;   /* w: [Cout][R*S*C], same dtype as x */
  const float* bias; const void* res; const void* mask; void* out; const float* alpha_ptr;
  int ldo, ldr, ldm;
} sg_conv_fwd_desc;
int sg_conv2d_fwd(const sg_conv_fwd_desc* d, sg_stream_t stream);
/* fp32 arithmetic of sg_conv2d_fwd's generic engine (process-wide; default 0): 0 = v_mfma_f32_32x32x2_f32, the exact fp32 FMA chain; 3 = "bf16x3": each fp32
 * operand element is split into two bf16 terms in registers and a 16-wide k-tile runs as three bf16 MFMAs with fp32 accumulation (~2^-16 relative per
 * product, 5.3x the matrix-pipe rate). fp32 tensors in and out either way. Used by the FID / IS feature extractor (metrics.InceptionV3 f32_mode). */
int sg_set_f32_mode(int mode);
int sg_get_f32_mode(void);

/* Residual-block tail in ONE launch (bf16): out = epilogue( alpha * [ conv3x3(x; w) + conv1x1(up2?(x2); w2) ] + bias + bias2 ).
 * Replaces `x0 = conv2d0(x0); out = x + x0` of the reference's blocks (src/models/big_resnet.py:28-42 GenBlock with nearest x2 on the
 * skip input, :221-242 DiscBlock where main and skip are both average-pooled: pooling is linear, so the sum is pooled once).
 * main: a 3x3 / stride 1 / pad 1 problem WITHOUT SG_PIX_UPSAMPLE; SG_PIX_RELU applies to x and x2 alike. x2: [N, Ho(/2), Wo(/2), ldx2] NHWC
 * with C2 % 32 == 0 channels, w2: [Cout][C2]. sg_conv2d_fwd_skip_ok() == 1 when the fused kernel takes the problem (else run two launches). */
typedef struct {
  sg_conv_fwd_desc main;
  const void* x2; const void* w2; const float* bias2;
  int C2, ldx2, x2_up;
  float* stats;   /* optional: [sg_conv2d_fwd_skip_stat_rows()][Cout][2] floats, per 256-pixel tile the sum and the sum of squares of the (bf16) result
                     per channel -- the statistics of the batch norm behind this convolution, taken in the epilogue (no extra pass over the
                     activation); reduce with sg_bn_stats_from_tiles. Ignored with SG_EPI_POOL. */
} sg_conv_skip_desc;
int sg_conv2d_fwd_skip_stat_rows(const sg_conv_skip_desc* d);
int sg_conv2d_fwd_skip(const sg_conv_skip_desc* d, sg_stream_t stream);
int sg_conv2d_fwd_skip_ok(const sg_conv_skip_desc* d);
/* Number of sg_conv2d_fwd problems this process has run on the row-streaming kernel (csrc/conv_rs.h: 3x3, <= 32 output channels, 128-pixel-wide
 * bf16 images -- the RGB layers of the 128 x 128 configurations). Introspection for tests and benchmarks: which engine took a shape. */
long long sg_conv_rs_launches(void);

/* ---- "quad" convolutions: a 3x3 / pad-1 convolution next to a 2x resampling, through the exact filter identity (csrc/conv_q.h) -------
 * SG_Q_POOL: out[N,Hl,Wl,Cout] = epilogue( alpha * avgpool2(conv3x3(relu?(x); w)) + bias ),  x: [N,2Hl,2Wl,C]
 *            = conv4x4 / stride 2 / pad 1 with the pooled filter: replaces `conv2d2` + `average_pooling` of the reference's discriminator
 *            blocks (src/models/big_resnet.py:177-192,221-242) and the data gradient of SG_Q_UP.
 * SG_Q_UP:   out[N,2Hl,2Wl,Cout] = epilogue( alpha * conv3x3(nearest_up2(relu?(x)); w) + bias ),  x: [N,Hl,Wl,C]
 *            = four 2x2 convolutions, one per output parity: replaces `F.interpolate(scale_factor=2)` + `conv2d1` of the generator blocks
 *            (big_resnet.py:28-42) and the data gradient of SG_Q_POOL. mask / res / out are tensors of the FINE grid.
 * 16 C MACs per low-resolution position and output channel instead of 36 C. wq: the quad filter image [Cout][4 views][4 taps][C] written by
 * sg_quad_pack from the 3x3 image (mode 0 / 1: forward images of POOL / UP from [Cout][3][3][C]; mode 2 / 3: the data-gradient images of
 * POOL / UP -- to be run with form UP / POOL -- from the flipped transposed image [Cin][2-r][2-s][Cout] the 3x3 data gradient uses).
 * bf16, C % 32 == 0, Cout % 64 == 0 or % 96 == 0, Hl / Wl powers of two, Wl >= 4. Epilogue order as sg_conv2d_fwd: scale, bias, mask, residual, ReLU. */
#define SG_Q_POOL 0
#define SG_Q_UP 1
typedef struct {
  int dtype, form;
  int N, Hl, Wl;                  /* the LOW-resolution grid */
  int C, ldx, Cout;
  int pix_flags, epi_flags;       /* SG_PIX_RELU; SG_EPI_RELU */
  float alpha, beta;
  const void* x; const void* wq;
  const float* bias; const void* res; const void* mask; void* out; const float* alpha_ptr;
  int ldo, ldr, ldm;
  /* optional (SG_Q_POOL): the block's 1x1 skip convolution in the same launch, out += avgpool2(conv1x1(relu?(x2); w2)) + bias2.
   * x2: [N,2Hl,2Wl,ldx2] with C2 % 32 == 0 channels; w2q: [Cout][C2] = the skip filter x 1/4 (sg_quad_pack_batch mode 4). NULL = none. */
  const void* x2; const void* w2q; const float* bias2;
  int C2, ldx2;
  float* stats;   /* optional: per-tile batch-norm statistics of the result, [sg_conv2d_q_stat_rows()][Cout][2] floats (see sg_conv_skip_desc) */
  int x2_norelu;  /* 1: SG_PIX_RELU applies to x only, not to the skip input (the first discriminator block: the skip reads the image). With C2 == 8
                     (the RGB image padded to 8 channels) w2q is [Cout][4][8] = sg_quad_pack_batch mode 5 */
} sg_convq_desc;
int sg_conv2d_q_stat_rows(const sg_convq_desc* d);
int sg_conv2d_q(const sg_convq_desc* d, sg_stream_t stream);
int sg_conv2d_q_ok(const sg_convq_desc* d);            /* 1 when sg_conv2d_q takes the problem */
int sg_quad_pack(int dtype, int mode, const void* src, void* dst, int M, int Cs, sg_stream_t stream);
/* the same for n images in one launch (items_dev: the table in device memory, items_host: the same table on the host) */
typedef struct { const void* src; void* dst; int M, Cs, mode, pad_; } sg_quad_item;   /* mode 4: dst[M][Cs] = src[M][Cs] / 4 (the fused skip's filter); mode 5 (Cs == 8): dst[M][4][8] = src[M][8] / 4, four times */
int sg_quad_pack_batch(int dtype, const sg_quad_item* items_dev, const sg_quad_item* items_host, int n, sg_stream_t stream);
/* weight gradient of the same two forms: dw[co][r][s][c] (the 3x3 gradient image, fp32, accumulated) += alpha * (gradient w.r.t. the
 * quad filter, folded back through the transpose of sg_quad_pack's sums); dbias[co] += sum of dy (optional). POOL: x fine, dy low;
 * UP: x low, dy fine. work: scratch for the deterministic two-stage reduction, sized by sg_conv2d_q_wgrad_plan (*splits == 0: not eligible). */
typedef struct {
  int dtype, form;
  int N, Hl, Wl;
  int C, ldx, x_flags;            /* SG_PIX_RELU */
  int Cout, ldg;
  float alpha; const float* alpha_ptr;
  const void* x; const void* dy; float* dw; float* dbias;
  float* work; long long work_floats;
  int splits;                     /* 0 = auto */
} sg_convq_wgrad_desc;
int sg_conv2d_q_wgrad_plan(const sg_convq_wgrad_desc* d, int* splits, long long* work_floats);
int sg_conv2d_q_wgrad(const sg_convq_wgrad_desc* d, sg_stream_t stream);
/* sg_prof_collect with a fourth column per kind: FLOPs the launches really executed (quad launches: 16 / 36 of the algorithmic count) */
int sg_prof_collect_ex(double* out, int nkinds);
/* per kernel family of the convolution engine (ids: csrc/common.h SG_ENG_*, names: bench.py ENGINES): out[id * 5 + {0..4}] = {launches, total ms,
 * algorithmic FLOPs, executed FLOPs, algorithmic HBM bytes}. Does not reset the log: call it BEFORE sg_prof_collect / sg_prof_collect_ex. */
int sg_prof_collect_tags(double* out, int ntags);

/* dw[co][r][s][c] += alpha * sum_{n,ho,wo} dy'[n,ho,wo,co] * x'[n, ho*stride-pad+r, wo*stride-pad+s, c] (fp32 atomics) */
typedef struct {
  int dtype;
  int N;
  int xHs, xWs, C, ldx, x_flags;
  int gHs, gWs, Cout, ldg, g_flags;
  int Ho, Wo;
  int R, S, stride, pad_h, pad_w;
  float alpha;
  const void* x; const void* dy; float* dw;
  const float* alpha_ptr;          /* optional device scalar multiplied into alpha */
  int splits;                      /* 0 = auto */
  int no_tr;                       /* 1 = use the gather fragment path instead of ds_read_b64_tr_b16 (test hook) */
  float* work; long long work_floats; /* optional scratch for the deterministic two-stage split-K (see sg_conv2d_wgrad_plan);
                                         without it k-splits fall back to fp32 atomics */
  float* dbias;                    /* optional: dbias[co] += sum over the stored dy pixels of dy[.., co] (the bias gradient of the same layer),
                                      computed from the dy fragments the weight-gradient kernel already holds. Only honoured when
                                      sg_conv2d_wgrad_fuses_bias(d) == 1; otherwise the caller runs sg_colsum */
} sg_conv_wgrad_desc;
int sg_conv2d_wgrad(const sg_conv_wgrad_desc* d, sg_stream_t stream);
/* 1 when sg_conv2d_wgrad will also produce d->dbias for this problem (halo kernel + workspace present) */
int sg_conv2d_wgrad_fuses_bias(const sg_conv_wgrad_desc* d);
/* the k-split count the launcher will use for this problem and the scratch floats its two-stage reduction wants */
int sg_conv2d_wgrad_plan(const sg_conv_wgrad_desc* d, int* splits, long long* work_floats);

/* batched OUT[b][j][i] = beta*res + alpha * sum_k P(i,k) Q(j,k) + bias[i]
 * form 0 ("KC"): operand stored [row][k] (k contiguous); form 1 ("MC"): stored [k][row] (row contiguous) */
typedef struct {
  int dtype, p_form, q_form;
  int I, J, K, batch;
  const void* p; long long p_bstride; int ldp;
  const void* q; long long q_bstride; int ldq;
  void* out; long long out_bstride; int ldo;
  const float* bias; const void* res; long long res_bstride; int ldr; float beta;
  float alpha; const float* alpha_ptr;
  int epi_flags; int splits; int no_tr;
} sg_gemm_desc;
int sg_gemm(const sg_gemm_desc* d, sg_stream_t stream);
/* n small fp32 linear layers in ONE launch: out[b][o] = bias[o] + sum_k w[o][k] y[b][k], b < B (the same B for every item).
 * Replaces the per-layer gain(y) / bias(y) products of ConditionalBatchNorm2d (reference src/utils/ops.py:21-27, the ten conditional batch norms of a BigGAN
 * generator forward, src/models/big_resnet.py:139-163): every conditioning vector is known when the forward starts. items_dev: the table in device
 * memory, items_host: the same table on the host. */
typedef struct { const float* w; const float* y; const float* bias; float* out; int rows, K, ldy, ldo; } sg_linear_item;
int sg_linear_group(const sg_linear_item* items_dev, const sg_linear_item* items_host, int n, int B, sg_stream_t stream);

/* ---- layout / elementwise ------------------------------------------------------------------------------ */
/* fp32 NCHW -> T NHWC (ldo = channel pitch of the destination; ldo > C: channels C .. ldo - 1 of every row are written as zeros) */
int sg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int ldo, sg_stream_t s);
/* T NHWC -> fp32 NCHW, optional tanh */
int sg_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int lds, int apply_tanh, sg_stream_t s);
/* d_pre(NHWC,T) = d_out(NCHW fp32) * (1 - y^2), y = NCHW fp32 tanh output (apply_tanh=0: plain layout change) */
int sg_nchw_grad_to_nhwc(int dtype, const float* dy, const float* y, void* dst, int N, int C, int H, int W, int ldo, int apply_tanh, sg_stream_t s);
int sg_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, sg_stream_t s);
int sg_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, sg_stream_t s);
/* 2x2 max pooling over a column slice [c0, c0+C) of a [N,H,W,ldx] tensor; idx gets the argmax (0..3) */
int sg_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, uint8_t* idx, int N, int H, int W, int C, sg_stream_t s);
int sg_maxpool2_bwd(int dtype, const void* dy, int ldy, const uint8_t* idx, void* dx, int ldx, int N, int H, int W, int C, sg_stream_t s);
/* row softmax: fp32 logits [rows][cols] -> probabilities as T; bwd: ds(T) = p * (dp(fp32) - sum(dp*p)) */
int sg_softmax_rows(int dtype, const float* s_in, void* p_out, long long rows, int cols, sg_stream_t st);
int sg_softmax_rows_bwd(int dtype, const void* p, const float* dp, void* ds, long long rows, int cols, sg_stream_t st);
/* element type conversion between SG dtypes */
int sg_convert(int src_dtype, int dst_dtype, const void* x, void* y, long long n, sg_stream_t s);
/* out = a + relu(x) (identity-skip DiscBlock, reference models/big_resnet.py:221-242 with the in-place ReLU) ; dx = dy*(x>0) */
int sg_add_relu(int dtype, const void* a, const void* x, void* out, long long n, sg_stream_t s);
int sg_relu_mask(int dtype, const void* dy, const void* x, void* dx, long long n, sg_stream_t s);
/* y = a*x + b*y elementwise on T tensors */
int sg_axpby(int dtype, const void* x, void* y, long long n, float a, float b, sg_stream_t s);
/* out[0] (+)= sum(x*y) over n elements of T (fp32 accumulate, deterministic two-stage when accumulate=0) */
int sg_dot(int dtype, const void* x, const void* y, long long n, float* out, float scale, const float* scale_ptr, sg_stream_t s);
/* column sums of a [rows][C] T matrix (optionally through a >0 mask) into fp32 out[C] (+=) : conv bias gradient */
int sg_colsum(int dtype, const void* x, int ldx, const void* mask, int ldm, long long rows, int C, float* out, float alpha, sg_stream_t s);

/* ---- batch norm / conditional batch norm (reference eps = 1e-4, momentum 0.1) ---------------------------- */
/* partial[c] = {sum x, sum x^2} in fp64 (zeroed by the caller); rows = N*H*W */
int sg_bn_partial_stats(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_stream_t s);
/* mean, invstd from (all-reduced) partial sums; updates running stats when running_mean != NULL
 * (unbiased variance, momentum) -- torch.nn.functional.batch_norm training semantics */
/* partial[2 c + {0, 1}] (fp64, caller-zeroed) += the per-tile sums a convolution epilogue wrote (sg_conv_skip_desc.stats / sg_convq_desc.stats) */
int sg_bn_stats_from_tiles(const float* stats, int nrows, int C, double* partial, sg_stream_t s);
int sg_bn_finalize(const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, sg_stream_t s);
/* ---- data-parallel exchanges over RCCL (one process per GPU; replaces DistributedDataParallel's bucketed gradient all-reduce and
 * SyncBatchNorm's statistics exchange, reference src/models/model.py:157-180). librccl is bound at run time (dlopen). ----------- */
typedef void* sg_comm_t; /* ncclComm_t */
/* rank 0: 128-byte RCCL unique id, to be handed to every rank by the host (any out-of-band channel) */
int sg_comm_unique_id(void* out128);
int sg_comm_init_rank(const void* id128, int nranks, int rank, sg_comm_t* comm);
int sg_comm_size(sg_comm_t comm, int* nranks);
int sg_comm_destroy(sg_comm_t comm);
/* in-place sum of a flat device buffer over the ranks of comm, enqueued on s; dtype SG_DTYPE_F32 (gradient arena) or SG_DTYPE_F64 (BN terms) */
int sg_allreduce_flat(sg_comm_t comm, void* buf, long long count, int dtype, sg_stream_t s);
/* sync-BN statistics in one call: local partial sums -> all-reduce over comm (NULL = single rank) -> mean / invstd (+ running
 * statistics when running_mean != NULL), count = rows * nranks; partial = fp64 scratch [2*C] */
int sg_bn_stats_sync(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_comm_t comm, float eps, float momentum,
                     float* mean, float* invstd, float* running_mean, float* running_var, sg_stream_t s);
/* reduce-scatter / all-gather halves of the gradient exchange (the sharded optimizer step: every rank reduces and updates 1/world of the arena, then the updated
 * parameters are gathered; the gather can stay in flight behind the next forward of the OTHER network). In place: the shard of rank r is
 * buf[r * per_rank, (r + 1) * per_rank); count = per_rank elements, fp32. */
int sg_reduce_scatter_flat(sg_comm_t comm, float* buf, long long per_rank, sg_stream_t s);
int sg_allgather_flat(sg_comm_t comm, float* buf, long long per_rank, sg_stream_t s);

/* ---- peer-store mailboxes: the sync-BN exchange FUSED INTO the statistics kernel (csrc/p2p.hip). Each rank owns a fine-grained device buffer
 * that every peer maps through an IPC handle; sg_bn_finalize_p2p's kernel writes this rank's 2*C partial sums into every peer's buffer (xGMI stores,
 * 8-byte {epoch | payload} granules), waits for the peers' granules of the same call and finalises mean / invstd / running statistics of the global
 * batch in the same launch -- one xGMI round trip instead of a collective call between two kernels (replaces SyncBatchNorm's all_gather, reference
 * src/models/model.py:161-165). Calls must be issued in the same order on every rank. ---- */
typedef void* sg_p2p_t;
/* allocates this rank's mailbox (room for max_doubles values per call) and returns its 64-byte IPC handle for the host to hand to every peer */
int sg_p2p_create(int world, int rank, long long max_doubles, sg_p2p_t* out, void* handle_out64);
/* handles: world x 64 bytes in rank order (entry [rank] is ignored) */
int sg_p2p_connect(sg_p2p_t p, const void* handles);
int sg_p2p_destroy(sg_p2p_t p);
/* number of granule waits that ran into the spin limit so far (a peer died or skipped a call): 0 on a healthy job */
int sg_p2p_timeouts(sg_p2p_t p, int* count);
/* in-place sum of n <= max_doubles doubles over the ranks in rank order (bit-identical on every rank); one launch */
int sg_p2p_allreduce_f64(sg_p2p_t p, double* buf, int n, sg_stream_t s);
/* partial [2*C]: this rank's sums; count = local rows x world. Exchange + finalize in ONE kernel */
int sg_bn_finalize_p2p(sg_p2p_t p, const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, sg_stream_t s);
/* sg_bn_stats_sync over the mailboxes: local partial sums, then the fused exchange + finalize */
int sg_bn_stats_sync_p2p(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_p2p_t p, float eps, float momentum,
                         float* mean, float* invstd, float* running_mean, float* running_var, sg_stream_t s);
/* eval mode: mean/invstd from running stats */
int sg_bn_from_running(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd, sg_stream_t s);
/* y = relu?( (x-mean)*invstd * gain + bias ), gain/bias either per-channel [C] (stride_n = 0) or per-sample [N][C] */
int sg_bn_apply(int dtype, const void* x, void* y, int N, long long HW, int C, const float* mean, const float* invstd,
                const float* gain, const float* bias, int gb_stride_n, int relu, sg_stream_t s);
/* backward, stage 1: per-(n,c) sums of dy' and dy'*xhat where dy' = dy * relu-mask; sums[N][C][2] fp32 (overwritten) */
int sg_bn_bwd_reduce(int dtype, const void* x, const void* dy, int N, long long HW, int C, const float* mean, const float* invstd,
                     const float* gain, const float* bias, int gb_stride_n, int relu, float* sums, sg_stream_t s);
/* stage 2: from sums -> dgain/dbias ([N] rows of pitch gb_stride_n when per-sample else [C], accumulated +=) and the per-channel
 * batch terms chan[C][2] = {sum_n gain*S1, sum_n gain*S2} in fp64 (for the cross-rank all-reduce) */
int sg_bn_bwd_finalize(const float* sums, int N, int C, const float* gain, int gb_stride_n, float* dgain, float* dbias,
                       double* chan, sg_stream_t s);
/* stage 3: dx = invstd * (gain*dy' - (chan0 + xhat*chan1)/count)   (count = global N*H*W; use_batch_stats=0 -> eval-mode BN) */
int sg_bn_bwd_apply(int dtype, const void* x, const void* dy, void* dx, int N, long long HW, int C, const float* mean,
                    const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu,
                    const double* chan, double count, int use_batch_stats, sg_stream_t s);
/* same, plus `res` (dx's shape and dtype, may be NULL) added to the result: the gradient the BN input received through another branch
 * (a residual block's skip path) -- replaces autograd's separate accumulation launch */
int sg_bn_bwd_apply_res(int dtype, const void* x, const void* dy, void* dx, int N, long long HW, int C, const float* mean,
                        const float* invstd, const float* gain, const float* bias, int gb_stride_n, int relu,
                        const double* chan, double count, int use_batch_stats, const void* res, sg_stream_t s);

/* second-order backward of BN's data gradient (WGAN-GP double backward; reference utils/losses.py:268-275,301-316 reach
 * it through torch autograd). u = dL/d(dx). Per-channel gain only. sums[N][C][5] fp32 (caller zeroes) ->
 * chan[C][5] fp64 = {sum u, sum u*xhat, sum dy', sum dy'*xhat, sum u*dy'} -> g_dy = dL/d(dy), g_x = dL/dx (either may be
 * NULL), dgain += dL/dgain from this rank's chan_local and the all-reduced chan. */
int sg_bn_bwd2_reduce(int dtype, const void* x, const void* dy, const void* u, int N, long long HW, int C, const float* mean,
                      const float* invstd, const float* gain, const float* bias, int relu, float* sums, sg_stream_t s);
int sg_bn_bwd2_finalize(const float* sums, int N, int C, double* chan, sg_stream_t s);
int sg_bn_bwd2_dgain(const double* chan_local, const double* chan, double count, const float* invstd, int C,
                     int use_batch_stats, float* dgain, sg_stream_t s);
int sg_bn_bwd2_apply(int dtype, const void* x, const void* dy, const void* u, void* g_dy, void* g_x, int N, long long HW, int C,
                     const float* mean, const float* invstd, const float* gain, const float* bias, int relu, const double* chan,
                     double count, int use_batch_stats, sg_stream_t s);

/* ---- spectral norm (torch.nn.utils.spectral_norm, eps 1e-6, one power iteration per forward) -------------- */
typedef struct {
  const float* w;     /* weight_orig viewed as [rows][cols] (OIHW flattened; [num_embeddings][dim] for embeddings) */
  float* u; float* v; /* power-iteration state, updated in place (rows / cols) */
  float* sigma;       /* out: 1 float */
  float* u_snap; float* v_snap; /* optional copies of the updated u, v for this forward's backward (may be NULL) */
  void* w_fwd;        /* out: W/sigma as T, [Cout][R][S][Cin] (or [rows][cols] when RS == 1) ; may be NULL */
  void* w_dgrad;      /* out: W/sigma as T, [Cin][R-1-r][S-1-s][Cout] ; may be NULL */
  float* w_f32;       /* out: W/sigma in fp32 [rows][cols] natural layout ; may be NULL */
  int rows, cols;     /* rows = Cout, cols = Cin*R*S */
  int Cin, RS;        /* cols == Cin*RS */
  int do_power_iter;  /* module.training */
  int apply_sn;       /* 0: plain layer, only emit operand images with sigma = 1 */
  int rows_pad;       /* w_fwd gets rows_pad >= rows rows (extra rows zero) ; 0 = rows */
  long long work_off; /* this layer's slice of work[]: needs 16*cols + rows floats (16 = the row groups of the W^T u pass) */
  int trans;          /* 1: ConvTranspose2d weight [Cin][Cout][R][S] (spectral norm over dim 1, rows = Cout) */
  int dgrad_noflip;   /* 1: w_dgrad = [Cin][r][s][Cout] without the spatial flip (strided / transposed convolutions) */
  int Cin_pad;        /* >= Cin (0 = Cin): channel pitch of w_fwd ([Cout][R][S][Cin_pad], zero filled) -- thin inputs (RGB) are
                         carried as 8-channel tensors so the 16-byte loaders apply; w_dgrad is [Cin][R][S][max(rows,rows_pad)] */
} sg_sn_layer;
/* runs all layers of a network in six batched launches (W^T u, v, W v, u / sigma, forward images, data-gradient images), each a flat table of exactly the tiles
 * its layers have (any mix of layer shapes in one call; tables longer than 64 layers are walked in runs of 64). `layers` is a DEVICE array of n descriptors;
 * work[] is a device scratch; each layer owns the slice [work_off, work_off + 16*cols + rows) */
int sg_sn_forward(int dtype, const sg_sn_layer* layers_dev, const sg_sn_layer* layers_host, int n, float eps, float* work, long long work_floats, sg_stream_t s);

typedef struct {
  const float* dwt;   /* gradient w.r.t. the normalised weight, fp32, [Cout][R][S][Cin] (or natural [rows][cols] if natural=1) */
  const float* w;     /* weight_orig [rows][cols] */
  const float* u; const float* v; const float* sigma; /* snapshot of that forward */
  float* dw;          /* grad of weight_orig, accumulated (+=), natural OIHW layout */
  int rows, cols, Cin, RS, natural, apply_sn, trans;
  int Cin_pad;        /* channel pitch of dwt when natural == 0 (0 = Cin) */
} sg_sn_bwd_layer;
int sg_sn_backward(const sg_sn_bwd_layer* layers_dev, const sg_sn_bwd_layer* layers_host, int n, float* work, long long work_floats, sg_stream_t s);

/* ---- embedding, D head, losses ---------------------------------------------------------------------------- */
int sg_embedding_fwd(const float* table, const int64_t* idx, float* out, int B, int dim, int num, sg_stream_t s);
int sg_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, int B, int dim, int num, sg_stream_t s);
/* h[b][c] = sum_hw relu(x[b,hw,c]) (fp32 out) ; bwd: dx[b,hw,c] = dh[b][c] * (x>0) */
int sg_relu_sum_hw_fwd(int dtype, const void* x, float* h, int B, int HW, int C, sg_stream_t s);
int sg_relu_sum_hw_bwd(int dtype, const void* x, const float* dh, void* dx, int B, int HW, int C, sg_stream_t s);
/* adv[b] = <h[b], w1> + b1 + <emb[b], h[b]>  (emb may be NULL -> unconditional) */
int sg_pd_head_fwd(const float* h, const float* w1, const float* b1, const float* emb, float* adv, int B, int C, sg_stream_t s);
int sg_pd_head_bwd(const float* h, const float* w1, const float* emb, const float* dadv, float* dh, float* dw1, float* db1,
                   float* demb, int B, int C, sg_stream_t s);
/* adversarial losses: kind 0 hinge, 1 wasserstein, 2 vanilla(BCE-with-logits). loss[0] overwritten; gradients of the
 * mean-reduced loss written to d_real / d_fake */
int sg_loss_d(int kind, const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake, sg_stream_t s);
int sg_loss_g(int kind, const float* fake, int B, float* loss, float* d_fake, sg_stream_t s);

/* WGAN-GP (reference utils/losses.py:301-316): interpolates[b] = alpha[b]*real[b] + (1-alpha[b])*fake[b] over rows of n
 * floats; penalty = mean_b (||grads[b]||_2 - 1)^2 with norms[B] kept for the backward; dgrads = gout * d penalty / d grads.
 * sg_masked_sum_hw: out[b][c] = sum_hw t[b,hw,c] * (x[b,hw,c] > 0)  (adjoint of sg_relu_sum_hw_bwd in the second-order pass) */
int sg_interp_rows(const float* real, const float* fake, const float* alpha, float* out, int B, long long n, sg_stream_t s);
/* kind 0: mean_b (||g_b|| - 1)^2 (WGAN-GP / DRA, losses.py:301-335); 1: 0.5 mean_b ||g_b||^2 (R1, :355-361); 2: max_b ||g_b||^2
 * (maxGP, :338-352). norms: B + 1 floats (row norms; slot B = arg-max row of kind 2) */
int sg_gp_fwd(int kind, const float* grads, int B, long long n, float* norms, float* loss, sg_stream_t s);
int sg_gp_bwd(int kind, const float* grads, const float* norms, const float* gout, float* dgrads, int B, long long n, sg_stream_t s);
int sg_masked_sum_hw(int dtype, const void* t, const void* x, float* out, int B, int HW, int C, sg_stream_t s);

/* ---- fused self-attention scores (reference utils/ops.py:83-103: softmax(theta . maxpool(phi)^T) and its backward), bf16 only.
 * theta [B][HW][Dp], phi [B][HW4][Dp] (pooled), g [B][HW4][Cg] (pooled), dO [B][HW][Cg]; Dp <= 32, Dp % 8 == 0.
 *   sg_attn_probs_fwd: P [B][HW][HW4] bf16 = row softmax of the scores, lse [B][HW] = row log-sum-exp (scores stay in registers)
 *   sg_attn_ds_bwd:    dS [B][HW][HW4] bf16 = P * (dP - sum_k P dP), dP = dO . g^T, P recomputed from theta / phi / lse
 * sg_attn_fused_ok returns 1 when the shape is supported (HW % 128 == 0, HW4 % 256 == 0, HW4 <= 2048, Cg <= 128). */
int sg_attn_fused_ok(int B, int HW, int HW4, int Dp, int Cg);
/* fused forward of the attention core: O = softmax(theta phi^T) g per image in one launch (probabilities stored only when P != NULL) */
int sg_attn_fwd_fused_ok(int B, int HW, int HW4, int Dp, int Cg);
/* the P == NULL form streams keys and values in 256-key chunks: no bound on HW4 (16384 queries x 4096 keys of BigGAN-deep-256's D,
 * reference models/big_resnet_deep_legacy.py:80-95); sg_attn_fwd_fused accepts P == NULL whenever this returns 1 */
int sg_attn_fwd_flash_ok(int B, int HW, int HW4, int Dp, int Cg);
int sg_attn_fwd_fused(const void* theta, const void* phi, const void* g, void* P, float* lse, void* O, float* O32, int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s);
/* fused backward of the attention core: dtheta, dphi, dg from theta / phi / g / dO / lse with P and dS recomputed on the fly (two launches:
 * query side + key side); delta = fp32 scratch [B][HW]. O32 = the unrounded fp32 copy of the forward output [B][HW][Cg] that
 * sg_attn_fwd_fused writes on request (P == NULL path): delta_q = dO_q . O_q, one key pass on the query side; NULL: delta from an extra pass
 * over the keys */
int sg_attn_bwd_fused_ok(int B, int HW, int HW4, int Dp, int Cg);
int sg_attn_bwd_fused(const void* theta, const void* phi, const void* g, const void* dO, const float* O32, const float* lse, float* delta, void* dtheta, void* dphi,
                      void* dg, int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s);
int sg_attn_probs_fwd(const void* theta, const void* phi, void* P, float* lse, int B, int HW, int HW4, int Dp, sg_stream_t s);
int sg_attn_ds_bwd(const void* theta, const void* phi, const void* g, const void* dO, const float* lse, void* dS,
                   int B, int HW, int HW4, int Dp, int Cg, sg_stream_t s);

/* BigGAN-deep skips (reference models/big_resnet_deep_legacy.py:53-56,74-77,236-238): channel slice (+ nearest x up, up in {1,2})
 * y [N][Hs*up][Ws*up][C] from x [N][Hs][Ws][ldx], its adjoint (dx gets all ldx channels, zeros beyond C), and a pitched channel copy */
int sg_slice_up_fwd(int dtype, const void* x, void* y, int N, int Hs, int Ws, int ldx, int C, int up, sg_stream_t s);
int sg_slice_up_bwd(int dtype, const void* dy, void* dx, int N, int Hs, int Ws, int ldx, int C, int up, sg_stream_t s);
int sg_copy_channels(int dtype, const void* src, int ld_src, void* dst, int ld_dst, long long rows, int C, sg_stream_t s);

/* ---- optimizer / EMA over flat arenas -------------------------------------------------------------------- */
/* torch.optim.Adam (no amsgrad, no weight decay unless wd != 0) on a flat fp32 arena, fused with the EMA of
 * the generator copy (ema may be NULL): p_ema = p_new.lerp(p_ema, decay) (reference utils/ema.py:27-35) */
int sg_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr, float beta1, float beta2,
                float eps, float wd, int step, float ema_decay, float grad_scale, sg_stream_t s);
int sg_ema_lerp(const float* src, float* ema, long long n, float decay, sg_stream_t s);

/* LeCam regulariser (reference src/utils/losses.py:262-265): loss = mean relu(real - ema_fake)^2 + mean relu(ema_real - fake)^2 and its
 * gradient w.r.t. the two logit vectors */
int sg_lecam(const float* real, const float* fake, int B, float ema_real, float ema_fake, float* loss, float* d_real, float* d_fake, sg_stream_t s);
/* uint8 input path (reference src/data_util.py:92-94): [N][H][W][3] uint8 (+ optional per-image horizontal-flip flags) ->
 * T [N][H][W][cpad] = (x/255 - 0.5)/0.5, zero-filled channels */
int sg_u8_to_nhwc(int dtype, const uint8_t* x, const uint8_t* flip, void* y, int N, int H, int W, int cpad, sg_stream_t s);

/* ---- evaluation path -------------------------------------------------------------------------------------- */
/* fp32 NCHW [-1,1] -> uint8 quantise (trunc((x+1)/2*255+0.5), clamp) -> bilinear (align_corners=False) resize to
 * OHxOW -> clip(0,255) -> (x/255-0.5)/0.5 -> T NHWC.  quant_out (optional) receives the uint8 NCHW image. */
int sg_quantize_resize_normalize(int dtype, const float* x, void* out, uint8_t* quant_out, int N, int C, int H, int W,
                                 int OH, int OW, int quantize, sg_stream_t s);
/* generic pooling on NHWC: mode 0 max, 1 avg (count_include_pad), 2 avg excluding padding */
int sg_pool2d(int dtype, const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad, int mode, int ldy, int c_off, sg_stream_t s);
/* PIL resizers of reference src/utils/resize.py:39-78 ("clean": bicubic, "friendly": bilinear for InceptionV3_tf) on quantised images: Pillow's
 * separable, support-scaled resampling (horizontal then vertical, double accumulation, float intermediate) with host-computed coefficient
 * windows bounds[o] = {first, count}, kk[o][ksize]; tmp = fp32 scratch [N][C][H][OW]; out = NHWC, (v / 255 - 0.5) / 0.5 */
int sg_pil_resize_normalize(int dtype, const float* x, void* out, float* tmp, int N, int C, int H, int W, int OH, int OW,
                            const int* bounds_h, const double* kk_h, int ksize_h, const int* bounds_v, const double* kk_v, int ksize_v,
                            int quantize, sg_stream_t s);
/* global average pool [N,HW,C] -> fp32 [N,C] */
int sg_global_avgpool(int dtype, const void* x, float* y, int N, int HW, int C, sg_stream_t s);
/* hits[n] = 1 iff the true class is within the top k of scores[n][0..ncls) with sklearn's tie rule (higher index wins a tie) */
int sg_topk_hits(const float* scores, int ld, int ncls, const int64_t* labels, int k, int N, uint8_t* hits, sg_stream_t s);
/* top-k training of G (reference src/worker.py:565-566 torch.topk(adv_output, k).values): vals/idx = the k largest of x[0..n),
 * descending, lower index first among equals; scatter = its gradient (dx[idx[r]] = g[r], 0 elsewhere) */
int sg_topk_select(const float* x, int n, int k, float* vals, int* idx, sg_stream_t s);
int sg_topk_scatter(const float* g, const int* idx, int k, float* dx, int n, sg_stream_t s);
/* sum_f[c] += sum_n f[n][c];  sum_ff[c1][c2] += sum_n f[n][c1] f[n][c2]  (fp64 accumulators; FID moments) */
int sg_feat_moments_accumulate(const float* f, int n, int C, double* sum_f, double* sum_ff, sg_stream_t s);
/* training "basket" out of a uint8 data set resident in HBM ([N][H][W][3], the HDF5 / in-memory layout of reference src/utils/hdf5.py:35-97):
 * dst[b] = (optionally mirrored) src[idx[b]], lab_dst[b] = lab_src[idx[b]] (both label pointers NULL: images only) */
int sg_gather_images_u8(const uint8_t* src, const int64_t* idx, const uint8_t* flip, uint8_t* dst, int B, int H, int W,
                        const int64_t* lab_src, int64_t* lab_dst, sg_stream_t s);
/* ---- class-conditioning heads and losses (reference src/utils/losses.py:40-165,242-252; src/models/big_resnet.py:307-333,380-413).
 * All fp32, [rows][cols] row-major; losses are means over rows and return the gradient of that mean in the same call. */
int sg_row_normalize_fwd(const float* x, float* y, float* inv, int rows, int cols, float eps, sg_stream_t s);   /* F.normalize(dim=1) */
int sg_row_normalize_bwd(const float* y, const float* inv, const float* dy, float* dx, int rows, int cols, sg_stream_t s);
int sg_row_dot(const float* a, const float* b, float* p, int rows, int cols, sg_stream_t s);                    /* p[r] = <a[r], b[r]> */
int sg_row_scale(const float* g, const float* x, float* out, int rows, int cols, int accumulate, sg_stream_t s); /* out[r] (+)= g[r] x[r] */
/* kind 0: cross entropy (AC head), 1: Crammer-Singer multi-hinge (MH head): row_loss[rows], loss[1] = mean, dz = d loss / d z */
int sg_class_loss(int kind, const float* z, const int64_t* label, int rows, int cols, float* row_loss, float* loss, float* dz, sg_stream_t s);
/* kind 0: conditional contrastive loss (2C), 1: data-to-data cross entropy (D2D-CE) over S = cos(e_i, e_j) [B][B], p = cos(e_i, proxy_i) */
int sg_contrastive_loss(int kind, const float* S, const float* p, const int64_t* label, int B, float temperature, float m_p,
                        float* row_loss, float* loss, float* dS, float* dp, sg_stream_t s);
int sg_gather_cols(const float* z, const int64_t* label, int rows, int cols, float* out, sg_stream_t s);       /* MD head: z[r][label[r]] */
int sg_scatter_cols(const float* g, const int64_t* label, int rows, int cols, float* dz, sg_stream_t s);
/* ---- precision / recall / density / coverage (reference src/metrics/prdc.py:87-168) on squared distances. The cross term
 * D[r][c] = |y_c|^2 - 2 x_r . y_c comes from sg_gemm (fp32, alpha = -2, bias = |y|^2) one row block at a time. */
int sg_row_sqnorm(const float* f, int n, int C, float* sq, sg_stream_t s);
/* out[r] = k-th smallest (1-based, k <= 16) of max(0, D[r][c] + row_add[r]) */
int sg_kth_smallest_rows(const float* D, long long ld, int rows, int cols, int k, const float* row_add, float* out, sg_stream_t s);
/* col_cnt[c] += #{r : d2 < r2_row[r]}; row_any[r] = any_c d2 < r2_col[c]; row_min[r] = min_c d2, with d2 = max(0, D[r][c] + row_add[r]) */
int sg_prdc_rows(const float* D, long long ld, int rows, int cols, const float* row_add, const float* r2_row, const float* r2_col,
                 int* col_cnt, uint8_t* row_any, float* row_min, sg_stream_t s);
/* ---- tr sqrtm(S1 S2) of the Frechet distance (reference src/metrics/fid.py:34-62, scipy.linalg.sqrtm on the host) in fp64 on the device:
 * Cholesky factors L1, L2, B = L2^T L1, singular values of B by one-sided Jacobi sweeps, sum of the row norms. [n][n] row-major doubles. */
int sg_chol_lower(double* A, int n, int* flag, sg_stream_t s);          /* in place; *flag = 1 + first non-positive pivot, 0 if SPD */
int sg_dgemm_tn(const double* A, const double* B, double* C, int n, sg_stream_t s);   /* C = A^T B */
int sg_jacobi_sweep(double* M, int n, double* offd, sg_stream_t s);    /* n even; *offd = max |<r_p, r_q>| / (|r_p| |r_q|) met in the sweep */
int sg_row_norm_sum(const double* M, int n, double* out, sg_stream_t s);

/* ---- StyleGAN2 / StyleGAN3 native operators (the reference's only CUDA code, the .cu files of src/utils/style_ops) ----------------
 * sg_bias_act: y = clamp(gain * act(x + b)) and its gradient evaluators, the contract of the reference's `_plugin.bias_act(x, b, xref, yref,
 * dy, grad, dim, act, alpha, gain, clamp)` (bias_act.cpp:27-94, kernel bias_act.cu:23-147). Elementwise over n values; the bias element of
 * value i is b[(i / step_b) % size_b] (step_b = product of the dimensions behind `dim`). act = cuda_idx of bias_act.py:20-30 (1 linear,
 * 2 relu, 3 lrelu, 4 tanh, 5 sigmoid, 6 elu, 7 selu, 8 softplus, 9 swish). grad = 0: forward (xref / yref / dy unused); grad = 1: x is the
 * incoming gradient, yref the forward output (xref the forward input for swish); grad = 2: second-order term, dy the first-order gradient.
 * b, xref, yref, dy may be NULL. fp32 or bf16, all tensors of the same dtype, 16-byte aligned. */
int sg_bias_act(int dtype, const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, long long n,
                long long step_b, int size_b, int grad, int act, float alpha, float gain, float clamp, sg_stream_t s);
/* sg_upfirdn2d: pad -> upsample by zero insertion -> 2-D FIR filter -> decimate in one pass over `planes` = N * C image planes
 * (x [planes][H][W] -> y [planes][Ho][Wo], Ho = (H * upy + pady0 + pady1 - fh + downy) / downy, likewise Wo): the reference's
 * `_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)` (upfirdn2d.cpp:25-100, kernels
 * upfirdn2d.cu:29-104). f: fp32 [fh][fw] on the device; without flip_filter the operator is a true convolution (the filter is flipped).
 * Negative padding crops. gain multiplies the result. */
int sg_upfirdn2d(int dtype, const void* x, const float* f, void* y, int planes, int H, int W, int fh, int fw, int upx, int upy,
                 int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip_filter, float gain, sg_stream_t s);
/* sg_filtered_lrelu: the forward of the reference's fused `_plugin.filtered_lrelu` (filtered_lrelu.cpp / .cu; definition filtered_lrelu.py:120-155)
 * for SEPARABLE filters in ONE launch: y = downfir_fd(clamp(lrelu(upfir_fu(x + b; up, padding, gain up^2), slope) * gain); down) on [N][C][H][W]
 * -> [N][C][Ho][Wo], Ho = (H * up + py0 + py1 - (fu_n - 1) - (fd_n - 1) + (down - 1)) / down. fu / fd: fp32 taps on the device (a one-tap {1}
 * filter = identity), b: per-channel bias of x's dtype or NULL, clamp < 0 = none. The up-sampled intermediate stays in LDS. Returns -3 when a
 * tile exceeds the LDS budget (run the chain sg_bias_act -> sg_upfirdn2d -> sg_bias_act -> sg_upfirdn2d instead, as for 2-D filters). */
int sg_filtered_lrelu(int dtype, const void* x, const float* fu, const float* fd, const void* b, void* y, int N, int C, int H, int W,
                      int fu_n, int fd_n, int up, int down, int px0, int px1, int py0, int py1, float gain, float slope, float clamp,
                      int flip_filter, sg_stream_t s);

/* ---- differentiable augmentations in front of the discriminator and the consistency regularisers' loss -----------------------
 * sg_augment_fwd: y = cutout(translate(flip(contrast(saturation(brightness(x)))))) on fp32 NCHW image batches in ONE gather pass (one more partial-sum
 * pass when SG_AUG_CONTRAST is set). Replaces the per-operator chains of the reference's
 *   utils/diffaug.py:47-95  rand_brightness / rand_saturation / rand_contrast / rand_translation (zero fill) / rand_cutout
 *                           (cfgs.AUG.series_augment, config.py:586-587; called at worker.py:276-278,549-550)
 *   utils/cr.py:24-48       random_flip / random_translation over F.pad(mode='reflect')  (cfgs.AUG.parallel_augment, worker.py:326-354)
 * with the random draws made by the caller (the host mirrors draw them with the reference's own calls, in the reference's order):
 *   color [N][3] fp32 : brightness offset (rand - 0.5), saturation factor (rand * 2), contrast factor (rand + 0.5)
 *   geom  [N][5] int32: row shift, column shift (out[i][j] = in[i + shift_r][j + shift_c]), cutout centre row, cutout centre column (the reference's
 *                       offset_x / offset_y: the window is [centre - cut/2, centre - cut/2 + cut) with its indices CLAMPED into the image), flip flag
 * Operators are selected by `ops` and always applied in the order above; a policy in another order is several calls. |shift| < the axis length; with
 * SG_AUG_TRANSLATE_REFLECT |shift| <= max_t < min(H, W) (F.pad's own limit). x and y must not alias. work: sg_augment_work_floats(d) floats of scratch
 * (contrast only; may be NULL otherwise). sg_augment_bwd: dx = d<dy, y>/dx, the transposed gather (every source pixel collects its at most 3 x 3
 * images: no atomics, bit-identical between runs). Both are linear in their tensor argument up to the brightness offset, so second-order passes
 * (R1 / gradient penalties through an augmented batch) are sg_augment_fwd without SG_AUG_BRIGHTNESS on the incoming cotangent. */
#define SG_AUG_BRIGHTNESS 1
#define SG_AUG_SATURATION 2
#define SG_AUG_CONTRAST 4
#define SG_AUG_FLIP 8
#define SG_AUG_TRANSLATE 16          /* zero fill outside the image (diffaug.py:61-76) */
#define SG_AUG_TRANSLATE_REFLECT 32  /* reflect padding, no edge repeat (cr.py:33-48) */
#define SG_AUG_CUTOUT 64
typedef struct {
  int N, C, H, W;        /* fp32 [N][C][H][W], 1 <= C <= 4 */
  int ops;               /* SG_AUG_* */
  int cut_h, cut_w;      /* cutout window (SG_AUG_CUTOUT) */
  int max_t;             /* bound of |shift| (SG_AUG_TRANSLATE_REFLECT) */
  const float* color;    /* [N][3] or NULL when no colour operator is selected */
  const int* geom;       /* [N][5] or NULL when no geometric operator is selected */
} sg_aug_desc;
int sg_augment_work_floats(const sg_aug_desc* d);
int sg_augment_fwd(const sg_aug_desc* d, const float* x, float* y, float* work, sg_stream_t s);
int sg_augment_bwd(const sg_aug_desc* d, const float* dy, float* dx, float* work, sg_stream_t s);
/* torch.nn.MSELoss(reduction='mean') between two fp32 tensors of n elements (the reference's `l2_loss`, worker.py:116,329-361,603): loss[0] = mean (a - b)^2
 * by a fixed-order two-level sum (work: sg_mse_work_floats() floats); sg_mse_bwd: da = gout[0] * 2 (a - b) / n, db = -da (either may be NULL). */
int sg_mse_work_floats(void);
int sg_mse_fwd(const float* a, const float* b, long long n, float* work, float* loss, sg_stream_t s);
int sg_mse_bwd(const float* a, const float* b, const float* gout, long long n, float* da, float* db, sg_stream_t s);
/* Least-squares adversarial loss (reference utils/losses.py:216-223 d_ls / g_ls): loss[0] = mean(0.5 (real - 1)^2 + 0.5 fake^2) resp. mean(0.5 (fake - 1)^2)
 * with the gradient w.r.t. the logits from the same launch (sg_loss_d / sg_loss_g cover hinge, wasserstein, vanilla = logistic). */
int sg_loss_ls_d(const float* real, const float* fake, int B, float* loss, float* d_real, float* d_fake, sg_stream_t s);
int sg_loss_ls_g(const float* fake, int B, float* loss, float* d_fake, sg_stream_t s);
/* Feature matching (reference utils/losses.py:254-259, worker.py:588-596): loss[0] = mean_c |mean_b fake_h[b][c] - mean_b real_h[b][c]| on [B][C] fp32
 * features, d_fake = its gradient w.r.t. fake_h (real_h is detached by the caller). work: sg_fm_work_floats(C) floats. */
int sg_fm_work_floats(int C);
int sg_fm_loss(const float* real_h, const float* fake_h, int B, int C, float* work, float* loss, float* d_fake, sg_stream_t s);
/* InfoGAN's Q heads (reference models/big_resnet.py:337-344,373-377; utils/losses.py:369-375 normal_nll_loss; worker.py:607-618): y = exp(x) and its backward
 * dx = dy * y (the variance head); loss[0] = -mean_b sum_k [-0.5 log(2 pi var + 1e-6) - (x - mu)^2 / (2 var + 1e-6)] over [B][K] fp32 with d loss / d mu, d var. */
int sg_exp_fwd(const float* x, float* y, long long n, sg_stream_t s);
int sg_exp_bwd(const float* dy, const float* y, float* dx, long long n, sg_stream_t s);
int sg_normal_nll(const float* x, const float* mu, const float* var, int B, int K, float* loss, float* dmu, float* dvar, sg_stream_t s);
/* Adjoints the create_graph pass through SelfAttention needs next to the first-order entry points (R1 / gradient penalties on a discriminator with
 * attention: reference utils/losses.py:301-316,355-361 through utils/ops.py:83-103):
 *   sg_maxpool2_gather    y[q][c] = x[2x2 window of q][idx[q][c]][c]: the pooling with the argmax of sg_maxpool2_fwd held fixed (adjoint of sg_maxpool2_bwd)
 *   sg_softmax_rows_bwd2  gP = u * (dP - <P, dP>) - dP * <u, P> per row: d/dP of sg_softmax_rows_bwd's dS = P * (dP - <P, dP>), contracted with u (fp32)
 *   sg_scale_by_ptr       y = sigma[0] * x with sigma on the device (SelfAttention's output gain) */
int sg_maxpool2_gather(int dtype, const void* x, int ldx, const uint8_t* idx, void* y, int ldy, int N, int H, int W, int C, sg_stream_t s);
int sg_softmax_rows_bwd2(const float* P, const float* dP, const float* u, float* gP, long long rows, int cols, sg_stream_t s);
int sg_scale_by_ptr(int dtype, const void* x, const float* sigma, void* y, long long n, sg_stream_t s);
/* tanh of the generator's output in a create_graph pass (latent optimisation, reference utils/losses.py:278-298): t = dy * (1 - y^2) and its derivative with
 * respect to y contracted with g: out = -2 g dy y (fp32) */
int sg_tanh_bwd(const float* dy, const float* y, float* t, long long n, sg_stream_t s);
int sg_tanh_bwd2(const float* g, const float* dy, const float* y, float* out, long long n, sg_stream_t s);
/* Weight clipping (reference worker.py:489-492, LOSS.apply_wc): p[i] = clamp(p[i], lo, hi) over a flat fp32 parameter buffer, one pass.
 * Adaptive pseudo augmentation (reference utils/apa_aug.py:10-21): out[n] = flag[n] ? a[n] : b[n] for N rows of `row` floats (a = fake, b = real images).
 * The ADA / APA heuristic's accumulator (worker.py:285-289,478-481): acc[0] += sum_b sign(logit[b]), acc[1] += B, without a host round trip. */
int sg_clamp_flat(float* p, long long n, float lo, float hi, sg_stream_t s);
int sg_select_rows(const uint8_t* flag, const float* a, const float* b, float* out, int N, long long row, sg_stream_t s);
int sg_sign_count(const float* logit, int B, float* acc, sg_stream_t s);
/* Image-side operators of the adaptive discriminator augmentation pipeline (reference utils/ada_aug.py:178-353; the per-image 3 x 3 / 4 x 4 transforms are
 * composed by the host mirror from the reference's draws), fp32 NCHW, each with its exact gather-form adjoint:
 *   sg_reflect_pad2d_fwd / _bwd   F.pad(mode='reflect') of `planes` [H][W] images by (l, r, t, b) < the image size (ada_aug.py:265)
 *   sg_affine_sample_fwd / _bwd   F.affine_grid(theta [N][2][3], align_corners=False) + grid_sample(bilinear, zeros) in one pass: x [N][C][Hi][Wi] ->
 *                                 y [N][C][Ho][Wo] (ada_aug.py:276-277); _bwd: dx from dy (gradient w.r.t. the image; theta is a draw, not a parameter)
 *   sg_color_affine               y = M[:, :, :3] x + M[:, :, 3] per image, M [N][3][4] (transpose = 1: the adjoint M[:, :, :3]^T dy); C == 1: y = x * M[n][0][0] + M[n][0][3]
 *                                 (ada_aug.py:339-347) */
int sg_reflect_pad2d_fwd(const float* x, float* y, int planes, int H, int W, int l, int r, int t, int b, sg_stream_t s);
int sg_reflect_pad2d_bwd(const float* dy, float* dx, int planes, int H, int W, int l, int r, int t, int b, sg_stream_t s);
int sg_affine_sample_fwd(const float* x, const float* theta, float* y, int N, int C, int Hi, int Wi, int Ho, int Wo, sg_stream_t s);
int sg_affine_sample_bwd(const float* dy, const float* theta, float* dx, int N, int C, int Hi, int Wi, int Ho, int Wo, sg_stream_t s);
int sg_color_affine(const float* x, const float* M, float* y, int N, int C, int HW, int transpose, sg_stream_t s);
/*   sg_fir_reflect                one axis (0: along a row, 1: along a column) of the per-image separable amplification filter over the reflect-padded image
 *                                 (ada_aug.py:352-389): y = sum_t taps[n][t] x[reflect(pos + t - T/2)], T odd; transpose = 1: its adjoint
 *   sg_ada_noise_cutout           y = (x + noise * sigma[n]) * cutout-mask(cut[n] = centre x, centre y, size x, size y in image fractions) (ada_aug.py:393-416);
 *                                 noise / sigma or cut may be NULL; the adjoint is the call without noise */
int sg_fir_reflect(const float* x, const float* taps, float* y, int N, int C, int H, int W, int T, int axis, int transpose, sg_stream_t s);
int sg_ada_noise_cutout(const float* x, const float* noise, const float* sigma, const float* cut, float* y, int N, int C, int H, int W, sg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
