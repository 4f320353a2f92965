/* synthanatomy_hip.h -- C ABI of libsynthanatomy_hip.so (MI355X / gfx950).
 *
 * The reference (AmigoLab/SynthAnatomy) is pure Python: its hot paths call torch's cuDNN / cuBLAS / NCCL back-end
 * from src/networks/{vqvae,transformers,discriminator}.  This library is what the MI355X build puts *under* that
 * unchanged plugin surface.  Each entry point names the reference call site(s) it replaces (paths relative to the
 * reference root).  Conventions:
 *   - raw device pointers + explicit sizes; no torch types; activations are channels-last [N, D, H, W, C]
 *   - no allocation, no synchronisation, no global state; `stream` is a hipStream_t passed as void*
 *   - return 0 on success, a negative SA_E* code or a positive hipError_t otherwise; never throws
 *   - collectives are NOT part of this ABI: the reference reduces gradients / EMA statistics through torch.distributed (NCCL), and so does this
 *     build (backend "nccl" = RCCL over xGMI, runtime/ddp.py); there are no sa_comm_* entry points
 *   - dtype: SA_F32 (exact-f32 MFMA 16x16x4) or SA_BF16 (MFMA 16x16x32, fp32 accumulate)
 */
#ifndef SYNTHANATOMY_HIP_H
#define SYNTHANATOMY_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_ABI_VERSION 4   /* 4 (round 6): + sa_rotary_pairs, sa_subpixel_pool_fwd / _bwd, sa_comm_* (additions only; developer switches SA_DBG_CELLS / SA_DBG_DENSE_RING retired).  3 (round 5): sa_sample_step takes top_k.  2 (round 4): sa_local_attn_fwd/bwd + sa_epilogue grew trailing pointers in round 3, SA_F16 operand type, sa_mse partials; a caller built against 1 must not load this library */
enum { SA_F32 = 0, SA_BF16 = 1, SA_F16 = 2 /* IEEE half: FORWARD operand / activation type only (the reference's AMP dtype, src/engines/trainer.py:161-163); see sa_conv_fprop */ };
enum { SA_ACT_NONE = 0, SA_ACT_RELU = 1, SA_ACT_LRELU = 2, SA_ACT_GELU = 3 };
enum { SA_MASK_NONE = 0, SA_MASK_POS = 1 /* out *= (mask > 0) */, SA_MASK_LRELU = 2 /* out *= mask>0 ? 1 : slope */,
       SA_MASK_GELU = 3 /* out *= gelu'(mask) */ };
enum { SA_EINVAL = -1, SA_EUNSUPPORTED = -2, SA_ENOGPU = -3, SA_ECOMM = -4 /* an RCCL call failed: sa_comm_last_error() */ };
#define SA_MAX_TAPS 64

/* Geometry of one implicit-GEMM convolution-like gather.  The GEMM M index enumerates a logical grid
 * (N, Dm, Hm, Wm); for tap t = (td, th, tw) the input voxel is  i_x = m_x * in_mult[x] + t_x * tap_step[x] + in_off[x]
 * (zero outside [0, I_x)), and the output voxel is o_x = m_x * out_mult[x] + out_off[x].  This one form covers
 *   nn.Conv3d k4 s2 p1 / k3 s1 p1 / k1        (baseline.py:218-227, :153, :156, :242-244, :258)
 *   nn.ConvTranspose3d k4 s2 p1, as 8 output-parity classes of 2x2x2 taps   (baseline.py:283-293)
 *   their data gradients (conv <-> transposed conv with the channel roles swapped) and nn.Linear (1 tap). */
typedef struct sa_conv_geom {
    int32_t N, Dm, Hm, Wm;          /* logical grid, M = N*Dm*Hm*Wm */
    int32_t Di, Hi, Wi, Cin;        /* input tensor [N,Di,Hi,Wi,Cin]; Cin = channel stride (multiple of 16 bytes) */
    int32_t Do, Ho, Wo, Cout;       /* output tensor [N,Do,Ho,Wo,Cout]; Cout = channel stride */
    int32_t cin_valid, cout_valid;  /* real channel counts (<= strides) */
    int32_t KT[3];                  /* taps per axis (d,h,w) */
    int32_t in_mult[3], tap_step[3], in_off[3];
    int32_t out_mult[3], out_off[3];
    int32_t Kpad;                   /* packed-weight row length in elements: roundup(KT0*KT1*KT2*Cin, 128 bytes) */
    int32_t CoutPad;                /* packed-weight rows: roundup(cout_valid, 128) */
} sa_conv_geom;

/* Fused epilogue of sa_conv_fprop:  v = acc (+bias[co]);  if add_before_act v += addend;  v = act(v);
 * if alpha v *= *alpha;  if !add_before_act v += addend;  v = mask-op(v, mask);  store as out_dtype. */
typedef struct sa_epilogue {
    const float *bias;    /* [CoutPad] or NULL */
    const void *addend;   /* same layout as out, dtype add_dtype, or NULL */
    const void *mask;     /* same layout as out, dtype mask_dtype, or NULL */
    const float *alpha;   /* device scalar or NULL (ReZero gate) */
    int32_t act;          /* SA_ACT_* */
    int32_t mask_mode;    /* SA_MASK_* */
    int32_t add_before_act;
    int32_t out_dtype, add_dtype, mask_dtype;
    float slope;          /* LeakyReLU slope */
    /* optional extra outputs of the same launch (bf16, laid out like out; the LDS-staged epilogue of the im2col-order kernels only, and out_lp also
     * in the register epilogue of SA_F16 launches -- a launch that would take another epilogue returns SA_EUNSUPPORTED):
     *   out_pre: the value BEFORE activation / alpha / addend (acc + bias): the pre-activation a GELU backward needs (nn.Linear -> GELU in one launch),
     *            or the branch output F of a ReZero block x + g F
     *   out_lp : a bf16 copy of the final value (the operand of the next dense layer when out itself is the fp32 residual stream) */
    void *out_pre;
    void *out_lp;
} sa_epilogue;

int sa_abi_version(void);
/* last hipError_t seen by this thread's launches, as text */
const char *sa_last_error(void);
/* kernel instance (rocprofv3 spelling) the calling thread's last sa_conv_fprop / sa_resblock_fprop / sa_conv_wgrad launched:
 * lets a profiler key its per-kernel timings exactly as the dispatcher decided, without mirroring the dispatch rules */
const char *sa_last_conv_kernel(void);
/* Kernel log (test aid; no profiler needed; process-wide, mutex-protected -- autograd issues the backward launches from its own thread):
 * after sa_kernel_log_begin() every launcher of this library notes the name of each kernel it dispatches -- each distinct name once, in first-launch order, spelled as in the source / rocprofv3 (template
 * arguments included for template instances).  sa_kernel_log_read copies the newline-separated list into buf (NUL-terminated, truncated to
 * cap) and returns the bytes needed; stop != 0 ends logging.  Convolution launches are noted with the instance name of sa_last_conv_kernel. */
void sa_kernel_log_begin(void);
int sa_kernel_log_read(char *buf, int cap, int stop);

/* Developer switches that select an alternative kernel for the SAME result (A/B measurements and the cross-family parity tests).  The
 * only process-wide state of the library: one atomic word, initialised ONCE at load time from the environment variables of the same names
 * (SA_NO_HALO=1 ...), never re-read by a launch.  sa_set_debug_flags returns the previous value.  Launches read it without locking: set
 * it while no other thread is inside the library. */
#define SA_DBG_NO_HALO           (1u << 0)   /* 3x3x3 convs on the im2col-order kernels instead of the halo mainloops (fprop, dgrad, wgrad) */
#define SA_DBG_NO_HALO256        (1u << 1)   /* 128-voxel halo tiles only */
#define SA_DBG_NO_HALO256_FUSE   (1u << 2)   /* fused residual block on the 128-voxel halo kernel */
#define SA_DBG_NO_DMA            (1u << 3)   /* register-staged mainloops instead of buffer_load ... lds */
#define SA_DBG_NO_SMALL_TILES    (1u << 4)   /* keep 128x128 tiles for small dense grids */
#define SA_DBG_NO_FUSED_DB       (1u << 5)   /* bias gradient by sa_colsum semantics inside wgrad disabled (separate pass) */
#define SA_DBG_NO_WGRAD_HALO9    (1u << 6)   /* three-tap weight-gradient halo kernel instead of the nine-tap one */
#define SA_DBG_IM2COL_DIRECT     (1u << 7)   /* sa_convt1_im2col without the LDS gather */
#define SA_DBG_SCAN_VALU         (1u << 8)   /* FAVOR+ scans on the VALU segment kernels */
#define SA_DBG_LOCAL_ATTN_EXACT  (1u << 9)   /* local attention on the exact-fp32 MFMA kernels */
#define SA_DBG_HALO256_4W        (1u << 13)  /* bf16 im2col-order forward / data-gradient and weight-gradient kernels with four waves per block instead of eight */
#define SA_DBG_RESERVED_14       (1u << 14)  /* (was SA_DBG_TILE256: 256 x 128 tiles for the im2col-order kernel -- measured slower, instance removed) */
#define SA_DBG_DENSE_NARROW      (1u << 15)  /* A/B: 128 x 64 tiles for every small dense grid (the round-2 rule) */
#define SA_DBG_NO_KGROUPS        (1u << 17)  /* dense layers with about one tile per CU on the one-group kernel instead of two K groups of eight waves (conv_fprop_dma_kernel<..., 2>) */
#define SA_DBG_RESERVED_19      (1u << 19)  /* (was SA_DBG_CELLS: the 128-voxel cell mainloop conv_fprop_cells_kernel -- a tie with im2col order in round 4, superseded by cells256, removed in round 6) */
#define SA_DBG_RESERVED_20      (1u << 20)  /* (was SA_DBG_DENSE_RING: the four-wave ring GEMM of dense.hip -- slower on 7 of the 8 Performer shapes in round 5, removed in round 6) */
#define SA_DBG_NO_CELLS256       (1u << 21)  /* stride-2 family on the im2col-order kernel instead of the 256-voxel cell mainloop (conv_fprop_cells256_kernel): A/B + cross-family tests */
#define SA_DBG_NO_CLASS_LAUNCH    (1u << 22)  /* sa_conv_fprop_classes answers SA_EUNSUPPORTED: the parity classes of a transposed convolution as separate launches (A/B, equality test) */
#define SA_DBG_FAVOR_SEQ_ALWAYS  (1u << 18)  /* FAVOR+ chunk states in the sequential form for every batch (default: from 40 (batch, head) pairs; tests) */
#define SA_DBG_DETERMINISTIC     (1u << 16)  /* fixed-order reductions where the library itself chooses (BatchNorm sums); see the deterministic-mode section */
#define SA_DBG_SCAN_EXACT_SHIFT  10          /* 3 bits: chunk states | scan A outputs | scan B outputs on the exact-fp32 MFMA kernels */
/* measurement aid (bench.py `roofline.peak_measured`): `blocks` x 4 waves each issue iters x 8 independent v_mfma_f32_32x32x16_bf16;
 * FLOPs per call = blocks * 4 * iters * 8 * 32768.  `scratch` = any 4 device bytes (never written in practice). */
int sa_bench_mfma_bf16(float *scratch, int blocks, int iters, void *stream);
/* the same probe at other occupancies / instruction shapes: `threads` per block (64 .. 1024), four accumulator tiles per wave, shape 0 =
 * v_mfma_f32_32x32x16_bf16 (32 768 FLOP each), 1 = v_mfma_f32_16x16x32_bf16 (16 384): FLOPs per call = blocks * threads/64 * iters * 4 * (32 768 | 16 384) */
int sa_bench_mfma_bf16_ex(float *scratch, int blocks, int threads, int iters, int shape, void *stream);
uint32_t sa_get_debug_flags(void);
uint32_t sa_set_debug_flags(uint32_t flags);

/* ---- weights: reference layout (fp32 nn.Parameter) -> packed [CoutPad][Kpad] GEMM operand ------------------------
 * element (row r, reduce channel c, tap t) is read at  w[r*s_row + c*s_red + tap_lut[t]]  (tap_lut NULL = identity).
 * Conv3d weight [Co,Ci,k,k,k] (baseline.py:218): s_row=Ci*T, s_red=T.  ConvTranspose3d weight [Ci,Co,k,k,k]
 * (baseline.py:283): s_row=T, s_red=Co*T.  Swapping the two gives the data-gradient operand. */
int sa_pack_weights(const float *w, void *wpk, int dtype, int rows, int red, int ntaps, const int32_t *tap_lut_host,
                    int64_t s_row, int64_t s_red, int rows_pad, int red_stride, int Kpad, void *stream);
/* The same for many operands in ONE launch (a training step re-packs every layer's forward and data-gradient operand after the optimizer
 * step: ~300 launches of a few microseconds each otherwise).  `table` is a DEVICE array of n descriptors and `block_first` a DEVICE array of
 * n + 1 ints: descriptor i is packed by blocks [block_first[i], block_first[i+1]); the caller uploads both once -- parameters and packed
 * operands keep their addresses across steps.  tap_lut holds ntaps entries (identity: 0, 1, 2, ...). */
typedef struct sa_pack_desc {
  const float *w;
  void *wpk;
  int32_t tap_lut[SA_MAX_TAPS];
  int64_t s_row, s_red;
  int32_t dtype, rows, red, ntaps, rows_pad, red_stride, Kpad, reserved;
} sa_pack_desc;
int sa_pack_weights_batch(const sa_pack_desc *table, const int32_t *block_first, int n, int total_blocks, void *stream);

/* ---- convolution forward / data gradient (implicit GEMM on MFMA) -- replaces cuDNN behind nn.Conv3d /
 * nn.ConvTranspose3d / nn.Linear at baseline.py:153-160,218-244,258-293; discriminator/baseline.py:41-80;
 * performer_pytorch to_q/to_k/to_v/to_out/FeedForward (performer.py:194-219) and performer.py:221 to_out.
 * dtype = the operand type of `in` and `wpk`: SA_F32 (exact-f32 MFMA), SA_BF16, or SA_F16 -- IEEE halves, the reference's AMP forward dtype
 * (src/engines/trainer.py:161-163), for FORWARD launches with DMA-addressable operands (< 4 GiB each): 16-bit addends / outputs of an SA_F16
 * launch are halves too (ep->add_dtype / out_dtype = SA_F16, or SA_F32), masks are never halves, and ep->out_lp receives a bf16 copy of the
 * output from either epilogue (what the bf16 backward pass of the next layer reads).  Mixing 16-bit types otherwise -> SA_EUNSUPPORTED. */
int sa_conv_fprop(const sa_conv_geom *g, int dtype, const void *in, const void *wpk, void *out, const sa_epilogue *ep,
                  void *stream);
/* n launch geometries of ONE layer that differ only in in_off / out_off, with their packed operands, in one launch: the eight output-parity classes of
 * nn.ConvTranspose3d k4 s2 p1 (baseline.py:283-293) and of the data gradient of the strided nn.Conv3d (baseline.py:218-227).  SA_EUNSUPPORTED (nothing
 * launched) when they differ in anything else, n > 8, or the kernel the dispatcher picks does not take classes (fp32, operands >= 4 GiB): the caller then
 * issues sa_conv_fprop per geometry. */
int sa_conv_fprop_classes(const sa_conv_geom *geoms, int n, int dtype, const void *in, const void *const *wpks, void *out, const sa_epilogue *ep,
                          void *stream);

/* ---- ResidualLayer forward in ONE launch (baseline.py:150-160), bf16 or f16 (dtype = type of x, the packed weights and y; addend / out dtypes of
 * ep must equal it; h_out is ALWAYS bf16 -- it is an operand of the bf16 backward pass -- and ep->out_lp takes a bf16 copy of an f16 y), 128 channels, k3 s1 p1 geometry `g`:
 *   h = relu(conv3x3x3(x) + bias1)  (stored to h_out when non-NULL: the backward pass needs it)
 *   y = epilogue(h . w1pk^T)  with ep = {bias = b2, addend = x, add_before_act = 1, act = RELU} for the reference block.
 * w3pk / w1pk are sa_pack_weights operands ([128][27*128] and [128][128]).  SA_EUNSUPPORTED for other shapes: use two sa_conv_fprop. */
int sa_resblock_fprop(const sa_conv_geom *g, int dtype, const void *x, const void *w3pk, const float *bias1, const void *w1pk, void *h_out,
                      void *y_out, const sa_epilogue *ep, void *stream);

/* ---- weight gradient: dw[r*s_row + c*s_red + tap_lut[t]] += sum_m in[gather(m,t)][c] * gout[out(m)][r]  (accumulates:
 * caller zeroes dw), and, when db is non-NULL, the bias gradient db[r] += sum_m gout[m][r] (summed from the gradient tiles the
 * kernel stages anyway; sa_colsum is the stand-alone form).  Replaces cuDNN wgrad / bias-grad behind the same modules' autograd. */
int sa_conv_wgrad(const sa_conv_geom *g, int dtype, const void *in, const void *gout, float *dw, float *db, const int32_t *tap_lut_host,
                  int64_t s_row, int64_t s_red, void *workspace, int64_t workspace_bytes, void *stream);
/* Backward of a 1x1x1, 128 -> 128 channel convolution whose INPUT is a post-ReLU tensor (second convolution of the residual block,
 * baseline.py:150-160) in ONE launch: dw / db as sa_conv_wgrad, and dx[m][ci] = (in[m][ci] > 0) * sum_co gout[m][co] W[co][ci] from the same
 * staged tiles.  dgrad_wpk = sa_pack_weights operand of the layer's data-gradient plan [128 ci][128 co].  bf16 only; workspace as for
 * sa_conv_wgrad (required).  SA_EUNSUPPORTED for anything else: use sa_conv_wgrad + sa_conv_fprop. */
int sa_conv1x1_backward(const sa_conv_geom *g, int dtype, const void *in, const void *gout, float *dw, float *db, int64_t s_row, int64_t s_red,
                        void *workspace, int64_t workspace_bytes, const void *dgrad_wpk, void *dx, void *stream);
/* bytes of scratch sa_conv_wgrad wants for this geometry (partial tiles of the voxel splits; a second kernel reduces them
 * without atomics).  With workspace == NULL the kernel falls back to fp32 atomics straight into dw. */
int64_t sa_conv_wgrad_workspace_bytes(const sa_conv_geom *g, int dtype);

/* db[c] += sum_m g[m][c]   (bias gradient), g is [M][cstride] of dtype */
int sa_colsum(const void *g, int dtype, int64_t M, int C, int cstride, float *db, void *stream);

/* ---- EMA vector quantizer -- replaces Quantizer_impl.forward (baseline.py:38-87) ---------------------------------
 * rows [M,D] fp32 (already channels-last = the reference's flat_inputs), codebook [K,D] fp32 (pre-update).
 * Writes idx[M] (int64, first-minimum tie-break), zq_st[M,D] = (W[idx]-x)+x, optional zq_lp (bf16 copy),
 * and accumulates counts[K], dw[K,D], sqerr[1] = sum (W[idx]-x)^2 (fp32 atomics; caller zeroes them).
 * wnorm[K] is scratch for |W_k|^2. */
int sa_vq_assign(const float *rows, const float *codebook, int64_t M, int K, int D, int64_t *idx, float *zq_st, void *zq_lp,
                 float *counts, float *dw, float *sqerr, float *wnorm, void *stream);
/* baseline.py:75-80: N <- g N + (1-g) counts; embed_avg <- g embed_avg + (1-g) dw; codebook <- embed_avg / smoothed N */
int sa_vq_ema_update(float *N, float *embed_avg, float *codebook, const float *counts, const float *dw, int K, int D,
                     float decay, float eps, void *stream);
/* baseline.py:110-120: perplexity = exp(-sum p log(p+1e-10)), p = counts / M */
int sa_vq_perplexity(const float *counts, int K, int64_t M, float *out, void *stream);
/* gradient of (zq_st, loss) wrt the encoder output: dz = g_zq + g_loss * beta * 2 (x - W[idx]) / (M D)   (baseline.py:82-85) */
int sa_vq_backward(const float *rows, const float *codebook, const int64_t *idx, const void *g_zq, int g_dtype,
                   const float *g_loss, float beta, int64_t M, int D, void *dz, int dz_dtype, void *stream);
/* codebook lookup  out[m] = W[idx[m]]  (Quantizer_impl.embed, baseline.py:89-91) */
int sa_vq_embed(const float *codebook, const int64_t *idx, int64_t M, int K, int D, void *out, int out_dtype, void *stream);

/* ---- small fused elementwise / reduction kernels ---------------------------------------------------------------- */
/* dst[i*dst_stride + c] = (c < src_c) ? src[i*src_c + c] : 0,  i < rows  (dtype conversion + channel padding) */
int sa_cast_pad(const void *src, int src_dtype, int src_c, void *dst, int dst_dtype, int dst_stride, int64_t rows, void *stream);
/* use_subpixel_conv=True (reference src/networks/vqvae/baseline.py:274-282: the last decoder layer is MONAI's SubpixelUpsample(3, n_channels // 2, 1, scale_factor=2,
 * apply_pad_pool=True)): the tail behind its conv_block.  c [N, D, H, W, 8] fp32 (channel = (fd*2 + fh)*2 + fw) -> pixelshuffle -> ConstantPad3d((1, 0) x 3) ->
 * AvgPool3d(2, stride 1) -> out [N, 2D, 2H, 2W] fp32; _bwd: the adjoint, g [N, 2D, 2H, 2W] fp32 -> dc [N, D, H, W, 8] in dc_dtype (SA_F32 / SA_BF16). */
int sa_subpixel_pool_fwd(const float *c, float *out, int N, int D, int H, int W, void *stream);
int sa_subpixel_pool_bwd(const float *g, void *dc, int dc_dtype, int N, int D, int H, int W, void *stream);
/* MSELoss (losses/vqvae/vqvae.py:14-71): loss_sum[0] += sum (a-b)^2 ; grad = (a-b) * (2*gscale/n) if grad != NULL */
int sa_mse(const float *a, const float *b, int64_t n, float *loss_sum, float *grad, float gscale, void *stream);
/* Adam (torch.optim.Adam semantics, run_vqvae.py:82-86) over a flat fp32 parameter buffer; step >= 1 */
int sa_adam(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
            float weight_decay, int step, float grad_scale, void *stream);

/* ==== Performer path (reference src/networks/transformers/performer.py:194-221,229-288 -> third-party performer-pytorch
 * 1.0.11 / local-attention / fast_transformers CausalDotProduct; torch embedding, LayerNorm, cross_entropy).  All fp32.
 * The dense projections (to_q/to_k/to_v/to_out, FeedForward w1/w2, final to_out) are 1-tap sa_conv_fprop / sa_conv_wgrad. */

/* out[r,:] = sum_t table_t[idx_t[per_position_t ? r % N : r], :]  (idx < 0 skips) -- token + spatial + positional embeddings
 * (performer.py:241-266); sa_embed_scatter is its gradient wrt one table (fp32 atomics). */
int sa_embed_sum(int ntab, const float *const *tables, const int64_t *const *idx, const int32_t *per_position, int dim, int N,
                 int64_t R, float *out, void *stream);
int sa_embed_scatter(const float *dy, float *dtable, const int64_t *idx, int per_position, int dim, int N, int64_t R, void *stream);

/* ---- stateful O(N) decoding (replaces the O(N^2) loop of TransformerBase.sample, src/networks/transformers/transformer.py:58-101, which
 * runs a full forward over the growing prefix per token).  One new position per call; `pos` is a DEVICE int (the index of the position
 * being produced) so that a whole per-token step can be captured in a HIP graph and replayed.  Equal to the O(N^2) loop up to fp32
 * rounding: the FAVOR+ key stabiliser is the running maximum of the prefix (the keys' global max of performer-pytorch 1.0.11) and the
 * state keeps the exp part and the +eps part of phi(k) apart, so it can be rescaled when that maximum grows.
 * sa_embed_step : out[b,:] = sum_t table_t[per_position_t ? idx_t[*pos] : idx_t[b], :]   (idx < 0 skips)
 * sa_favor_step : global heads.  proj [m, dh] = projection matrix with the data normaliser dh^-1/4 folded in; state smax[2] (both -inf at
 *                 the start), kmax[2] (int; the ordered encoding of -inf, 0x807fffff, at the start), E [B*G, LDF, dh], Ez [B*G, LDF],
 *                 V1 [B*G, dh] (0 at the start); dd [2, B*G, LDF] scratch; writes the attention rows of position *pos.
 * sa_local_attn_step : local heads.  Rotates q_t / k_t with row *pos of the rotary tables, appends (k_t, v_t) to the caches
 *                 [B, L, N, dh] and attends over the previous and the current window up to *pos (look_backward = 1, causal). */
int sa_embed_step(int ntab, const float *const *tables, const int64_t *const *idx, const int32_t *per_position, int dim, const int *pos, int B,
                  float *out, void *stream);
/* The decision of one decode step for all B rows (TransformerBase.sample_next_index, transformer.py:11-17,19-56) + the sequence update:
 * logits [B, V] / temperature -> top-k cut (top_k in (0, V): everything below the k-th largest value of a row is masked, ties with it stay; <= 0 or >= V: none) -> softmax -> categorical draw by inverse CDF with the caller's uniforms u[*pos * u_stride + b] (do_sample; u_stride = B: a table
 * drawn once per sample() call, 0: one vector per step) or arg-max; seq[b, *pos + 1] receives the token unless that position belongs to the given prefix (< P);
 * tok[b] = seq[b, *pos + 1] (what the next step embeds); *pos += 1.  ticket: NULL (one block walks the rows) or a zeroed int the launch leaves zeroed (one
 * block per row; the last one to finish advances *pos). */
int sa_sample_step(const float *logits, int B, int V, float temperature, const float *u, int u_stride, int do_sample, int top_k, int64_t *seq, int total, int P,
                   int *pos, int *ticket, int64_t *tok, void *stream);
int sa_favor_step(const float *q, int q_stride, int q_off, const float *k, int k_stride, int k_off, const float *v, int v_stride, int v_off,
                  const float *proj, int B, int G, int dh, int m, int LDF, float *smax, int *kmax, float *dd, float *E, float *Ez, float *V1,
                  const int *pos, float *out, int out_stride, int out_off, void *stream);
int sa_local_attn_step(const float *q, int q_stride, int q_off, const float *k, int k_stride, int k_off, const float *v, int v_stride, int v_off,
                       const float *cosb, const float *sinb, float *kcache, float *vcache, const int *pos, int B, int N, int L, int W, int dh,
                       float *out, int out_stride, int out_off, void *stream);
/* sa_favor_step + sa_local_attn_step of one layer in TWO launches when it has both kinds of heads (q | k | v = column blocks of `qkv`, row stride
 * `stride`, inner = (G + L) * dh, global heads first): [projections | local heads over four key segments] then [FAVOR+ update | combine].  part: B * L * 4 * 66
 * floats of scratch.  Equal to the two calls up to the summation order of the local softmax. */
int sa_attn_step(const float *qkv, int stride, int inner, const float *proj, int B, int G, int L, int dh, int m, int LDF, float *smax, int *kmax, float *dd,
                 float *E, float *Ez, float *V1, const float *cosb, const float *sinb, float *kcache, float *vcache, int N, int W, float *part,
                 const int *pos, float *out, int out_stride, void *stream);
/* small-batch dense layer of the decode step (B <= 32 rows; a stream over the fp32 nn.Linear weights, up to three tensors concatenated
 * along the outputs, e.g. q | k | v): y[b][o] = epi(sum_i x[b][i] W[o][i] + bias[o]); act 0 none / 1 GELU; then y = res + gate * y when
 * res is given (gate: device scalar or NULL = 1).  round_in / round_w / round_out reproduce the bf16 operand / output rounding of the
 * MFMA path; round_w = 2: the `w` tensors already HOLD bf16 values ([out][in] bf16 copies of the parameters: half the bytes per token). */
int sa_gemv_rows(const float *x, int x_stride, int in, int B, int nseg, const float *const *w, const float *const *bias, const int32_t *seg_out,
                 float *y, int y_stride, int act, const float *res, int res_stride, const float *gate, int round_in, int round_w, int round_out,
                 void *stream);
/* nn.LayerNorm (performer.py:220,273); stats[2r] = mean, stats[2r+1] = rstd; y_lp optional copy in lp_dtype */
int sa_layernorm_fwd(const float *x, const float *w, const float *b, float *y, void *y_lp, int lp_dtype, float *stats, int64_t R, int C,
                     float eps, void *stream);
int sa_layernorm_bwd(const float *dy, const float *x, const float *w, const float *stats, float *dx, float *dw, float *db, int64_t R,
                     int C, void *stream);
/* FeedForward activation (exact erf GELU) */
int sa_gelu(const void *u, int u_dtype, void *h, int h_dtype, int64_t n, void *stream);
/* ReZero residual: y = x + g*F ; backward dF = g*dy, dg += sum dy*F */
int sa_rezero_fwd(const float *x, const void *F, int f_dtype, const float *g, float *y, void *y_lp, int lp_dtype, int64_t n, void *stream);
int sa_rezero_bwd(const float *dy, const void *F, int f_dtype, const float *g, void *dF, int df_dtype, float *dg, int64_t n, void *stream);
int sa_axpy(float *y, const float *x, float alpha, int64_t n, void *stream);
/* FAVOR+ softmax_kernel feature map on top of the projection GEMM output dd [rows, LDF] (rows = B*N*G):
 * feat = m^-1/2 (exp(dd - |x|^2 d^-1/2 / 2 - stab) + 1e-4), stab = row max (is_query = 1) or the GLOBAL max (keys; gmax_ws = 8 bytes:
 * is_query = 0 computes it here, is_query = 2 takes it as left there by sa_favor_project). */
int sa_favor_features_fwd(const float *dd, const float *src, int src_stride, int h0, int G, int dh, int is_query, float *feat, void *gmax_ws,
                          int64_t rows, int m, int LDF, void *stream);
/* backward: ddd = d loss / d dd; dsrc (same stride / head offset as src) is OVERWRITTEN with the gradient through the -|x|^2 term, the
 * projection adjoint (sa_favor_project_bwd with addend = dx, or a dgrad GEMM with an addend) adds the rest */
int sa_favor_features_bwd(const float *dfeat, const float *feat, const float *dd, const float *src, int src_stride, int h0, int G, int dh,
                          int is_query, float *ddd, float *dsrc, const void *gmax_ws, float *tsum_ws /* [rows] */, int64_t rows, int m, int LDF,
                          void *stream);
/* FastAttention.redraw_projection_matrix for nmat layers at once: out[nmat,m,d] = rowwise-orthonormalised Gaussian blocks
 * [nmat,nblk,d,d] scaled by |rows[nmat,m,d]| */
int sa_favor_projection(const float *blocks, const float *rows, float *out, int nmat, int nblk, int m, int d, void *stream);
/* causal running-state scans replacing fast_transformers' CausalDotProduct (forward and both backward directions):
 *   scan_a: T[m][d] += a_i[m] b_i[d] ; y_i[d] = (sum_m c_i[m] T[m][d]) * y_scale_i        (a, c: [B,N,G,LDF]; b, y: strided head blocks)
 *   scan_b: T[m][d] += a_i[m] b_i[d] ; y_i[m] = sum_d T[m][d] c_i[d] + ex_scale_i (ex_vec_i[m] + ex_const)   (y: [B,N,G,LDF])
 * state_ws (sa_favor_scan_workspace_bytes; NULL = one block per (b, g) walks all N positions) lets <= 16 segments of ~128
 * positions be scanned by independent blocks: segment state sums -> exclusive prefix -> segment scans from the prefix. */
int64_t sa_favor_scan_workspace_bytes(int B, int N, int G, int LDF, int dv);
int sa_favor_scan_a(const float *a, const float *c, const float *b, int b_stride, int b_off, const float *b_scale, float *y, int y_stride,
                    int y_off, const float *y_scale, int B, int N, int G, int LDF, int dv, int reverse, int accumulate, float *state_ws,
                    void *stream);
int sa_favor_scan_b(const float *a, const float *b, int b_stride, int b_off, const float *b_scale, const float *c, int c_stride, int c_off,
                    const float *c_scale, float *y, const float *ex_scale, const float *ex_vec, float ex_const, int B, int N, int G,
                    int LDF, int dv, int reverse, float *state_ws, void *stream);
/* The same scans with the column running sums fused (one extra state column on the chunked MFMA path; SA_EUNSUPPORTED elsewhere):
 * sa_favor_scan_a_norm: y_i = (sum_{j<=i} (c_i . a_j) b_j) / (c_i . (sum_{j<=i} a_j + den_eps)), inv_out[i] = 1 / denominator
 *                       (= sa_cumsum_rows + sa_favor_den + sa_favor_scan_a with y_scale = inv).
 * sa_favor_scan_b_cum : ex_mode 1: y_i[m] += ex_scale_i * (sum_{j<=i} a_j[m] + ex_const);  ex_mode 2: y_i[m] += sum_{j<=i} a_j[m] ex_scale_j
 *                       (j <= i in scan order; = sa_cumsum_rows + sa_favor_scan_b with ex_vec). */
int sa_favor_scan_a_norm(const float *a, const float *c, const float *b, int b_stride, int b_off, float *y, int y_stride, int y_off, float *inv_out,
                         float den_eps, int B, int N, int G, int LDF, int dv, float *state_ws, int state_flags, void *stream);
int sa_favor_scan_b_cum(const float *a, const float *b, int b_stride, int b_off, const float *b_scale, const float *c, int c_stride, int c_off,
                        const float *c_scale, float *y, const float *ex_scale, int ex_mode, float ex_const, int B, int N, int G, int LDF, int dv,
                        int reverse, float *state_ws, int state_flags, void *stream);
/* state_flags bit 0: state_ws already holds the exclusive chunk prefixes of exactly this (a, b, b_scale, reverse) -- written by an earlier
 * scan on the same operands -- so the state and prefix passes are skipped; bit 1 (sa_favor_scan_a_state): the buffer has the extra
 * running-sum column.  Forward and dq' share one state set, dk' and dv another: two state passes per head instead of four.
 * bit 2: evaluate every product on the exact-fp32 MFMA.  Default is split-bf16 (x = hi + lo in bf16, hi*hi + hi*lo + lo*hi with fp32
 * accumulation, ~1e-5 relative): plenty next to bf16 dense layers, but the query-side gradients cancel to ~1e-3 of their terms, so the
 * fp32 parity mode of the Python engine sets the bit.  (Environment SA_SCAN_EXACT=7 forces it for every entry point.) */
int sa_favor_scan_a_state(const float *a, const float *c, const float *b, int b_stride, int b_off, const float *b_scale, float *y, int y_stride,
                          int y_off, const float *y_scale, int B, int N, int G, int LDF, int dv, int reverse, int accumulate, float *state_ws,
                          int state_flags, void *stream);
/* FAVOR+ random-feature projection (performer_pytorch softmax_kernel: data_dash = data_normalizer * data @ projection^T) and its adjoint as
 * HBM-bound kernels; proj [m][dh] already carries the data normalizer.  Row r of x / dx is head block r % heads of the wider row r / heads:
 * it starts at (r / heads) * stride + (r % heads) * dh floats (heads = 1: plain rows).  dd [rows][LDF] (columns >= m are written as zeros);
 * gmax_ws (8 bytes, optional): the global (value, index) maximum of dd that the key feature map needs, taken from the accumulators --
 * pass it to sa_favor_features_fwd with is_query = 2 and the separate pass over dd is skipped;
 * dx = ddd @ proj (+ addend, laid out like dx; addend == dx is allowed).  dh = 64, LDF % 16 == 0, LDF <= 272.  Products are split-bf16 (~1e-5 relative); the exact
 * alternative is sa_conv_fprop on the same operands as a 1x1x1 convolution. */
int sa_favor_project(const float *x, int x_stride, int heads, const float *proj, float *dd, void *gmax_ws, int64_t rows, int m, int LDF, int dh,
                     void *stream);
/* sa_favor_project + the QUERY feature map (sa_favor_features_fwd with is_query = 1) in one launch: dd and feat [rows][LDF] */
int sa_favor_project_features(const float *x, int x_stride, int heads, const float *proj, float *dd, float *feat, int64_t rows, int m, int LDF, int dh,
                              void *stream);
int sa_favor_project_bwd(const float *ddd, const float *proj, const float *addend, float *dx, int dx_stride, int heads, int64_t rows, int m, int LDF,
                         int dh, void *stream);
/* sa_favor_features_bwd + sa_favor_project_bwd in one launch (+ one fix-up launch for keys): the intermediate d loss / d dd is never written.
 * src / dsrc rows are head blocks as for sa_favor_project (stride, heads); proj carries the data normalizer; dsrc is overwritten.
 * tsum_ws (keys): at least one float of scratch (tsum_ws[0] = sum over the rows of the stabiliser's share, for the fix-up launch). */
int sa_favor_features_project_bwd(const float *dfeat, const float *feat, const float *dd, const float *src, int src_stride, int heads, const float *proj,
                                  int is_query, float *dsrc, const void *gmax_ws, float *tsum_ws, int64_t rows, int m, int LDF, int dh, void *stream);
int sa_cumsum_rows(const float *x, const float *scale, float *out, int B, int N, int G, int LDF, int reverse, float *seg_ws, void *stream);
int sa_favor_den(const float *q, const float *z, float eps, float *inv, int64_t rows, int m, int LDF, void *stream);
int sa_favor_dden(const float *dout, const float *out, int stride, int off, int G, int dv, const float *inv, float *dden, int64_t rows,
                  void *stream);
/* rotary embedding of the local heads (local-attention >= 1.2); transpose=1 applies the adjoint (backward) */
int sa_rotary(const float *x, int stride, int off, int L, int dh, const float *cosb, const float *sinb, float *y, int y_stride, int y_off,
              int N, int64_t R, int transpose, int accumulate, void *stream);
/* the same rotation for `ngroups` operands in one launch (q and k of a layer): operand gi is read x_goff and written y_goff ELEMENTS behind operand 0 */
int sa_rotary_groups(const float *x, int stride, int off, int L, int dh, const float *cosb, const float *sinb, float *y, int y_stride, int y_off,
                     int N, int64_t R, int transpose, int accumulate, int ngroups, int64_t x_goff, int64_t y_goff, void *y_lp, void *stream);
/* rotary embedding of the GLOBAL heads -- the wrapper's rotary_position_emb=True (reference src/networks/transformers/performer.py:134-137,246: layer_pos_emb handed
 * to every performer_pytorch SelfAttention, whose apply_rotary_pos_emb rotates q / k of the FAVOR+ heads): pairs of consecutive dimensions (2i, 2i + 1) of L head
 * rows of width dh at column `off` of x [R, stride] are rotated by the angle of columns i (sine) and dh/2 + i (cosine) of row (r mod N) of `sincos` [N, dh];
 * transpose=1 applies the adjoint; `ngroups` operands x_goff / y_goff ELEMENTS apart (q and k) in one launch; y may alias x. */
int sa_rotary_pairs(const float *x, int stride, int off, int L, int dh, const float *sincos, float *y, int y_stride, int y_off, int N, int64_t R, int transpose,
                    int ngroups, int64_t x_goff, int64_t y_goff, void *stream);
/* "_lp" outputs (here and in sa_local_attn_* / sa_favor_fused_*): optional (NULL = none) bf16 mirror of an fp32 output matrix -- every element written
 * to the fp32 rows is also written, rounded to nearest even, at the SAME element offset (strides and offsets in elements) of the bf16 buffer.  It is the
 * operand the next dense layer consumes, so the stand-alone fp32 -> bf16 cast launch (and its read of the fp32 matrix) disappears. */
/* causal local-window attention (window W, look back one window): per query softmax over keys [max(0,(n/W-1)W), n].
 * fp32 in / out; products are evaluated as split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate; ~1e-5 relative) unless the environment
 * has SA_LOCAL_ATTN_EXACT=1 (exact-fp32 MFMA).  N * max(stride) * 4 must stay below 2^31 (SA_EUNSUPPORTED otherwise). */
int sa_local_attn_fwd(const float *q, int q_stride, int q_off, const float *k, int k_stride, int k_off, const float *v, int v_stride, int v_off,
                      float *o, int o_stride, int o_off, float *lse, int B, int N, int L, int W, int dh, void *o_lp, void *stream);
/* dq/dk/dv use the strides and offsets of q/k/v */
int sa_local_attn_bwd(const float *q, int q_stride, int q_off, const float *k, int k_stride, int k_off, const float *v, int v_stride, int v_off,
                      const float *out, const float *dout, int o_stride, int o_off, const float *lse, float *dq, float *dk, float *dv,
                      float *Dbuf, int B, int N, int L, int W, int dh, void *dv_lp, void *stream);
/* CELoss (losses/transformer/transformer.py:24-33): loss_sum += sum_r (lse_r - logit[r,target_r]); dlogits = (softmax - onehot) * gscale */
int sa_cross_entropy(const float *logits, const int64_t *target, int64_t R, int V, float *loss_sum, void *dlogits, int d_dtype, float gscale,
                     void *stream);

/* ==== discriminator normalisation (reference src/networks/discriminator/baseline.py:52-79: nn.BatchNorm3d + LeakyReLU) ====
 * x, y, g, dx: channels-last [M, C] of dtype.  training: batch statistics (+ running-stat update, momentum, unbiased variance);
 * eval: running statistics.  mean/rstd [C] are outputs kept for the backward; sums_ws is scratch: 2*C floats, and (1 + 64)*2*C floats when the
 * SA_DBG_DETERMINISTIC flag is set (64 per-block partial sums per statistic, added in block order: sa_bn_sums_ws_floats(C) returns the size). */
int64_t sa_bn_sums_ws_floats(int C);
int sa_bn_forward(const void *x, int dtype, int64_t M, int C, const float *w, const float *b, float *running_mean, float *running_var,
                  float momentum, float eps, int training, float slope, void *y, float *mean, float *rstd, float *sums_ws, void *stream);
int sa_bn_backward(const void *x, const void *g, int dtype, int64_t M, int C, const float *w, const float *mean, const float *rstd,
                   int training, void *dx, float *dw, float *db, float *sums_ws, void *stream);
/* g = dy * (y > 0 ? 1 : slope) */
int sa_lrelu_mask(const void *dy, const void *y, int dtype, void *g, int64_t n, float slope, void *stream);

/* ==== first encoder layer nn.Conv3d(1 -> 128, k4 s2 p1) (+ReLU) (baseline.py:218-226, level 0), bf16: the taps are gathered straight from
 * the fp32 volume x [N,2D,2H,2W] into LDS (no channel padding, no im2col matrix in HBM).  (N, D, H, W) is the OUTPUT grid.
 * sa_conv1_fwd  : y [N,D,H,W,128] bf16 = act(bias + W x);  wpk = the layer's weight [128][64 taps] as bf16 (sa_pack_weights with rows = 128,
 *                 red = 64, one tap);  act = SA_ACT_NONE / SA_ACT_RELU.
 * sa_conv1_wgrad: dw [128][64] += g^T x_taps, db [128] += sum g  (g [N,D,H,W,128] bf16; accumulated with fp32 atomics: zero them first).
 * cout != 128 -> SA_EUNSUPPORTED: use sa_convt1_im2col + sa_conv_fprop / sa_conv_wgrad on the [cells][64] matrix. */
int sa_conv1_fwd(const float *x, const void *wpk, const float *bias, void *y, int N, int D, int H, int W, int cout, int act, void *stream);
/* sa_conv1_fwd_f16: the same with IEEE-half taps, weights (wpk packed as SA_F16) and output y -- the first layer of an f16 forward chain -- and an
 * optional bf16 copy y_lp of y (what the bf16 backward pass of the NEXT layer reads). */
int sa_conv1_fwd_f16(const float *x, const void *wpk, const float *bias, void *y, void *y_lp, int N, int D, int H, int W, int cout, int act, void *stream);
int sa_conv1_wgrad(const float *x, const void *g, float *dw, float *db, int N, int D, int H, int W, int cout, void *stream);
/* Backward of the last decoder layer nn.ConvTranspose3d(128 -> 1, k4 s2 p1) (baseline.py:283-293, last level), bf16, on the two kernels above:
 * g = d loss / d output [N,2D,2H,2W] fp32, x = the layer input [N,D,H,W,128] bf16, wpk = the transposed-convolution weight [128][64 taps] as bf16
 * (the operand sa_conv1_fwd takes).  dx [N,D,H,W,128] bf16 = sum_t g[2 cell - 1 + t] W[c][t] (zeroed where x <= 0 when mask_input),
 * dw [128][64] += sum_cells x[cell][c] g[2 cell - 1 + t], db [1] += sum g (fp32 atomics: zero dw / db first). */
/* Forward of the same layer in ONE launch (bf16 input [N,D,H,W,128], fp32 output [N,2D,2H,2W]): the per-cell tap products stay in LDS (two rolling depth
 * planes of an 8 x 8-cell patch + halo), no [cells][64] matrix in HBM.  wpk = the weight as [64 taps][128 channels] bf16 (row stride 128). */
int sa_convt1_fused_fwd(const void *x, const void *wpk, const float *bias, float *out, int N, int D, int H, int W, void *stream);
int sa_convt1_backward(const float *g, const void *x, const void *wpk, int mask_input, void *dx, float *dw, float *db, int N, int D, int H, int W,
                       void *stream);

/* ==== final decoder layer nn.ConvTranspose3d(128 -> 1, k4 s2 p1) (baseline.py:283-293, last level): HBM-bound direct kernels ====
 * x [N,D,H,W,128] (dtype), w [128][64] fp32 (the reference weight [Cin,1,4,4,4]), out / g [N,2D,2H,2W] fp32.
 * sa_convt1_bwd: dx = dgrad * (relu_mask > 0) (dx may be NULL), dw += wgrad, db += sum g. */
int sa_convt1_fwd(const void *x, int dtype, const float *w, const float *bias, float *out, int N, int D, int H, int W, int C, void *stream);
int sa_convt1_bwd(const void *x, int dtype, const float *w, const float *g, const void *relu_mask, void *dx, float *dw, float *db, int N,
                  int D, int H, int W, int C, void *stream);
/* GEMM route of the same layer: the 64 taps are the channels of a 1x1x1 convolution run by sa_conv_fprop / sa_conv_wgrad, and these
 * two kernels move between the output voxel grid and the [cell][64 tap] matrices.
 * sa_convt1_gather: out[o] = bias + the 8 entries of P [cells][64] (fp32) that feed output voxel o.
 * sa_convt1_im2col: Gc[cell][tap] = g[2 cell - 1 + tap] (zero outside), stored as `dtype`; db += sum g (db may be NULL). */
int sa_convt1_gather(const float *p, const float *bias, float *out, int N, int D, int H, int W, void *stream);
int sa_convt1_im2col(const float *g, int dtype, void *gc, float *db, int N, int D, int H, int W, void *stream);

/* ---- FAVOR+ global heads with the feature maps recomputed on chip (csrc/favor_fused.hip; throughput mode) ------------------------------------
 * Replaces, for performer_pytorch.FastAttention / softmax_kernel / causal_linear_attention (reference src/networks/transformers/performer.py:194-219
 * delegates to them), the chain sa_favor_project* -> sa_favor_features_fwd -> sa_favor_scan_a_norm (forward) and sa_favor_dden -> sa_favor_scan_b_cum x2 ->
 * sa_favor_scan_a_state -> sa_favor_features_project_bwd x2 (backward): no [B*N*G, LDF] tensor (dd, phi, d phi) is ever written.
 * q / k / v / dq / dk / dv: fp32 rows of `stride` floats whose first G*64 columns are the global heads; ps = projection matrix [m][64] with the
 * data normaliser folded in; tiles = 5 * 16 KiB from sa_favor_fused_proj_tiles(ps); offq / offk [B*N*G] floats, amq [B*N*G] int32 and gmax_ws (8 bytes)
 * from sa_favor_fused_prepass; state buffers of sa_favor_fused_state_bytes bytes; dden_ws B*N*G floats, tsum_ws B*G*ceil(N/64) floats.
 * No fp32 atomics: results are run-to-run deterministic.  m <= 272, head width 64.
 * Round 6: sa_favor_fused_proj_tiles also records (in the last 16 bytes of `tiles`, never an operand row) whether every lo half of the split matrix is zero, i.e.
 * whether ps is bf16-representable -- what the Python layer hands over in throughput mode (a bf16 copy of the folded fp32 matrix, like every dense weight of
 * that mode).  The kernels then skip the lo * hi products and the lo half of every projection-slab transfer: exact zeros, bit-identical results. */
int64_t sa_favor_fused_state_bytes(int B, int N, int G, int m);
int sa_favor_fused_proj_tiles(const float *ps, int m, void *tiles, void *stream);
int sa_favor_fused_prepass(const float *q, const float *k, int stride, int G, const void *tiles, float *offq, int32_t *amq, float *offk, void *gmax_ws,
                           int64_t rows, int m, int dh, void *stream);
/* Local-window heads co-launched with the FAVOR+ heads (last argument of sa_favor_fused_fwd / _bwd; NULL = none): the arguments of sa_local_attn_fwd
 * (q, k rotated; o, lse, o_lp) or sa_local_attn_bwd (out, dout, lse_in, dq, dk, dv, Dbuf, dv_lp) for the same B, N.  Their blocks are appended to the FAVOR+
 * launches they do not depend on ([scan A | local forward]; [scan B dq | reversed states | local dq], [scan B dk | scan A dv | local dk dv]) and fill those
 * launches' tails; results are identical to separate calls.  SA_EUNSUPPORTED (nothing launched) with SA_DBG_LOCAL_ATTN_EXACT or, backward, without state_fwd. */
typedef struct sa_local_attn_args {
    const float *q, *k, *v;
    int32_t q_stride, q_off, k_stride, k_off, v_stride, v_off, o_stride, o_off;
    float *o, *lse;            /* forward */
    void *o_lp;
    const float *out, *dout, *lse_in;   /* backward */
    float *dq, *dk, *dv, *Dbuf;
    void *dv_lp;
    int32_t L, W;
} sa_local_attn_args;
int sa_favor_fused_fwd(const float *q, const float *k, const float *v, int stride, const void *tiles, const float *ps, const float *offq, const float *offk,
                       const void *gmax_ws, float *attn, int attn_stride, float *inv_out, float den_eps, int B, int N, int G, int m, float *state,
                       void *attn_lp, const sa_local_attn_args *la, void *stream);
int sa_favor_fused_bwd(const float *q, const float *k, const float *v, int stride, const void *tiles, const float *ps, const float *offq, const int32_t *amq,
                       const float *offk, const void *gmax_ws, const float *dattn, const float *attn, int attn_stride, const float *inv, float *dq, float *dk,
                       float *dv, int B, int N, int G, int m, const float *state_fwd, float *state_ws, float *dden_ws, float *tsum_ws, void *dq_lp,
                       void *dk_lp, void *dv_lp, const sa_local_attn_args *la, void *stream);

/* ---- deterministic mode (the reference's --deterministic flag: torch.backends.cudnn.deterministic, src/utils/general.py:336-338) -------------------
 * Fixed-order forms of the reductions the throughput path accumulates with fp32 atomics (csrc/deterministic.hip); the host calls them INSTEAD of the
 * fused / atomic forms when the flag is set, and SA_DBG_DETERMINISTIC makes sa_bn_forward / sa_bn_backward keep per-block partial sums in a larger
 * sums_ws ((1 + 64) * 2 * C floats).  Results are bit-identical from run to run.
 *   sa_colsum_det   : db[c] += sum_m g[m][c] (bias gradients); ws >= sa_colsum_det_workspace_bytes(C)
 *   sa_vq_stats_det : counts / dw / sqerr of sa_vq_assign recomputed from its idx output (baseline.py:66-69, :82); err_ws = K floats
 *   sa_embed_scatter_det : sa_embed_scatter over a table of nrows rows */
int64_t sa_colsum_det_workspace_bytes(int C);
int sa_colsum_det(const void *g, int dtype, int64_t M, int C, int cstride, float *db, void *ws, int64_t ws_bytes, void *stream);
int sa_vq_stats_det(const float *rows, const float *codebook, const int64_t *idx, int64_t M, int K, int D, float *counts, float *dw, float *sqerr,
                    float *err_ws, void *stream);
int sa_embed_scatter_det(const float *dy, float *dtable, const int64_t *idx, int per_position, int dim, int N, int64_t R, int nrows, void *stream);
/* Performer sums in a fixed order: out[0] (+)= sum x (one block); out[0] (+)= a . b (b fp32 / bf16; ws = 1 024 floats); sa_cross_entropy with per-row losses
 * (row_loss[R], summed by sa_sum_det); the summand matrix of the LayerNorm weight gradient (column sums through sa_colsum_det). */
int sa_sum_det(const float *x, int64_t n, float *out, int accumulate, void *stream);
/* sa_mse with the squared errors summed in a fixed order (2 048 contiguous ranges, then one block): loss_sum += sum; ws: 2 048 floats.  The logged
 * loss and the validation MSE (the key metric that selects checkpoint_key_metric=*.pt) are then bit-reproducible too. */
int sa_mse_det(const float *a, const float *b, int64_t n, float *loss_sum, float *grad, float gscale, float *ws, void *stream);
int sa_dot_det(const float *a, const void *b, int b_dtype, int64_t n, float *out, int accumulate, float *ws, void *stream);
int sa_cross_entropy_rows(const float *logits, const int64_t *target, int64_t R, int V, float *row_loss, void *dlogits, int d_dtype, float gscale, void *stream);
int sa_layernorm_dwprod(const float *dy, const float *x, const float *stats, float *prod, int64_t R, int C, void *stream);

/* ---- collectives of the data-parallel path for a host without torch.distributed (csrc/comm.hip): RCCL behind the C ABI ----------------------------------
 * One communicator per process (one process per GPU).  Replaces DistributedDataParallel's gradient all-reduce (reference run_vqvae.py:71-77,
 * run_transformer.py:95-103; here per flat 32 MiB bucket, or reduce-scatter + all-gather) and dist.all_reduce(encodings_sum) / dist.all_reduce(dw) of the EMA
 * quantizer (src/networks/vqvae/baseline.py:70-72; here ONE call on the packed [K + K D] statistics buffer that sa_vq_assign fills and sa_vq_ema_update reads).
 * RCCL is resolved with dlopen at the first call (librccl.so.1, or SA_RCCL_LIB): the library has no link-time dependency on it, and SA_EUNSUPPORTED says it was
 * not found.  Every call enqueues on the caller's stream and returns.  The Python host of this repository keeps using torch.distributed (backend nccl = RCCL);
 * these entry points are for a non-torch host.  id: 128 bytes from sa_comm_unique_id on rank 0, handed to the other ranks by the host
 * (file, socket, environment).  dtype: SA_F32 / SA_BF16 / SA_F16. */
typedef struct sa_comm sa_comm;
int sa_comm_unique_id(void *id_out_128_bytes);
int sa_comm_init(sa_comm **comm, const void *id_128_bytes, int rank, int world);
int sa_comm_destroy(sa_comm *comm);
int sa_comm_rank(const sa_comm *comm);
int sa_comm_world(const sa_comm *comm);
int sa_comm_all_reduce_sum(sa_comm *comm, void *buf, int64_t n, int dtype, void *stream);                                       /* in place */
int sa_comm_reduce_scatter_sum(sa_comm *comm, const void *send, void *recv, int64_t n_per_rank, int dtype, void *stream);      /* send: world * n_per_rank */
int sa_comm_all_gather(sa_comm *comm, const void *send, void *recv, int64_t n_per_rank, int dtype, void *stream);              /* recv: world * n_per_rank */
const char *sa_comm_last_error(void);   /* text of the calling thread's last failed sa_comm_* call */

#ifdef __cplusplus
}
#endif
#endif
