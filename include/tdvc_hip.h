/*
 * tdvc_hip.h — C-ABI of libtdvc_hip.so: MI355X (gfx950) kernels for TDVC's per-P-frame
 * encode / reconstruct hot path (`main/model/pnet.py::VideoCompressor.forward`).
 *
 * Boundary rules
 *   - plain pointers, sizes and a `void* stream` (a hipStream_t; NULL = default stream);
 *     no torch types.  All device pointers must be 16-byte aligned.
 *   - every entry point ENQUEUES on `stream` and returns immediately (no host sync, no
 *     allocation, graph-capture safe) — the contract of the reference's only FFI crossing,
 *     `_ext.dcn_v2_forward`, which enqueues on the current CUDA stream
 *     (main/utils/dcnv2/src/cuda/dcn_v2_cuda.cu:78).
 *   - return value: 0 on success, a negative TDVC_E* code on invalid arguments, or the
 *     positive hipError_t of a failed launch.  (The reference only printf's launch failures,
 *     src/cuda/dcn_v2_im2col_cuda.cu:346-350; here they are returned.)
 *
 * Feature maps ("fmap") are channel-innermost: element (n, y, x, c) lives at
 *     base + n*sn + (y*W + x)*sp + c            (strides in ELEMENTS)
 * so a tensor may be a channel slice of a wider buffer (concatenation without copies).
 * fp16 fmaps need C, sp and the channel offset to be multiples of 8.
 *
 * Which reference function each entry replaces is stated per declaration (file:line under
 * /root/reference).
 */
#ifndef TDVC_HIP_H
#define TDVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDVC_OK 0
#define TDVC_EINVAL (-1)   /* bad shape / alignment / unsupported configuration */
#define TDVC_ENOSUP (-2)

/* ---------------------------------------------------------------- descriptors */
typedef struct {
  void* p;      /* device pointer to element (0,0,0,0) of the view */
  int32_t N, H, W, C;
  int64_t sn;   /* batch stride (elements) */
  int32_t sp;   /* pixel stride (elements) */
  int32_t dtype; /* TDVC_F16 or TDVC_F32 */
} tdvc_fmap;

enum { TDVC_F16 = 0, TDVC_F32 = 1 };
enum { TDVC_ACT_NONE = 0, TDVC_ACT_RELU = 1, TDVC_ACT_LRELU = 2, TDVC_ACT_CLAMP01 = 3, TDVC_ACT_SIGMOID = 4 };
enum { TDVC_GDN_NONE = 0, TDVC_GDN_FWD = 1, TDVC_GDN_INV = 2 };
enum { TDVC_OUT_NHWC = 0,       /* y is an fmap (fp16 or fp32) */
       TDVC_OUT_SHUFFLE2 = 1,   /* PixelShuffle(2): out channel o -> (c=o/4, i=(o/2)&1, j=o&1), y has 2H x 2W */
       TDVC_OUT_NCHW_F32 = 2 }; /* planar fp32 [N][C][H][W] (module boundary) */

#define TDVC_MAX_TAPS 49

typedef struct {
  tdvc_fmap x;            /* input, fp16 */
  tdvc_fmap y;            /* output view (post-shuffle geometry for SHUFFLE2) */
  const void* w;          /* weights packed by tdvc_pack_conv_weights (fp16, MFMA fragment order) */
  const float* bias;      /* [cout_pad] fp32, may be NULL */
  int32_t cout;           /* real output channels */
  int32_t ntaps;          /* taps actually evaluated (masked taps are dropped at pack time) */
  int8_t tap_dy[TDVC_MAX_TAPS], tap_dx[TDVC_MAX_TAPS]; /* tap offsets in [0,KH) x [0,KW) */
  int32_t kh, kw, stride, pad;
  int32_t ck;             /* channel chunk staged in LDS per pass: 8,16,32 or 64 (from tdvc_conv_plan) */
  int32_t square_input;   /* 1: stage x^2 (GDN norm pool) */
  int32_t gdn;            /* TDVC_GDN_*: v = aux * rsqrt(v) (FWD) or aux * sqrt(v) (INV) */
  tdvc_fmap aux;          /* GDN multiplicand (same geometry as y pre-shuffle), fp16 */
  int32_t act; float slope;
  int32_t round_before_act; /* 1: round v to fp16 before the activation (DCN quirk, dcn_v2_amp.py:67-68) */
  tdvc_fmap res;          /* residual added AFTER the activation, at output coordinates; p==NULL: none */
  tdvc_fmap res2;         /* optional second residual (fp16), same rules */
  int32_t out_mode;
  int32_t s2d;            /* 1: `x` is read through a 2x2 space-to-depth view: the virtual input is
                             (H/2, W/2, 4C) with channel q*C + c = pixel (2Y + q/2, 2X + q%2), channel c.
                             A 3x3 stride-2 pad-1 conv becomes a 2x2 stride-1 conv (kh=kw=2, pad=1 on the
                             top/left only) over that view; weights are packed for the virtual conv
                             (tdvc_amd/ops.py::pack_conv(s2d=True)).  Needs H, W even and C % 32 == 0. */
  int32_t bcast_T;        /* ABI 3.  > 0: the conv result b (1x1, stride 1, cout == y.C == 64, no activation / residual / GDN) is not
                             stored but broadcast-added over the bcast_T channel slices of width y.C that start at y:
                             y[:, t] = lrelu(y[:, t] + b), in place (y's pixel stride covers the slices) -- Bottleneck3D's temporal
                             conv + `out + temporal` + LeakyReLU (pnet.py:304-314) as ONE pass over the 4-frame buffer.  The conv
                             input may be channels of the same buffer (a 1x1 conv reads only the pixels it then overwrites). */
  float bcast_slope;
  float* chan_sum;        /* ABI 4.  not NULL: next to storing y the launch writes partial per-channel sums of the STORED fp16 values,
                             chan_sum[n][row][cout] for row < tdvc_conv_chan_sum_rows(d) (fp32; every row is written) -- the
                             SELayer's global average pool (inflate.py:204) without a second pass over y.  Only the convs for which
                             tdvc_conv_chan_sum_rows() returns > 0 accept it. */
} tdvc_conv_desc;

/* ---------------------------------------------------------------- library */
/* Library/ABI version; bumps when a struct above changes. */
int tdvc_abi_version(void);
/* Human-readable description of the last error on this thread (never NULL). */
const char* tdvc_last_error(void);
/* Creates the current device's scratch pages (a zero page the halo DMA reads, a dump page masked stores go to).  The
 * launchers create them on first use, synchronously: call this once per device before capturing launches into a graph.
 * Thread-safe; one allocation per device for the life of the process. */
int tdvc_prepare_device(void);
/* Name of the kernel the last tdvc_conv2d call on this thread dispatched to ("conv_mfma_v7", "conv_mfma<4,2,1>",
 * ...; "" before the first call).  Diagnostic for profiles and benchmarks; never NULL. */
const char* tdvc_last_conv_kernel(void);

/* ---------------------------------------------------------------- conv transforms
 * Replaces every torch.nn.Conv2d / Conv3d(1,3,3) / Conv3d(3,1,1) / compressai
 * conv3x3 / subpel_conv3x3 / MaskedConv2d / GDN 1x1 call on the path:
 * main/model/pnet.py:93-96,132-166,180-184,213-262,278-292,309-317,327-332;
 * main/model/flownet.py:187-227; main/model/encoder_v3.py:17-40,46-69;
 * main/utils/utils.py:52-56; compressai layers (not in tree).                       */

/* Choose the LDS channel chunk for a geometry. Returns ck (8/16/32/64) or <0. */
int tdvc_conv_plan(int cin, int kh, int kw, int stride);
/* Size in bytes of the packed weight blob for (cout, cin, ntaps, ck). */
int64_t tdvc_conv_packed_bytes(int cout, int cin, int ntaps, int ck);
/* Host-side packing: w_oihw fp32 [cout][cin_real][kh][kw] -> fragment-ordered fp16 (host
 * memory, caller uploads).  tap list selects/permutes kernel positions; channels
 * cin_real..cin-1 are zero.  dst is uint16 (IEEE half bits). */
int tdvc_pack_conv_weights(const float* w_oihw, int cout, int cin_real, int cin, int kh, int kw,
                           int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, int ck, uint16_t* dst);
int tdvc_conv2d(const tdvc_conv_desc* d, void* stream);
/* Rows per image of tdvc_conv_desc::chan_sum this conv would write, 0 when its kernel has no fused channel sum (the caller then
 * runs tdvc_channel_sum over y), < 0 on a malformed descriptor.  `chan_sum` itself is ignored here. */
int tdvc_conv_chan_sum_rows(const tdvc_conv_desc* d);
/* The fp32 islands: main/model/pnet.py:33-49,57-73 run both coders with autocast OFF (and enabled_amp=False runs the
 * whole model fp32).  Same descriptor with an fp32 input fmap (C %% 8 == 0) and `w` = the fp32 packing below; fp32
 * accumulation on v_mfma_f32_32x32x2_f32; aux / residual / output fmaps fp32 or fp16.  tdvc_conv2d forwards here
 * when d->x.dtype == TDVC_F32, so fixed descriptor chains (tdvc_ar_decode_serial) run in either precision. */
int tdvc_conv2d_f32(const tdvc_conv_desc* d, void* stream);

/* ---------------------------------------------------------------- fused conv pair
 * y = act2(conv2(act1(conv1(x)))) [+ x] [+ res2] for two 3x3 / stride 1 / pad 1 / 64 -> 64 convs in one launch, the
 * intermediate map staying in LDS: `Res_Block.forward` (main/utils/utils.py:52-56: conv, ReLU, conv, + identity) and the
 * LeakyReLU conv pairs of main/model/pnet.py:132-166.  Results equal two tdvc_conv2d launches up to the fp32 summation
 * order (the intermediate is rounded to fp16 exactly as the stored map would be).  Inference path: nothing is kept for a
 * backward pass. */
typedef struct {
  tdvc_fmap x, y;         /* fp16 maps of 64 channels, same geometry; views (sp > 64) allowed; not in place */
  const void* w;          /* tdvc_pack_conv_pair_weights, on the device */
  const float* bias;      /* [2][64] on the device: conv1 | conv2 */
  int32_t act1; float slope1;   /* TDVC_ACT_NONE / RELU / LRELU after conv1 */
  int32_t act2; float slope2;   /* ... after conv2, before the additions */
  int32_t add_input;      /* 1: y += x */
  tdvc_fmap res2;         /* optional further residual (fp16, output geometry); p == NULL: none */
} tdvc_conv_pair_desc;
int64_t tdvc_conv_pair_packed_bytes(void);
/* host-side packing of the two fp32 OIHW [64][64][3][3] weights into per-wave v_mfma_f32_16x16x32_f16 A fragments
 * (layout in tdvc_amd/csrc/conv_pair.hip); dst: tdvc_conv_pair_packed_bytes() bytes of host memory, caller uploads */
int tdvc_pack_conv_pair_weights(const float* w1_oihw, const float* w2_oihw, uint16_t* dst);
/* 1 when tdvc_conv_pair takes this descriptor (geometry / size limits, LeakyReLU slopes in [0, 1], x and y not overlapping
 * except as disjoint channel windows of one buffer), 0: run the two convs through tdvc_conv2d.  y.p == x.p asks about the
 * geometry only (output not allocated yet); tdvc_conv_pair itself refuses that descriptor. */
int tdvc_conv_pair_supported(const tdvc_conv_pair_desc* d);
int tdvc_conv_pair(const tdvc_conv_pair_desc* d, void* stream);

/* ---------------------------------------------------------------- deformable conv (motion compensation)
 * Fused modulated deformable 3x3 conv, fp16 NHWC, no column buffer:
 * replaces DCN.forward's `_DCNv2.apply` (main/utils/dcnv2/dcn_v2_amp.py:219-234) =
 * dcn_v2_cuda_forward (src/cuda/dcn_v2_cuda.cu:20-95) + modulated_deformable_im2col
 * (src/cuda/dcn_v2_im2col_cuda.cu:125-195).
 * om: fp16 fmap with 27*G channels = raw conv_offset_mask output [o1 | o2 | mask_logits]
 * (sigmoid applied here, dcn_v2_amp.py:220-223). Cin = Cout = 8*G, 8 channels per group. */
typedef struct {
  tdvc_fmap x, om, y;
  const void* w;        /* packed with tdvc_pack_conv_weights(ck = 64, 9 taps row-major) */
  const float* bias;
  int32_t groups;
  int32_t act; float slope; int32_t round_before_act;
  void* x_planar;       /* optional scratch of N*H*W*8*groups fp16 values (ABI v2): when non-NULL, x is first re-laid
                           group-planar ([n][g][H][W][8]) and the bilinear gathers read that copy (neighbouring pixels of a
                           group share 128-byte lines); NULL: gather from the NHWC map */
} tdvc_dcn_desc;
int tdvc_dcn_fused(const tdvc_dcn_desc* d, void* stream);

/* fp32 NCHW operator with the exact signature semantics of `_ext.dcn_v2_forward`
 * (src/vision.cpp:3-8, src/dcn_v2.h:9-45): contiguous fp32 tensors, output [B][Cout][Ho][Wo]. */
int tdvc_dcn_v2_forward_f32(const float* input, const float* weight, const float* bias,
                            const float* offset, const float* mask, float* output,
                            int B, int C, int H, int W, int Cout,
                            int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                            int deformable_group, void* stream);
/* `_ext.dcn_v2_backward` (src/dcn_v2.h:48-92, src/cuda/dcn_v2_cuda.cu:97-216). All grads
 * are OVERWRITTEN (zero-initialised inside). `columns` is caller-provided scratch of
 * C*kh*kw*Ho*Wo floats (per-sample column buffer, reused across the batch). */
int tdvc_dcn_v2_backward_f32(const float* input, const float* weight, const float* bias,
                             const float* offset, const float* mask, const float* grad_output,
                             float* grad_input, float* grad_offset, float* grad_mask,
                             float* grad_weight, float* grad_bias, float* columns,
                             int B, int C, int H, int W, int Cout,
                             int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                             int deformable_group, void* stream);

/* ---------------------------------------------------------------- layout / elementwise */
/* NCHW fp32 [N][C][H][W] -> fmap (fp16 or fp32), channels >= C zero-filled up to y.C. */
int tdvc_nchw_to_fmap(const float* src, int C, const tdvc_fmap* y, void* stream);
/* fmap (fp16/fp32) -> NCHW fp32, first C channels. */
int tdvc_fmap_to_nchw(const tdvc_fmap* x, int C, float* dst, void* stream);
/* y = act(a * gate[n][c] (optional) ) + r (optional) ; also y2 (optional, other dtype) gets the same value.
 * gate NULL => no scaling.  Used for SE scaling (main/model/inflate.py:204-208) and plain add/sub. */
int tdvc_scale_act_res(const tdvc_fmap* a, const float* gate, int act, float slope,
                       const tdvc_fmap* r, float r_sign, const tdvc_fmap* y, const tdvc_fmap* y2, void* stream);
/* offset[c] += flow[c & 1] (main/model/pnet.py:163); off fp16 in place, flow fp32 2-channel fmap. */
int tdvc_add_flow(const tdvc_fmap* off, const tdvc_fmap* flow, void* stream);
/* x[:, t] = lrelu(x[:, t] + b) for the T channel-slices of width b.C (pnet.py:313-314). */
int tdvc_bcast_add_act(const tdvc_fmap* x, const tdvc_fmap* b, int T, float slope, void* stream);

/* ---------------------------------------------------------------- SE attention
 * main/model/inflate.py:159-208.  Deterministic two-stage mean: partial[N][nblocks][C] fp32
 * scratch, then the gate kernel sums the partials in fixed order, applies the two 1x1 convs
 * (ReLU, Sigmoid) and writes gate[N][C].  w1: [Cmid][C], w2: [C][Cmid]. */
int tdvc_channel_sum(const tdvc_fmap* x, float* partial, int nblocks, void* stream);
int tdvc_se_gate(const float* partial, int nblocks, float inv_count, int N, int C, int Cmid,
                 const float* w1, const float* b1, const float* w2, const float* b2,
                 float* gate, void* stream);

/* ---------------------------------------------------------------- resampling
 * bilinear x2 upsample, align_corners=False (nn.Upsample, main/model/pnet.py:117). fp16 fmaps. */
int tdvc_upsample2x(const tdvc_fmap* x, const tdvc_fmap* y, void* stream);
/* 2x2 average pooling, fp32 fmap -> fp32 fmap (F.avg_pool2d, main/model/flownet.py:103-114). */
int tdvc_avgpool2(const tdvc_fmap* x, const tdvc_fmap* y, void* stream);
/* One SPyNet level's input assembly — the "bilinear warp" of the path
 * (main/model/flownet.py:119-138 + flow_warp :8-48):
 *   flow_up = 2 * bilinear_x2(flow_lo, align_corners=True)   (zeros if flow_lo == NULL)
 *   warped  = grid_sample(supp, grid + flow_up, bilinear, border, align_corners=True)
 *   cat8    = [ref(3) | warped(3) | flow_up(2)] as fp16; flow_up is also kept in fp32.
 * ref/supp: fp32 fmaps with C>=3; flow_lo/flow_up: fp32 2-channel fmaps; cat8: fp16 C=8. */
int tdvc_spynet_level_input(const tdvc_fmap* ref, const tdvc_fmap* supp, const tdvc_fmap* flow_lo,
                            const tdvc_fmap* flow_up, const tdvc_fmap* cat8, void* stream);
/* generic bilinear resize, align_corners=False, fp32 fmaps, optional per-channel scale of the
 * output (flow rescale, flownet.py:153-173). */
int tdvc_resize_bilinear(const tdvc_fmap* x, const tdvc_fmap* y, const float* chscale, void* stream);

/* ---------------------------------------------------------------- in-loop filter matching
 * FeatureFix.forward, main/model/pnet.py:219-255. */
/* scale x scale average pooling (floor), fp16 fmap -> fp32 [N][hp][wp][C] (nn.AvgPool2d, :224-225).
 * Two order-fixed stages through `work` (>= tdvc_avgpool_k_work_floats(..) floats of device memory). */
int64_t tdvc_avgpool_k_work_floats(int N, int hp, int wp, int C, int scale);
int tdvc_avgpool_k(const tdvc_fmap* x, int scale, float* pooled, int hp, int wp, float* work, int64_t work_floats, void* stream);
/* 3x3 / stride-3 / pad-3 patches of both pooled maps, L2-normalise, cosine similarity, first
 * -index argmax over reference patches -> idx[N][L] int32, L = ((hp+3)/3+1)*((wp+3)/3+1)
 * (F.unfold + normalize + bmm + max, :230-236). */
int tdvc_patch_match(const float* pin, const float* pref, int N, int hp, int wp, int C,
                     int32_t* idx, void* stream);
/* gather full-resolution reference blocks by idx (unfold/gather/fold with kernel 3*scale,
 * :247-254), per-pixel cosine similarity with fin over channels (:255), write
 * cat = [fin*cor | out*cor] (fp16, 128 channels) (:257). */
int tdvc_match_gather(const tdvc_fmap* fin, const tdvc_fmap* fref, const int32_t* idx, int scale,
                      int hp, int wp, const tdvc_fmap* cat, void* stream);

/* ---------------------------------------------------------------- entropy model (rate terms)
 * compressai EntropyBottleneck.forward, called from main/model/pnet.py:34,58:
 *   z_hat = round(z - median) + median (eval)  |  z + noise (train, noise != NULL)
 *   lik   = |sigmoid(s*upper) - sigmoid(s*lower)|, logits chain over filters (3,3,3,3), >= 1e-9
 * z: fp32 fmap; params: [C][59] floats = softplus(matrix0..4) (33), bias0..4 (13),
 * tanh(factor0..3) (12), median (1); z_hat: fp16 or fp32 fmap.
 * *bits_out = sum(-log2 lik) as a double, by a deterministic two-stage reduction through
 * partial[partial_cap] (needs >= ceil(numel/256) floats). */
int tdvc_eb_forward(const tdvc_fmap* z, const float* params, const tdvc_fmap* noise,
                    const tdvc_fmap* z_hat, double* bits_out, float* partial, int partial_cap, void* stream);
/* compressai GaussianConditional likelihood:
 *   v   = |round(y - mean)| (eval) | |y + noise - mean| (train)
 *   lik = Phi((0.5-v)/s) - Phi((-0.5-v)/s), s = max(scale, 0.11), lik >= 1e-9
 * gp: fp32 fmap with 2C channels [scales | means]. */
int tdvc_gc_forward(const tdvc_fmap* y, const tdvc_fmap* gp, const tdvc_fmap* noise,
                    double* bits_out, float* partial, int partial_cap, void* stream);
/* y_hat = round(y) (half-to-even, torch.round) or y + noise. */
int tdvc_quantize(const tdvc_fmap* y, const tdvc_fmap* noise, const tdvc_fmap* y_hat, void* stream);

/* ---------------------------------------------------------------- range coder (host side)
 * compressai `ans` extension (rans64, 16-bit precision, 4-bit bypass) used by
 * Cheng2020Anchor.compress (main/model/pnet.py:46-49,70-73).  Pure host code, like the
 * reference's.  Returns number of bytes written to `out` (<= cap) or <0. */
int64_t tdvc_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n,
                         const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                         const int32_t* offsets, uint8_t* out, int64_t cap);
int tdvc_rans_decode(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                     const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                     const int32_t* offsets, int32_t* symbols_out);

/* Stateful decoder (compressai RansDecoder.set_stream / decode_stream): create copies the stream;
 * decode continues where the previous call stopped. */
void* tdvc_rans_decoder_create(const uint8_t* data, int64_t nbytes);
int tdvc_rans_decoder_decode(void* handle, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                             int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                             int32_t* symbols_out);
void tdvc_rans_decoder_destroy(void* handle);

/* compressai `_CXX.pmf_to_quantized_cdf` (used by EntropyModel._pmf_to_cdf inside update(),
 * main/model/pnet.py:47,71): pmf[n] -> quantised cdf[n+1] at `precision` bits, every bin >= 1. Host code. */
int tdvc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf_out);

/* ---------------------------------------------------------------- autoregressive context model (compress)
 * `_compress_ar` / `_decompress_ar` of compressai's JointAutoregressiveHierarchicalPriors, called from
 * main/model/pnet.py:48,72.  Position (h, w) depends on rows h-2..h within +-2 columns, so positions
 * with equal w + 3h are independent: the encoder walks W + 3(H-1) anti-diagonal steps, each step a
 * batch of <= H positions whose 12-tap neighbourhoods are gathered into a dense matrix
 * (tdvc_ar_gather), pushed through the context conv and entropy_parameters as 1x1 MFMA convs, then
 * quantised and written back (tdvc_ar_quantize).
 * pos: int32 [npos][2] = (h, w) on the device.  y_hat: fp16 fmap (H, W, M), zero-initialised.
 * params: fp16 fmap (H, W, 2M) (h_s output).  x1: fp16 fmap (1, 1, npos, 12*M) neighbourhoods in tap-major
 * order; pc: fp16 fmap (1, 1, npos, 4M): channels [0, 2M) receive params of the positions. */
int tdvc_ar_gather(const tdvc_fmap* y_hat, const tdvc_fmap* params, const int32_t* pos, int npos,
                   const tdvc_fmap* x1, const tdvc_fmap* pc, void* stream);
/* gp: fp32 fmap (1, 1, npos, 2M) [scales | means].  Encoder (symbols_in == NULL): q = round(y - mean);
 * decoder: q = symbols_in[h][w][c].  Writes y_hat[h][w][c] = q + mean, symbols[h][w][c] = q (int32,
 * [H][W][M]) and indexes[h][w][c] = #{table entries < max(scale, 0.11)} clipped to ntable-1
 * (GaussianConditional.build_indexes).  scale_table: ntable floats. */
int tdvc_ar_quantize(const tdvc_fmap* y, const tdvc_fmap* gp, const int32_t* pos, int npos,
                     const float* scale_table, int ntable, const int32_t* symbols_in,
                     const tdvc_fmap* y_hat, int32_t* symbols, int32_t* indexes, void* stream);
/* indexes only (decoder: needed before the symbols can be read from the stream). */
int tdvc_ar_indexes(const tdvc_fmap* gp, const int32_t* pos, int npos, const float* scale_table, int ntable,
                    int M, int W, int32_t* indexes, void* stream);
/* The decoder's whole context loop for ONE image, natively: positions pos_table[0 .. npos_total) in raster (stream)
 * order, per position tdvc_ar_gather -> convs[0 .. nconvs) (the caller's descriptors: context conv over x1 into pc,
 * entropy_parameters into gp; fixed buffers) -> tdvc_ar_indexes -> host range decoder on `data` (tables as in
 * tdvc_rans_decode, HOST memory) -> tdvc_ar_quantize.  idx_dev / sym_dev: device int32 [H][W][M]; y_hat is filled.
 * Synchronises the stream once per position (the stream order of compressai's bitstream leaves no other choice). */
int tdvc_ar_decode_serial(const uint8_t* data, int64_t nbytes, const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                          const int32_t* offsets, const tdvc_fmap* y_hat, const tdvc_fmap* params, const tdvc_fmap* x1,
                          const tdvc_fmap* pc, const tdvc_conv_desc* convs, int nconvs, const tdvc_fmap* gp,
                          const int32_t* pos_table, int npos_total, int M, int W, const float* scale_table, int ntable,
                          int32_t* idx_dev, int32_t* sym_dev, void* stream);
/* The context loop of ONE image over anti-diagonals (wavefront order), natively, in either direction.  pos_dev: device
 * int32 [H*W][2] in wavefront order, step s owning step_sizes[s] (HOST array) consecutive entries; x1 / pc / gp and the
 * conv descriptors' x / y maps are (1, 1, >= max step, C) staging buffers whose width is set to the step's size per step.
 * Encoder (y != NULL, data == NULL; replaces `_compress_ar`'s position loop, main/model/pnet.py:48,72): gather -> convs ->
 * tdvc_ar_quantize per step; sym_dev / idx_dev are raster [H][W][M]; nothing synchronises.
 * Decoder (data != NULL, y == NULL) of a stream whose symbols were emitted in wavefront order (an extension: the
 * reference's streams are raster-ordered, tdvc_ar_decode_serial reads those): gather -> convs -> indexes -> host range
 * decoder -> quantise, one stream synchronisation per step; sym_dev / idx_dev are [H*W][M] in wavefront order. */
int tdvc_ar_wavefront(const uint8_t* data, int64_t nbytes, const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                      const int32_t* offsets, const tdvc_fmap* y, const tdvc_fmap* y_hat, const tdvc_fmap* params,
                      const tdvc_fmap* x1, const tdvc_fmap* pc, const tdvc_conv_desc* convs, int nconvs, const tdvc_fmap* gp,
                      const int32_t* pos_dev, const int32_t* step_sizes, int nsteps, int M, int W,
                      const float* scale_table, int ntable, int32_t* idx_dev, int32_t* sym_dev, void* stream);
/* q[n][h][w][c] = round(z - median[c]) as int32 in the fmap's own order (factorised-prior symbols). */
int tdvc_round_symbols(const tdvc_fmap* z, const float* median, int32_t* out, void* stream);

/* ---------------------------------------------------------------- conv backward (training path)
 * torch.autograd's conv backward on the reference path (tools/train.py:142-159 drives loss.backward()):
 * dX runs on tdvc_conv2d with weights re-packed by tdvc_pack_conv_weights_indexed (tdvc_amd/convpack.py builds
 * the index tables); dW is tdvc_conv_wgrad; db is tdvc_channel_sum. */
/* Device-side packing: element (row r, channel c, tap t) of the packed conv = w[row_off[r] + chan_off[c] + tap_off[t]],
 * zero when row_off[r] < 0, chan_off[c] < 0 or tap_mask[t] & (row_mask[r] | chan_mask[c]).  All pointers are device
 * memory; `dst` has tdvc_conv_packed_bytes(cout, cin, ntaps, ck) bytes. */
int tdvc_pack_conv_weights_indexed(const float* w, const int32_t* row_off, const int32_t* chan_off, const int32_t* tap_off,
                                   const uint8_t* row_mask, const uint8_t* chan_mask, const uint8_t* tap_mask,
                                   int cout, int cin, int ntaps, int ck, void* dst, void* stream);
/* fp32 twin of the packing (same item order, 8 floats per item: 2 x tdvc_conv_packed_bytes bytes) for tdvc_conv2d_f32. */
int tdvc_pack_conv_weights_indexed_f32(const float* w, const int32_t* row_off, const int32_t* chan_off, const int32_t* tap_off,
                                       const uint8_t* row_mask, const uint8_t* chan_mask, const uint8_t* tap_mask,
                                       int cout, int cin, int ntaps, int ck, float* dst, void* stream);
/* The same for many layers in one launch (after an optimizer step every packed layer follows its parameters).  `jobs`
 * and `block_start` (njobs + 1 prefix sums of tdvc_pack_job_blocks) are DEVICE arrays; a job with bias_src also
 * gathers its bias: bias_dst[i] = bias_src[bias_perm ? bias_perm[i] : i], i < cout. */
typedef struct tdvc_pack_job {
  const float* w;
  const int32_t *row_off, *chan_off, *tap_off;
  const uint8_t *row_mask, *chan_mask, *tap_mask;
  void* dst;
  const float* bias_src;
  float* bias_dst;
  const int32_t* bias_perm;
  int32_t cout, cin, ntaps, ck;
} tdvc_pack_job;
int64_t tdvc_pack_job_blocks(int cout, int cin, int ntaps, int ck);
int tdvc_pack_conv_weights_batch(const tdvc_pack_job* jobs, const int32_t* block_start, int njobs, int total_blocks, void* stream);
/* dW[row_off[co] + chan_off[ci] + tap_off[t]] += scale * sum_{n,oy,ox} g[n,oy,ox,co] * x[n, oy*stride + dy_t - pad, ox*stride + dx_t - pad, ci]
 * for the forward conv y = conv(x, W); g = dL/dy (fp16 fmap, >= cout channels; for a sub-pixel conv the un-shuffled
 * gradient in packed-row order), dW the fp32 parameter gradient.  row_off[cout] / chan_off[x.C] / tap_off[ntaps] are the
 * layer's forward packing tables (device int32; negative row / channel entries are skipped); square_x contracts with
 * x^2 instead of x (GDN norm pool).  Two order-fixed stages
 * through `work` (>= tdvc_conv_wgrad_work_floats(cout, x.C, ntaps, N, Ho, Wo) floats of device memory). */
int64_t tdvc_conv_wgrad_work_floats(int cout, int cin, int ntaps, int N, int Ho, int Wo);
int tdvc_conv_wgrad(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                    int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                    const int32_t* tap_off, int square_x, float scale, float* dw, float* work, int64_t work_floats, void* stream);
/* The same, and the bias gradient from the dY tiles the kernel already holds (saves a pass over dY):
 * db[bias_index ? bias_index[co] : co] += scale * sum_{n,oy,ox} g[n,oy,ox,co], co < cout; db == NULL skips it. */
int tdvc_conv_wgrad_bias(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                         int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                         const int32_t* tap_off, int square_x, float scale, float* dw, const int32_t* bias_index, float* db,
                         float* work, int64_t work_floats, void* stream);

/* The two stages apart (training: ~220 layers per step, each reduce a 15 us launch of its own on the side stream): tdvc_conv_wgrad_partials
 * runs the first stage only and fills `job` (a HOST struct) with what the second stage needs; tdvc_wgrad_reduce_batch runs the second stage
 * of many layers in ONE launch.  `jobs` and `block_start` (njobs + 1 prefix sums of tdvc_wgrad_reduce_job::nblocks) are DEVICE arrays.
 * Jobs of one batch must write disjoint dw / db ranges (the sums are `+=` without atomics): a caller with several jobs for one parameter
 * (shared layers, per-image launches) issues them in successive batches, which also keeps the summation order fixed.  `work` must stay
 * alive and unmodified until the batch that reduces it has run. */
typedef struct tdvc_wgrad_reduce_job {
  const float* work;
  const float* bwork;
  const int32_t *row_off, *chan_off, *tap_off, *bias_index;
  float *dw, *db;
  float scale;
  int32_t nworkers, cow, ciw, ntaps, cout, cin, wblocks, nblocks;
} tdvc_wgrad_reduce_job;
int tdvc_conv_wgrad_partials(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                             int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                             const int32_t* tap_off, int square_x, float scale, float* dw, const int32_t* bias_index, float* db,
                             float* work, int64_t work_floats, tdvc_wgrad_reduce_job* job, void* stream);
int tdvc_wgrad_reduce_batch(const tdvc_wgrad_reduce_job* jobs, const int32_t* block_start, int njobs, int total_blocks, void* stream);

/* out = g * act'(z), ReLU / LeakyReLU, the sign of z taken from the stored output (y - res); out may alias g. */
int tdvc_act_backward(const tdvc_fmap* g, const tdvc_fmap* y, const tdvc_fmap* res, int act, float slope, const tdvc_fmap* out, void* stream);
/* out[n][Y][X][(i*2+j)*C + c] = y[n][2Y+i][2X+j][c] (adjoint of the PixelShuffle(2) store, packed-row order). */
int tdvc_pixel_unshuffle(const tdvc_fmap* y, const tdvc_fmap* out, void* stream);
/* db[dst_index[c] or c] += scale * sum_{n,h,w} g[n,h,w,c] for c < nvalid (two order-fixed stages through `work`). */
int64_t tdvc_bias_grad_work_floats(int N, int C);
int tdvc_bias_grad(const tdvc_fmap* g, int nvalid, const int32_t* dst_index, float scale, float* db, float* work, int64_t work_floats, void* stream);

/* ---------------------------------------------------------------- backward of the streaming operators (training path)
 * Adjoints of the kernels above; every gradient ACCUMULATES into its destination (tdvc_amd/autograd.py). */
/* dst[..., c] = c < src.C ? src[..., c] : 0 with dtype conversion (fp32 gradients -> the fp16 MFMA backward kernels). */
int tdvc_copy_cast(const tdvc_fmap* src, const tdvc_fmap* dst, void* stream);
/* g *= [0 < y < 1] (the reconstruction clamp, pnet.py:78). */
int tdvc_clamp01_backward(const tdvc_fmap* g, const tdvc_fmap* y, void* stream);
/* y = a * gate[n][c] (SELayer scaling, inflate.py:208): da += g * gate (da may be NULL), dgate[n][c] += sum_pix g * a. */
int64_t tdvc_gate_backward_work_floats(int N, int C);
int tdvc_gate_backward(const tdvc_fmap* g, const tdvc_fmap* a, const float* gate, const tdvc_fmap* da, float* dgate, float* work,
                       int64_t work_floats, void* stream);
/* backward of tdvc_se_gate: dmean[N][C] (written) and the four parameter gradients (+= scale * ...). */
int tdvc_se_gate_backward(const float* partial, int nblocks, float inv_count, int N, int C, int Cmid, const float* w1, const float* b1,
                          const float* w2, const float* b2, const float* gate, const float* dgate, float scale, float* dmean,
                          float* dw1, float* db1, float* dw2, float* db2, void* stream);
/* dx[n][pix][c] += v[n][c] * scale (adjoint of the global average pool). */
int tdvc_bcast_channel_add(const tdvc_fmap* dx, const float* v, float scale, void* stream);
/* adjoint of tdvc_add_flow: dflow[.., 0/1] += sum of the even / odd channels of doff. */
int tdvc_add_flow_backward(const tdvc_fmap* doff, const tdvc_fmap* dflow, void* stream);
/* adjoint of tdvc_bcast_add_act: dx *= lrelu'(x) in place, db += sum over the T channel slices of dx. */
int tdvc_bcast_add_act_backward(const tdvc_fmap* dx, const tdvc_fmap* x, const tdvc_fmap* db, float slope, void* stream);
/* adjoint of tdvc_upsample2x: dx += U^T dy. */
int tdvc_upsample2x_backward(const tdvc_fmap* dy, const tdvc_fmap* dx, void* stream);
/* adjoint of tdvc_resize_bilinear (flownet.py:153-173, the flow's way back from the x32-padded size): dx += R^T (chscale * dy),
 * fp32 maps, gathered per input pixel (reproducible).  Only reached when H or W is not a multiple of 32. */
int tdvc_resize_bilinear_backward(const tdvc_fmap* dy, const tdvc_fmap* dx, const float* chscale, void* stream);

/* backward of tdvc_spynet_level_input (flownet.py:82-140): dflow_up += flow channels of dcat8 + the warp gradient
 * (grid_sample bilinear / border / align_corners=True w.r.t. the flow), then dflow_lo += 2 * U^T dflow_up.  The images
 * need no gradient.  dflow_lo may be NULL (coarsest level). */
int tdvc_spynet_level_input_backward(const tdvc_fmap* supp, const tdvc_fmap* flow_up, const tdvc_fmap* dcat8, const tdvc_fmap* dflow_up,
                                     const tdvc_fmap* dflow_lo, void* stream);

/* flat fp32 helpers of the DCN backward glue: out = sigmoid(x); g *= s*(1-s); dst += scale*src. */
int tdvc_sigmoid_f32(const float* x, float* out, int64_t n, void* stream);
int tdvc_sigmoid_backward_f32(float* g, const float* s, int64_t n, void* stream);
int tdvc_axpy_f32(float* dst, const float* src, float scale, int64_t n, void* stream);

/* backward of tdvc_match_gather (pnet.py:240-255): the matching indices carry no gradient; dfin += d(cat)/d(fin),
 * dfref += the gathered blocks' gradients in gather form (fixed summation order, no atomics). */
int tdvc_match_gather_backward(const tdvc_fmap* fin, const tdvc_fmap* fref, const int32_t* idx, int scale, int hp, int wp,
                               const tdvc_fmap* dcat, const tdvc_fmap* dfin, const tdvc_fmap* dfref, void* stream);

/* GDN / inverse GDN backward, elementwise part (y = x * n^(-1/2) | x * n^(+1/2), n = beta + gamma . x^2 recomputed in
 * fp32): dx += g * n^(-+1/2);  dn = g * x * (-+1/2) * n^(-+1/2 - 1) (fp16, feeds conv_dgrad / conv_wgrad of the norm pool). */
int tdvc_gdn_backward(const tdvc_fmap* g, const tdvc_fmap* x, const tdvc_fmap* n32, int inverse, const tdvc_fmap* dn, const tdvc_fmap* dx, void* stream);
/* dx += 2 * x * t (chain rule through x^2). */
int tdvc_mul2_accumulate(const tdvc_fmap* dx, const tdvc_fmap* x, const tdvc_fmap* t, void* stream);

/* backward of the rate terms in training mode (additive uniform noise), bits = -log2(likelihood):
 * tdvc_eb_backward: dz += gscale * dbits/dz, dparams[C][59] += gscale * dbits/d(packed parameters);
 * tdvc_gc_backward: dy += gscale * dbits/dy, dgp[.., 0:M] (scales) and dgp[.., M:2M] (means) likewise. */
int tdvc_eb_backward(const tdvc_fmap* z, const float* params, const tdvc_fmap* noise, float gscale, const tdvc_fmap* dz, float* dparams, void* stream);
int tdvc_gc_backward(const tdvc_fmap* y, const tdvc_fmap* gp, const tdvc_fmap* noise, float gscale, const tdvc_fmap* dy, const tdvc_fmap* dgp, void* stream);
/* The factorised prior's parameter-space work of a training step (compressai EntropyBottleneck: `_logits_cumulative`'s softplus / tanh
 * reparametrisation and `loss()`, the auxiliary quantile loss; the reference's tools/train.py:150-151 runs the latter through autograd).
 * `raw` / `grad`: DEVICE arrays of 14 pointers -- _matrix0..4, _bias0..4, _factor0..3, each [C][len] fp32 -- in the column order of the
 * packed table [C][59] that tdvc_eb_forward / tdvc_eb_backward read.
 * tdvc_eb_pack: packed = softplus(matrices) | biases | tanh(factors) | median (quantiles[c][1]).
 * tdvc_eb_param_chain: grad[k] += scale * dpacked * d packed / d raw (the median column is skipped).
 * tdvc_eb_aux: loss[0] = sum |logits_cumulative(quantiles) - (-target, 0, +target)|, dq[C][3] = its gradient (overwritten); 3 C <= 1024. */
int tdvc_eb_pack(const float* const* raw, const float* quantiles, float* packed, int C, void* stream);
int tdvc_eb_param_chain(const float* dpacked, const float* const* raw, float* const* grad, float scale, int C, void* stream);
int tdvc_eb_aux(const float* params, const float* quantiles, float target, float* dq, float* loss, int C, void* stream);

/* fp16 channel-innermost pieces of the fused DCN's backward (dcn_v2_cuda.cu:97-216 restructured): column channel
 * k = group*72 + tap*8 + j.  tdvc_dcn_columns: col = mask * bilinear samples (operand of dW = dY col^T, a 1x1
 * tdvc_conv_wgrad).  tdvc_dcn_col2im: from dcol = W^T dY (a 1x1 tdvc_conv2d): offset / mask gradients accumulate
 * into dom (fp16, the offset-mask conv's output gradient, mask through its sigmoid), the bilinear scatter into
 * dx32 (fp32 [N][H][W][8G], zero-initialised by the caller).  The scatter is privatised: one 32x32-pixel LDS window
 * per 16x16 tile and group, stored to `work` (tdvc_dcn_col2im_work_floats floats, 16-byte aligned) and summed per
 * pixel in a fixed order; only samples displaced by more than 8 pixels use float atomics (the reference's col2im
 * uses them for every sample). */
int tdvc_dcn_columns(const tdvc_fmap* x, const tdvc_fmap* om, int groups, const tdvc_fmap* col, void* stream);
int64_t tdvc_dcn_col2im_work_floats(int N, int H, int W, int groups);
int tdvc_dcn_col2im(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                    float* work, int64_t work_floats, void* stream);
/* Run-to-run reproducible form (tdvc_amd.ops.DETERMINISTIC; the reproducible training run of the parity tests): samples
 * displaced out of their tile's window are recorded instead of added with float atomics -- far_keys[i] (unique: target
 * (n, y, x, group) << 27 | source pixel * 36 + tap * 4 + corner), far_vals[i][8]; *far_count (device, zeroed by the caller)
 * counts every such sample, records beyond far_cap are dropped (the caller checks the count).  The caller sorts the keys
 * (any stable device sort; `order` = the permutation) and tdvc_dcn_far_apply adds each target's records in key order. */
int tdvc_dcn_col2im_det(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                        float* work, int64_t work_floats, int64_t* far_keys, float* far_vals, int32_t* far_count, int32_t far_cap, void* stream);
int tdvc_dcn_far_apply(const int64_t* keys_sorted, const int64_t* order, const float* far_vals, int32_t count, float* dx32, void* stream);

/* ---------------------------------------------------------------- quality metrics (evaluation loop)
 * One level of MS-SSIM as main/model/ms_ssim_torch.py:33-83 computes it on a float32 NCHW pair: valid separable
 * filter `win` (HOST array of win_size taps, odd, <= 15) of X, Y, X^2, Y^2, XY; ssim_out[n] / cs_out[n] = the means of
 * the ssim / cs maps over (C, H - win + 1, W - win + 1), BEFORE the reference's (v + 1) / 2.  x, y, outputs and `work`
 * (tdvc_ssim_level_work_floats floats) are device memory.  tdvc_avgpool2_pad_f32 is the pooling between levels
 * (:178-180): kernel 2, stride 2, padding (H % 2, W % 2) with the zeros counted; out is
 * [planes][(H + 2*(H%2) - 2) / 2 + 1][(W + 2*(W%2) - 2) / 2 + 1]. */
int64_t tdvc_ssim_level_work_floats(int N, int C, int H, int W, int win_size);
int tdvc_ssim_level(const float* x, const float* y, int N, int C, int H, int W, const float* win, int win_size,
                    float c1, float c2, float* ssim_out, float* cs_out, float* work, int64_t work_floats, void* stream);
int tdvc_avgpool2_pad_f32(const float* x, int64_t planes, int H, int W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDVC_HIP_H */
