/*
 * flowse_hip.h -- C ABI of libflowse_hip.so: the MI355X (gfx950) implementation of the flowmse sampling
 * hot path (Euler ODE loop x NCSN++ vector field).
 *
 * Conventions
 *   - Every function returns an int status (0 = FLOWSE_OK); flowse_last_error() returns a thread-local,
 *     human readable message for the last non-zero status.  Nothing throws across the ABI.
 *   - All tensor arguments are RAW DEVICE POINTERS owned by the caller (e.g. torch.Tensor.data_ptr());
 *     weights and workspace are owned by the model handle.  Exception: arguments documented "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work and return;
 *     they never synchronise the device (flowse_model_load_weights and flowse_model_reserve may allocate).
 *     The NULL (legacy default) stream cannot be captured into a hipGraph, so with graph replay enabled (FLOWSE_GRAPH=1)
 *     and stream == NULL the model handle runs the call on an internal stream fenced by events against the NULL stream on
 *     both sides: the call is ordered after
 *     everything enqueued on the NULL stream before it, and later NULL-stream work is ordered after the call -- the same
 *     ordering a launch on the NULL stream itself would have had (PyTorch's default stream is the NULL stream).
 *   - One handle per GPU per process; calls on one handle must be serialised by the caller (the reference is
 *     single-threaded, single-stream, torch.no_grad()).
 *   - Boundary tensors follow the reference: complex64 interleaved (re, im), [B, 1, F, T] contiguous
 *     (flowmse/backbones/ncsnpp.py:402-403), F == image_size, T a multiple of 2^(levels-1) (pad_spec,
 *     flowmse/util/other.py:83-90); time t is float32 [B] in (0, 1].
 *   - Per-op entry points (flowse_op_*) use the library's internal activation layout NHWC float32
 *     [B][H][W][C]; they exist for unit parity tests and for callers that fuse their own graphs.
 *
 * The reference has no C ABI for this path except the pybind11 `upfirdn2d` operator
 * (flowmse/backbones/ncsnpp_utils/op/upfirdn2d.cpp:12-22); flowse_upfirdn2d() is its drop-in.  All other
 * entry points replace PyTorch module calls; each one cites the reference code it stands for.
 */
#ifndef FLOWSE_HIP_H
#define FLOWSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOWSE_OK 0
#define FLOWSE_ERR_ARG 1
#define FLOWSE_ERR_HIP 2
#define FLOWSE_ERR_STATE 3
#define FLOWSE_ERR_SHAPE 4

#define FLOWSE_ABI_VERSION 3
#define FLOWSE_MAX_LEVELS 8
#define FLOWSE_MAX_ATTN 4

/* Constructor arguments of flowmse.backbones.ncsnpp.NCSNpp that shape the graph (ncsnpp.py:45-67).
 * The remaining constructor flags are fixed at the reference defaults (biggan blocks, FIR [1,3,3,1],
 * skip_rescale, output_skip / input_skip progressive branches, 'sum' combine, fourier embedding). */
typedef struct flowse_config {
    int32_t nf;                                /* 128 */
    int32_t num_levels;                        /* len(ch_mult) = 7 */
    int32_t ch_mult[FLOWSE_MAX_LEVELS];        /* (1,1,2,2,2,2,2) */
    int32_t num_res_blocks;                    /* 2 */
    int32_t num_attn;                          /* len(attn_resolutions) = 1 */
    int32_t attn_resolutions[FLOWSE_MAX_ATTN]; /* (16,) */
    int32_t image_size;                        /* 256 = number of frequency bins F */
} flowse_config;

typedef struct flowse_model flowse_model;      /* opaque */

int flowse_abi_version(void);
const char* flowse_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int flowse_device_count(void);

/* ---- model handle -------------------------------------------------------------------------------
 * flowse_model_create builds the module list of NCSNpp.__init__ (ncsnpp.py:97-245) and its parameter
 * table on the host only -- no device is touched, so it also works on a CPU-only machine. */
int flowse_model_create(const flowse_config* cfg, flowse_model** out);
void flowse_model_destroy(flowse_model* m);

/* ---- single-module handles (unit parity against the reference's modules) -------------------------------------
 * A handle that holds ONE module of the network behind the same weight packer, launch planner and kernels the full
 * model uses: FLOWSE_BLOCK_RESNET = ResnetBlockBigGANpp(in_ch, out_ch, up, down) (layerspp.py:212-274),
 * FLOWSE_BLOCK_ATTN = AttnBlockpp(channels = in_ch = out_ch) (layerspp.py:62-91), FLOWSE_BLOCK_COMBINE =
 * Combine(4 -> out_ch, 'sum') (layerspp.py:44-59).  The parameter table / canonical blob / flowse_model_load_weights
 * calls work as for a model handle, with the reference's module-local keys under "all_modules.0." (GroupNorm_0.weight,
 * Conv_0.weight, Dense_0.weight, NIN_0.W, ...).
 * flowse_block_forward: NHWC float32 device tensors.  RESNET: out = block(cat[in1, in2], temb) with in1 [B,H,W,C1],
 * in2 [B,H,W,in_ch-C1] or NULL (then C1 = in_ch); `temb_act` = SiLU(temb) [B][temb_dim] (the block applies Dense_0 to
 * it, layerspp.py:262-263); out [B,H',W',out_ch] (H' = 2H / H/2 for up / down).  ATTN: out = block(in1).
 * COMBINE: out = Conv_0(in1 [B,H,W,4]) + in2 [B,H,W,out_ch]. */
#define FLOWSE_BLOCK_RESNET 0
#define FLOWSE_BLOCK_ATTN 1
#define FLOWSE_BLOCK_COMBINE 2
int flowse_block_create(int kind, int in_ch, int out_ch, int up, int down, int temb_dim, flowse_model** out);
int flowse_block_forward(flowse_model* m, const float* in1, int C1, const float* in2, const float* temb_act, float* out,
                         int B, int H, int W, void* stream);

/* Parameter table in the order of NCSNpp.parameters() / state_dict() (the order torch_ema's shadow_params
 * use, flowmse/model.py:81-103): output_layer.{weight,bias}, then all_modules.{i}.*.  `name` receives the
 * reference state_dict key; shape[0..ndim) the reference shape; `offset` the element offset of the tensor in
 * the canonical weight blob (all tensors contiguous, reference layout, back to back in table order). */
int flowse_model_num_params(const flowse_model* m);
int flowse_model_num_modules(const flowse_model* m);
int64_t flowse_model_blob_numel(const flowse_model* m);
int flowse_model_param_info(const flowse_model* m, int index, char* name, int name_cap, int64_t shape[4],
                            int* ndim, int64_t* offset);

/* Upload weights.  `blob` is a HOST pointer to flowse_model_blob_numel() floats in canonical order.  The
 * library re-packs into its kernel-native layouts (channel-last conv weights, transposed NIN, fused q/k/v,
 * one stacked Dense_0 matrix) on the current HIP device. */
int flowse_model_load_weights(flowse_model* m, const float* blob, int64_t numel);

/* Precision mode (call BEFORE flowse_model_load_weights; a change drops the uploaded weights).
 * 0 (default): every operand, product and accumulation is fp32 (v_mfma_f32_32x32x2_f32), activations fp32.  3x3
 *   convolutions with Cin % 32 == 0 and Cout % 64 == 0 on images the LDS-halo kernels cover are evaluated in Winograd
 *   form on transformed fp32 operands: whole-K launches of >= 128 blocks of 16 x 16 pixels x 64 channels in the
 *   two-dimensional form F(4,3) vertical x F(2,3) horizontal (conv3x3_w2d_kernel: one third of the multiplies, 3.4e-7 ..
 *   8.8e-7 rel-L2 per layer against the fp64 convolution), the rest in F(4,3) along the vertical axis only
 *   (conv3x3_f43_kernel: half the multiplies, 4e-7 .. 2e-6); FLOWSE_W2D=0 keeps everything on the one-dimensional form,
 *   FLOWSE_NO_WINOGRAD=1 selects the direct form, whose result is bit for bit an fmaf chain.
 * 1 "bf16x3": fp32 activations; the operands of the big 3x3 convs are split x = hi + lo in bf16 and the products
 *   hi*hi + hi*lo + lo*hi accumulated in fp32 (fp32-class accuracy, ~1e-5 end to end).
 * 2 "bf16" (BASELINE config 3) / 3 "fp16" (BASELINE config 5): 16-bit STORAGE modes -- every wide activation tensor
 *   between kernels is bf16 / IEEE half, all matrix products run on the 16-bit matrix cores (a 16-bit twin of the packed
 *   weights is kept) -- including the attention core (16-bit q / k / v and P, v_mfma_f32_32x32x16, attention16_kernel)
 *   -- while accumulators, GroupNorm statistics, the softmax state (running max / sum), time-embedding tables, split-K
 *   slabs and the 4-channel pyramid tensors stay fp32.  Applies to networks whose wide channel counts are
 *   multiples of 32 (the released configuration); otherwise storage stays fp32 and only the operands of the 3x3 convs
 *   with Cout % 128 == 0 become 16-bit.
 * The boundary tensors (x, y, out: complex64; t: float32) are the same in every mode. */
int flowse_model_set_precision(flowse_model* m, int mode);

/* Optional: plan buffers for a shape ahead of time (otherwise done lazily by the first call).
 * `workspace_bytes` (may be NULL) receives the activation workspace size. */
int flowse_model_reserve(flowse_model* m, int B, int F, int T, int64_t* workspace_bytes);

/* ---- vector field --------------------------------------------------------------------------------
 * mode 0: out = dnn(cat[x, y], t)     == NCSNpp.forward          (ncsnpp.py:247-404)
 * mode 1: out = -dnn(cat[x, y], t)    == VFModel.forward(x,t,y)  (flowmse/model.py:164-170)
 * x, y, out: complex64 [B,1,F,T] device; t: float32 [B] device. */
int flowse_vf_forward(flowse_model* m, const void* x, const void* y, const float* t, void* out, int B, int F,
                      int T, int mode, void* stream);

/* ---- sampler --------------------------------------------------------------------------------------
 * x <- y + sigma * z                                            (FLOWMATCHING.prior_sampling, odes.py:93-100) */
int flowse_prior_sample(const void* y, const void* z, float sigma, void* x_out, int64_t numel_complex,
                        void* stream);
/* N Euler steps in place on x (ode_solver loop, flowmse/sampling/__init__.py:45-57, with
 * EulerODEsolver.update_fn, sampling/odesolvers.py:42-47):  for i: x <- x + VF(x, ts[i], y) * (-dts[i]).
 * ts, dts: HOST float32 arrays of length N (the caller reproduces torch.linspace and the step rule, including
 * the final step dts[N-1] = ts[N-1]); they are consumed before the call returns (passed to the device as kernel
 * arguments, no asynchronous host copy).  No host synchronisation.  The launch list of a shape holds no per-call
 * argument; with FLOWSE_GRAPH=1 (read at flowse_model_create) it is captured on its second use and each network
 * evaluation becomes one hipGraph launch, counted by flowse_model_graph_launches().  The default is plain launches from
 * this C loop: measured on MI355X / ROCm 7.2 the replay is 5 % slower at [1,1,256,256] and equal at [8,1,256,256]. */
int flowse_euler_sample(flowse_model* m, void* x_inout, const void* y, const float* ts, const float* dts, int N,
                        int B, int F, int T, void* stream);
/* The same loop with a fixed-step explicit Runge-Kutta update per grid step (BASELINE config 5's "N = 25 RK solver").
 * The reference has no fixed-step RK -- its only Runge-Kutta is scipy's adaptive RK45 black box
 * (flowmse/sampling/__init__.py:64-114) -- so this is the reference's white-box loop (sampling/__init__.py:45-57, same
 * grid and step rule) with the update of an ODEsolverRegistry plugin (sampling/odesolvers.py:9-34) in place of
 * EulerODEsolver.update_fn.  tableau: FLOWSE_TABLEAU_EULER (== flowse_euler_sample), _HEUN (explicit trapezoid, 2
 * network evaluations per step) or _RK4 (classical, 4 per step).  A step that ends at t = 0 -- the last step of the
 * reference's grid -- is taken as the reference's Euler update: the field divides by t (ncsnpp.py:398) and embeds
 * log t, so no stage is ever evaluated at t <= 0.  Stages are chained through the head kernel (next stage input and
 * slope accumulation fused into it): no extra launches, no host synchronisation; each evaluation is the shape's launch
 * list issued as plain launches (default) or, under FLOWSE_GRAPH=1, one hipGraph launch. */
#define FLOWSE_TABLEAU_EULER 0
#define FLOWSE_TABLEAU_HEUN 1
#define FLOWSE_TABLEAU_RK4 2
int flowse_rk_sample(flowse_model* m, void* x_inout, const void* y, const float* ts, const float* dts, int N,
                     int tableau, int B, int F, int T, void* stream);
/* Number of hipGraphLaunch calls this handle has issued so far (0 while every evaluation ran as plain launches). */
int64_t flowse_model_graph_launches(const flowse_model* m);
/* One generic explicit update from a caller-held slope: x <- x + dt * k (complex64 as float pairs). */
int flowse_axpy(const void* x, const void* k, float dt, void* out, int64_t numel_complex, void* stream);

/* ---- spectrogram transforms either side of the sampler -------------------------------------
 * flowse_stft_compress: sig float32 [B][L] -> complex64 [B,1,256,Tpad].  Equals pad_spec(spec_fwd(stft(sig * scale_in)))
 * of the reference (data_module.py:149-162,199-201; util/other.py:83-90) for n_fft 510, hop 128, periodic hann,
 * center=True (reflect): T = L / 128 + 1 frames, frames T..Tpad-1 are zero.  spec_fwd = factor * |z|^exponent *
 * exp(j arg z) (transform_type "exponent"; exponent 1 = plain scaling).
 * flowse_istft_decompress: the inverse chain istft(spec_back(spec), length = Lout) * scale_out
 * (data_module.py:164-175,203-205; model.py:190-191) over frames 0..T-1 of a spectrogram whose frame pitch is Tpad.
 * The reference runs the iSTFT over ALL frames of the padded sample (model.py:190-191, evaluate.py:132), i.e.
 * T == Tpad there: the zero-padded frames are no longer zero after enhancement and reach the last ~127 samples. */
int flowse_stft_compress(const float* sig, int B, int L, float scale_in, void* out_c64, int T, int Tpad, float factor,
                         float exponent, void* stream);
int flowse_istft_decompress(const void* spec_c64, int B, int T, int Tpad, float factor, float exponent, float* out,
                            int Lout, float scale_out, void* stream);

/* ---- in-library kernel timing (used by bench.py for the live roofline figure) -------------------------
 * Between _begin and _end every selected launch of this handle is bracketed by HIP events on the launch
 * stream (and the handle launches eagerly instead of replaying its hipGraph).  mode 0: only launches of the dominant
 * kernel -- the 3x3 ResBlock convolutions with fused GroupNorm+SiLU input and Cout > 64 (conv3x3_w2d_kernel<2, .> in
 * the fp32 mode, conv3x3_pc16_kernel<2, ., .> in the 16-bit storage modes) -- reported under the key
 * "dominant_conv3x3"; mode 1: every launch, keyed by op label.  _end synchronises
 * on the recorded events and writes a JSON object {label: {"launches", "ms", "flops", "bytes", "issued"}} into `json`:
 * algorithmic flops / bytes of the bracketed launches, and `issued` = the flops the matrix cores execute for them
 * (1/3 of the algorithmic direct-convolution flops per launch of the two-dimensional Winograd form, 1/2 per launch of
 * the one-dimensional F(4,3) form, all of them otherwise).  The extra key "_all_launches"
 * totals launches / flops / issued over EVERY launch made between _begin and _end (no timing). */
int flowse_profile_begin(flowse_model* m, int mode);
int flowse_profile_end(flowse_model* m, char* json, int cap);

/* ---- upfirdn2d: drop-in for the reference's native operator ------------------------------------------
 * Reference: upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 * (op/upfirdn2d.cpp:12-22; kernels op/upfirdn2d_kernel.cu:49-207).  input: float32 [planes, in_h, in_w]
 * (= NCHW with planes = N*C), kernel: float32 [kh, kw] device, out: [planes, out_h, out_w] with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh) / down_y + 1 (same for w).  Pads must be >= 0. */
int flowse_upfirdn2d(const float* input, const float* kernel, int planes, int in_h, int in_w, int kh, int kw,
                     int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                     float* out, int out_h, int out_w, void* stream);

/* ---- per-op entry points (NHWC float32 device tensors) ------------------------------------------------
 * conv: out = (conv_{taps}(cat[in1,in2]; w) + bias + bias2[b] + res) * scale.  w: [Cout][taps][C1+C2] (channel
 * last), taps 9 (3x3, pad 1) or 1; in2/bias/bias2/res may be NULL.  Stands for ddpm_conv3x3 / ddpm_conv1x1 /
 * NIN (layers.py:100-124, 546-555) with the ResnetBlockBigGANpp epilogue (layerspp.py:262-274). */
int flowse_op_conv2d(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                     const float* bias2, int bias2_stride, const float* res, float* out, int B, int H, int W,
                     int Cout, int taps, float scale, float* splitk_scratch, void* stream);
/* Small images are computed split-K (K sliced over extra thread blocks, deterministic two-pass reduction) when
 * `splitk_scratch` holds flowse_op_conv2d_scratch_floats(...) floats (0 = this shape never splits); with
 * splitk_scratch == NULL the single-pass kernel is used. */
int64_t flowse_op_conv2d_scratch_floats(int B, int H, int W, int Cin, int Cout, int taps);
/* The same contract on the 16-bit matrix cores with 16-bit activation storage (BASELINE configs 3 / 5; dt 1 = bf16,
 * 2 = IEEE half): the fp32 tensors are rounded to dt on the way in, the conv runs as in the 16-bit modes of the model
 * handle (producer / consumer LDS-halo kernel for the 3x3 shapes it covers -- its fragment-order copy of the weights is
 * made here per call --, per-tap halo kernel or flat kernel (+ split-K) otherwise), the result is widened back.
 * Optional fused GroupNorm(+SiLU) of the input from per-(sample, channel) gn_mean / gn_scale [B][C1+C2] and gn_beta
 * [C1+C2] (LDS-halo shapes only).  `scratch`: device memory; sufficient for every shape: 2*(in + 2*w + res + out
 * elements) + 4*ksplit*out elements + 4 KB bytes (each of the up to seven sub-buffers is rounded up to 256 bytes; the
 * error message of a too-small call states the exact byte count).  Cout == 4 with taps == 9 on >= 64 tiles of
 * 16 x 16 pixels is the progressive-output head (ncsnpp.py:345-366): 16-bit input, but `res` (the pyramid) and `out` stay fp32
 * as in the model (conv3x3_head4_16_kernel; fused GroupNorm input allowed). */
int flowse_op_conv2d_16(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                        const float* res, const float* gn_mean, const float* gn_scale, const float* gn_beta, int silu,
                        float* out, int B, int H, int W, int Cout, int taps, float scale, int dt, void* scratch,
                        int64_t scratch_bytes, void* stream);
/* Tail of ResnetBlockBigGANpp (layerspp.py:265-274) in 16-bit storage as ONE launch of the producer / consumer kernel:
 *   out = (conv3x3(act(GroupNorm(h)); w1) + b1 + conv1x1(cat[x1, x2]; w2) + b2) * scale
 * The 1x1 shortcut Conv_2(x) runs as extra K steps of Conv_1's launch (the form the 16-bit model modes use wherever
 * Conv_1 is on that kernel; FLOWSE_NO_SCFOLD=1 at plan time restores the separate launch).  h: [B][H][W][C] with per-
 * (sample, channel) gn_mean / gn_scale [B][C] and gn_beta [C] (NULL: no normalisation); w1 [Cout][9][C]; x1 / x2 (NULL)
 * the shortcut's input channels (XC1, XC2 multiples of 32, >= 96 in all), w2 [Cout][1][XC1 + XC2].  Shapes: H, W
 * multiples of 16, C % 32 == 0, Cout % 128 == 0, at least 64 (16 x 16 tile, 128-channel block) items, else
 * FLOWSE_ERR_SHAPE.  `scratch`: 2 * (h + x1 + x2 + 2 w1 + 2 w2 + out elements) + 4 KB bytes. */
int flowse_op_resblock_tail_16(const float* h, int C, const float* gn_mean, const float* gn_scale, const float* gn_beta,
                               int silu, const float* w1, const float* b1, const float* x1, int XC1, const float* x2,
                               int XC2, const float* w2, const float* b2, float* out, int B, int H, int W, int Cout,
                               float scale, int dt, void* scratch, int64_t scratch_bytes, void* stream);
/* Test / A-B hook, process-wide: channel-block width of conv3x3_pc16_kernel.  -1 (default): 128-channel blocks, 64-channel
 * blocks for launches with fewer than 3/4 of a (16 x 16 tile, 128-channel block) item per compute unit; 0: always 128;
 * 1: always 64 (also FLOWSE_PC16_NARROW=0 / 1 in the environment).  Results differ only in summation grouping of the
 * GroupNorm partial statistics (the convolution sums are identical). */
int flowse_op_pc16_channel_blocks(int mode);
/* Fused ResnetBlock half:  out = (conv3x3(act(GroupNorm(cat[in1,in2]))) + bias + bias2[b] + res) * scale
 * (layerspp.py:246-249 / :265-267) with the normalisation + SiLU applied while the input tile is staged into LDS.
 * Only for shapes the halo kernel covers (H % 8 == 0, W % 16 == 0, C1 % 32 == 0, C2 % 32 == 0, image large
 * enough not to be split-K): otherwise FLOWSE_ERR_SHAPE.  `scratch`: flowse_op_group_norm_scratch_floats(). */
int flowse_op_conv3x3_gn(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                         float eps, int silu, const float* w, const float* bias, const float* bias2,
                         int bias2_stride, const float* res, float* out, int B, int H, int W, int Cout, float scale,
                         float* scratch, void* stream);
/* The same fused ResnetBlock half computed with the F(4,3) Winograd form of the 3x3 filter along its vertical axis: 6
 * multiplies per four outputs instead of 12 (half of the direct-convolution FLOPs on the matrix cores; fp32 error ~3x the
 * direct sum's) -- the kernel the model handle uses for every 3x3 convolution with Cout % 64 == 0 on images large enough
 * for the halo tiling (FLOWSE_NO_WINOGRAD=1 selects the direct kernel).  gamma == NULL: plain convolution without the
 * GroupNorm + SiLU input stage.  `w` is the packed [Cout][9][Cin] weight as for flowse_op_conv2d; the transformed
 * weights are derived into `scratch` (flowse_op_conv3x3_f43_scratch_floats floats).  Other shapes: FLOWSE_ERR_SHAPE. */
int flowse_op_conv3x3_f43(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2,
                          int bias2_stride, const float* res, float* out, int B, int H, int W, int Cout, float scale,
                          float* scratch, void* stream);
int64_t flowse_op_conv3x3_f43_scratch_floats(int B, int H, int W, int C, int Cout);
/* The same contract in the TWO-dimensional Winograd form F(4,3) (vertical) x F(2,3) (horizontal): 24 multiplies per
 * 4 x 2 output patch instead of 72 (one third of the direct-convolution FLOPs on the matrix cores; fp32 error ~1.5x the
 * 1-D form's).  Covers H % 16 == 0, W % 16 == 0, channel counts multiples of 32, Cout % 64 == 0; other shapes:
 * FLOWSE_ERR_SHAPE.  The model handle uses it for launches of at least 128 blocks of 16 x 16 pixels x 64 channels (below
 * 256 of them with 32-channel blocks, so that every CU gets one) unless FLOWSE_W2D=0. */
int flowse_op_conv3x3_w2d(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2,
                          int bias2_stride, const float* res, float* out, int B, int H, int W, int Cout, float scale,
                          float* scratch, void* stream);
int64_t flowse_op_conv3x3_w2d_scratch_floats(int B, int H, int W, int C, int Cout);
/* GroupNorm(min(C/4,32) groups, eps) [+ SiLU] over cat[in1,in2] (layerspp.py:219,231; ncsnpp.py:337).
 * `scratch` must hold flowse_op_group_norm_scratch_floats(B,H*W,C1+C2) floats. */
int64_t flowse_op_group_norm_scratch_floats(int B, int HW, int C);
int flowse_op_group_norm(const float* in1, int C1, const float* in2, int C2, const float* gamma,
                         const float* beta, float eps, int silu, float* out, int B, int H, int W, float* scratch,
                         void* stream);
/* FIR x2 resampling with [1,3,3,1] (up_or_down_sampling.py:195-257). up: out [B,2H,2W,C]; down: [B,H/2,W/2,C] */
int flowse_op_fir_up(const float* in, float* out, int B, int H, int W, int C, void* stream);
int flowse_op_fir_down(const float* in, float* out, int B, int H, int W, int C, void* stream);
/* softmax(q k^T C^-1/2) v over L tokens; qkv [B][L][3C], out [B][L][C] (layerspp.py:82-86). */
int flowse_op_attention(const float* qkv, float* out, int B, int L, int C, void* stream);
/* GaussianFourierProjection(log t) (layerspp.py:39-41, ncsnpp.py:259): out [B][2E] */
int flowse_op_gfp(const float* t, const float* W, float* out, int B, int E, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWSE_HIP_H */
