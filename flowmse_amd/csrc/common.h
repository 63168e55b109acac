// Shared declarations for the flowse HIP library (gfx950 / MI355X only).
//
// Internal activation layout is NHWC fp32: x[b][h][w][c] with c contiguous, so
// the channel axis is the contiguous GEMM-K axis of the implicit-GEMM
// convolutions and every normalisation / resampling kernel is a float4 stream
// over channels.  h = frequency bin (F), w = STFT frame (T).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flowse {

// status codes of the C ABI (include/flowse_hip.h)
enum { OK = 0, ERR_ARG = 1, ERR_HIP = 2, ERR_STATE = 3, ERR_SHAPE = 4 };

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define FLOWSE_HIP(call)                                                        \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return ::flowse::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define FLOWSE_LAUNCH_CHECK()                                                   \
    do {                                                                        \
        hipError_t _e = hipGetLastError();                                      \
        if (_e != hipSuccess) return ::flowse::hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------ activation storage types
// Activations between kernels are NHWC tensors of float (default) or, in the 16-bit precision modes, bf16 / IEEE half
// (BASELINE configs 3 / 5: half the HBM traffic of every HBM-bound kernel).  All arithmetic on them is fp32: a kernel
// widens on load and rounds (to nearest even) once on store; GroupNorm statistics are taken over the ROUNDED values,
// i.e. over exactly what the consumer will read.  The 4-channel tensors (input pack, input / output pyramids), all
// statistics, time-embedding tables and split-K partial slabs stay fp32 in every mode.
enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
struct bf16_t { unsigned short v; };
struct f16_t { unsigned short v; };
inline int dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

template <class ST> struct St;
template <> struct St<float> {
    static constexpr int dt = DT_F32;
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ float4 rnd4(float4 v) { return v; }      // value as it will be read back
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct St<bf16_t> {
    static constexpr int dt = DT_BF16;
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        // round to nearest even; ONE instruction (two scalar __bf16 casts lower to two conversions plus SDWA / v_bitop3 merges)
        unsigned r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ void unpack2(unsigned w, float& a, float& b) {
        a = __uint_as_float(w << 16);
        b = __uint_as_float(w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, float4 v) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
    }
    static __device__ __forceinline__ float4 rnd4(float4 v) {
        const unsigned a = pack2(v.x, v.y), b = pack2(v.z, v.w);
        return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16),
                           __uint_as_float(b & 0xffff0000u));
    }
    static __device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((unsigned)p->v << 16); }
    static __device__ __forceinline__ void st1(bf16_t* p, float v) {
        const __bf16 x = (__bf16)v;
        p->v = __builtin_bit_cast(unsigned short, x);
    }
};
template <> struct St<f16_t> {
    static constexpr int dt = DT_F16;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        // round to nearest even in ONE instruction (gfx950); bit-identical to two (_Float16) casts over all 2^32 fp32
        // patterns in both slots (tools/probes/cvt_pk_f16.hip)
        unsigned r;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ void unpack2(unsigned w, float& a, float& b) {
        a = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
        b = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
    }
    static __device__ __forceinline__ float4 ld4(const f16_t* p) {
        const h4 r = *reinterpret_cast<const h4*>(p);
        return make_float4((float)r.x, (float)r.y, (float)r.z, (float)r.w);
    }
    static __device__ __forceinline__ void st4(f16_t* p, float4 v) {
        h4 r = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<h4*>(p) = r;
    }
    static __device__ __forceinline__ float4 rnd4(float4 v) {
        return make_float4((float)(_Float16)v.x, (float)(_Float16)v.y, (float)(_Float16)v.z, (float)(_Float16)v.w);
    }
    static __device__ __forceinline__ float ld1(const f16_t* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
    static __device__ __forceinline__ void st1(f16_t* p, float v) {
        const _Float16 x = (_Float16)v;
        p->v = __builtin_bit_cast(unsigned short, x);
    }
};
// 16 bytes of consecutive elements: 4 floats, or 8 x 16 bit (widened into v[0..7])
template <class ST> struct Vec16 { static constexpr int N = 16 / (int)sizeof(ST); };
template <class ST>
__device__ __forceinline__ void ld16(const ST* p, float* v) {
    if constexpr (sizeof(ST) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (St<ST>::dt == DT_BF16) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            } else {
                v[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xffffu));
                v[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16));
            }
        }
    }
}
// stores v[0..N) rounded to ST and leaves the rounded values (as read back later) in v
template <class ST>
__device__ __forceinline__ void st16_round(ST* p, float* v) {
    if constexpr (sizeof(ST) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (St<ST>::dt == DT_BF16) {
                w[i] = St<bf16_t>::pack2(v[2 * i], v[2 * i + 1]);
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            } else {
                const _Float16 a = (_Float16)v[2 * i], b = (_Float16)v[2 * i + 1];
                w[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
                v[2 * i] = (float)a;
                v[2 * i + 1] = (float)b;
            }
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// run `F.template operator()<ST>()` for the storage type tagged dt
#define FLOWSE_DT_SWITCH(dt, ST, ...)                                          \
    switch (dt) {                                                              \
        case ::flowse::DT_F32: { using ST = float; __VA_ARGS__; } break;       \
        case ::flowse::DT_BF16: { using ST = ::flowse::bf16_t; __VA_ARGS__; } break; \
        default: { using ST = ::flowse::f16_t; __VA_ARGS__; } break;           \
    }

// GroupNorm parameters resolved per (sample, channel) by gn_finalize:
//   y = (x - mean[b][c]) * scale[b][c] + beta[c],  scale = rstd * gamma
struct GnParams {
    const float* mean;   // [B][C]
    const float* scale;  // [B][C]
    const float* beta;   // [C]
};

// ------------------------------------------------------------------ GroupNorm partial statistics
// Every producer of a tensor that a GroupNorm will consume emits, per (sample, pixel block, channel), the block's
// MEAN and its centred second moment M2 = sum (x - mean)^2 -- never raw sum / sum of squares, whose difference
// cancels catastrophically in fp32 when |mean| >> std.  Threads accumulate about a pivot (their first value),
// partials are merged with Chan's parallel formula, and gn_finalize merges blocks and channels in fp64.
// Layout: partial[((b * nblk + blk) * C + c) * 2 + {0: mean, 1: M2}], block blk covering pixels
// [blk * ppb, min(HW, (blk + 1) * ppb)).
struct Stat4 {
    float4 p, s1, s2;
    int n;
    __device__ __forceinline__ void init() {
        n = 0;
        p = s1 = s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void add(const float4& v) {
        if (n == 0) p = v;
        const float dx = v.x - p.x, dy = v.y - p.y, dz = v.z - p.z, dw = v.w - p.w;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
        ++n;
    }
    // out8 = {mean.xyzw, M2.xyzw}
    __device__ __forceinline__ void finish(float* out8) const {
        const float inv = n > 0 ? 1.f / (float)n : 0.f;
        const float a[4] = {s1.x, s1.y, s1.z, s1.w}, q[4] = {s2.x, s2.y, s2.z, s2.w}, pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out8[j] = pv[j] + a[j] * inv;
            out8[4 + j] = fmaxf(q[j] - a[j] * a[j] * inv, 0.f);
        }
    }
};
// merge block B (nB, meanB, M2B) into A, per channel of a quad; acc8 / in8 = {mean.xyzw, M2.xyzw}
__device__ __forceinline__ void chan_merge4(float& nA, float* acc8, float nB, const float* in8) {
    if (nB <= 0.f) return;
    const float n = nA + nB, w = nB / n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = in8[j] - acc8[j];
        acc8[j] += d * w;
        acc8[4 + j] += in8[4 + j] + d * d * nA * w;
    }
    nA = n;
}

// ------------------------------------------------------------------ conv (implicit GEMM, MFMA fp32)
struct ConvArgs {
    // in1 / in2 hold elements of type in_dt, res / out of type out_dt (declared float*: the fp32 kernels use them as
    // is, the 16-bit kernels reinterpret them); everything else is fp32
    const float* in1;   // [B][H][W][C1]
    const float* in2;   // [B][H][W][C2] or null: channel concat [in1, in2]
    int C1, C2;
    const float* w;     // packed [Cout][taps][C1+C2]
    const float* bias;  // [Cout] or null
    const float* bias2; // per-sample bias table, element (b, co) at bias2[b*bias2_stride + co], or null
    int bias2_stride;
    const float* res;   // residual [B][H][W][Cout] or null (may alias out)
    float* out;         // [B][H][W][Cout]
    int B, H, W, Cout;
    int taps;           // 9 (3x3, pad 1) or 1 (1x1)
    float scale;        // out = (acc + bias + bias2 + res) * scale
    // split-K (small images): K steps are divided over gridDim.y slices, each slice writes its raw partial
    // tile to `partial` [ksplit][B*H*W][Cout]; splitk_reduce sums the slices and applies the epilogue.
    int ksplit = 1;
    float* partial = nullptr;
    // Second partial set folded into the same reduction: the split-K slices of ANOTHER convolution
    // with the same output shape -- the 1x1 shortcut Conv_2(x) of a ResnetBlock, whose sum with Conv_1(h) is the block's
    // output (layerspp.py:268-274) -- are added slice by slice after this conv's own, plus that conv's bias `bias_x`.
    // Saves the shortcut's own reduction launch and the round trip of its output through HBM.
    const float* partial2 = nullptr;    // [ksplit2][B*H*W][Cout]
    int ksplit2 = 0;
    const float* bias_x = nullptr;      // [Cout] or null
    // fused GroupNorm statistics of the OUTPUT (optional): per-(sample, pixel tile, channel) sum / sum of squares
    // written to stats[((b * stats_nblk + tile) * Cout + c) * 2 + {0,1}], the layout gn_finalize consumes.
    // Only honoured when H*W % 128 == 0 and ksplit == 1 (see conv_fused_stats_blocks).
    float* stats = nullptr;
    int stats_nblk = 0;
    // fused GroupNorm(+SiLU) on the INPUT (optional; only the halo-tile 3x3 kernel applies it, see
    // conv_supports_fused_gn): A = act((x - mean[b][c]) * scale[b][c] + beta[c]) with c the concat channel index
    GnParams gn = {nullptr, nullptr, nullptr};
    int gn_silu = 0;
    // optional bf16 matrix-core path for the halo 3x3 kernel (Cout % 128 == 0): weights pre-split into bf16
    // planes, packed [Cout][9][Cin/32][planes][32]; terms = 3: x = hi + lo, products hi*hi + hi*lo + lo*hi
    // ("bf16x3", fp32-class accuracy, planes = 2); terms = 1: plain bf16 operands (planes = 1).  Accumulation,
    // GroupNorm, residuals and all activations in HBM stay fp32.
    const void* wq = nullptr;
    int terms = 0;
    int wq_f16 = 0;     // 1: the planes hold IEEE half instead of bf16 (terms must be 1; BASELINE config 5)
    const void* wfrag = nullptr;   // the same 3x3 weights in MFMA fragment order (launch_pc16_weights): conv3x3_pc16_kernel's B operand
    // The 1x1 shortcut of a ResnetBlock folded into its second 3x3 (conv3x3_pc16_kernel only): out = (conv3x3(act(GN(in)))
    // + bias + conv1x1(cat[sc1, sc2]; wfrag_sc) + bias_x) * scale -- layerspp.py:265-274 in ONE launch.  The shortcut's
    // SC1 + SC2 raw input channels (16-bit storage, same B / H / W, no GroupNorm) are extra K steps of every tile after its
    // nine-tap chunks; wfrag_sc = the 1x1 weights [Cout][1][SC1 + SC2] in fragment order (launch_pc16_weights, taps = 1).
    // Excludes `res`.
    const void* sc1 = nullptr;
    const void* sc2 = nullptr;
    int SC1 = 0, SC2 = 0;
    const void* wfrag_sc = nullptr;
    // optional F(4,3) Winograd weights of a 3x3 conv in MFMA fragment order (launch_f43_weights): when set and the
    // shape qualifies (conv_supports_wino) the fp32 3x3 runs the Winograd kernel (18 transformed taps per channel pair)
    const float* wino = nullptr;
    // optional F(4,3) x F(2,3) two-dimensional Winograd weights (launch_w2d_weights, 24 transformed taps per channel pair):
    // when set and the shape qualifies (conv_supports_w2d) the fp32 3x3 runs conv3x3_w2d_kernel -- its fused statistics
    // come in blocks of 64 pixels (4 x 16 strips: stats_nblk = H W / 64, set by the plan, Builder::conv)
    const float* wino2 = nullptr;
    // optional copy of `w` in MFMA fragment order (launch_smallm_weights): when set and the shape qualifies
    // (conv_smallm_ok) the fp32 conv runs conv_smallm_kernel -- K split inside the block, no slab, no second launch; its
    // fused statistics come in blocks of min(32, H W) pixels (conv_smallm_stats_blocks)
    const float* wsm = nullptr;
    const float* wsm16 = nullptr;       // ... and in the 16 x 16-tile fragment order (<= 512 pixels: conv_smallm16_kernel)
    // activation storage types (DT_*).  16-bit inputs are taken by the 16-bit matrix-core kernels (halo 3x3 with
    // Cout % 128 == 0, flat 1x1 / small 3x3: `wq` = the [Cout][taps][Cin] weights in the matching 16-bit type, terms = 1)
    // and by the 4-channel heads; 16-bit outputs by those plus the 4-channel input convs, the fp32 flat kernel
    // (attention output projection) and the split-K reductions.
    int in_dt = DT_F32, out_dt = DT_F32;
};
// K slices of the 16-bit flat kernel for a shape (its own policy: no Winograd alternative, two K steps per stage)
int conv16_ksplit(int B, int H, int W, int Cin, int Cout, int taps);
// true when a 3x3 conv with 16-bit operands runs the LDS-halo kernel (fused GroupNorm input possible, statistics of the
// output fused, H*W/128 partial blocks); otherwise the flat 16-bit kernel (+ split-K) takes it
bool conv16_uses_halo(int B, int H, int W, int C1, int C2, int Cout, int taps);
void pc16_set_channel_blocks(int mode);                                           // flowse_op_pc16_channel_blocks
bool conv16_uses_pc(int B, int H, int W, int C1, int C2, int Cout, int taps);     // ... and the producer / consumer form of it (needs ConvArgs::wfrag)
// number of per-sample partial-statistics blocks the 16-bit conv path writes for this shape (0 = none)
int conv16_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps);
// elementwise storage conversion (n elements, n % 4 == 0)
int launch_convert(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, hipStream_t s);
// F(4,3) Winograd weight transform along the kernel's vertical axis, packed [Cout][9][Cin] -> fragment order, on device
int launch_f43_weights(const float* w_packed, int Cout, int Cin, float* out, hipStream_t s);
// [Cout][9][Cin] 16-bit weights -> MFMA fragment order for conv3x3_pc16_kernel (same element count)
int launch_pc16_weights(const void* w16, int Cout, int Cin, void* dst, hipStream_t s, int taps = 9);
// 16-bit small-image kernel (conv16_smallm.hip): shapes it takes, its statistics geometry (blocks of min(32, H W) pixels)
bool conv16_smallm_ok(int B, int H, int W, int C1, int C2, int Cout, int taps);
int conv16_smallm_stats_blocks(int B, int H, int W);
inline int64_t conv_wino_numel(int Cout, int Cin) { return (int64_t)Cout * 18 * Cin; }
// F(4,3) x F(2,3) weight transform, packed [Cout][9][Cin] -> fragment order [Cout/32][h][Cin/32][6][4][64][4], on device
int launch_w2d_weights(const float* w_packed, int Cout, int Cin, float* out, hipStream_t s);
inline int64_t conv_w2d_numel(int Cout, int Cin) { return (int64_t)Cout * 24 * Cin; }
// conv_w2d_shape_ok: the two-dimensional Winograd kernel can run this shape (launch_conv takes it whenever
// ConvArgs::wino2 is set); conv_supports_w2d: ... and the policy wants it (enough blocks to fill the chip): the plan's test
bool conv_w2d_shape_ok(int B, int H, int W, int C1, int C2, int Cout, int taps);
bool conv_supports_w2d(int B, int H, int W, int C1, int C2, int Cout, int taps);
// small-M kernel (conv_smallm.hip): shapes it takes (<= 2048 pixels in the batch, channel counts multiples of 32; honours
// FLOWSE_NO_SMALLM=1), its weight copy and statistics geometry
bool conv_smallm_ok(int B, int H, int W, int C1, int C2, int Cout, int taps);
int launch_smallm_weights(const float* w_packed, int Cout, int taps, int Cin, float* out, hipStream_t s, bool tile16 = false);
bool conv_smallm_tile16(int B, int H, int W);
int conv_smallm_stats_blocks(int B, int H, int W);
// streaming fp32 1x1 kernel (conv_1x1.hip): shapes it takes, its statistics geometry (blocks of 256 pixels)
bool conv1x1_stream_ok(int B, int H, int W, int C1, int C2, int Cout, int taps);
int conv1x1_stream_stats_blocks(int B, int H, int W);
bool conv_w2d_enabled();                   // FLOWSE_W2D (read once): the model handle keeps the 2-D weights and uses the kernel
bool conv_supports_wino(int B, int H, int W, int C1, int C2, int Cout, int taps);
// whole-K F(4,3) launches of this shape use 128-channel blocks (conv3x3_f43_kernel<GN, false, 2>)
bool conv_f43_wide(int B, int H, int W, int Cout);
// host helper: split fp32 conv weights [Cout][Cin][3][3] into the packed bf16 planes described above
void pack_conv_bf16(const float* w, int Cout, int Cin, int terms, uint16_t* dst, bool f16 = false);
inline int64_t conv_bf16_numel(int Cout, int Cin, int terms) { return (int64_t)Cout * 9 * Cin * (terms == 1 ? 1 : 2); }
bool conv_supports_bf16(int B, int H, int W, int C1, int C2, int Cout, int taps);
// true when launch_conv will run the LDS-halo 3x3 kernel for this shape (the only one that can normalise its
// input on the fly)
bool conv_supports_fused_gn(int B, int H, int W, int C1, int C2, int Cout, int taps);
// true when the 4-output-channel 3x3 heads run the dedicated v_mfma_f32_4x4x1 kernel (fused GroupNorm input ok)
bool conv_supports_head4(int B, int H, int W, int C1, int C2, int Cout, int taps);
// number of per-sample partial blocks a conv writes when stats fusion applies to this shape, else 0
int conv_fused_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps);
// with_reduce = false: a split-K launch only writes the partial slices (the caller runs launch_splitk_reduce)
int launch_conv(const ConvArgs& a, hipStream_t s, bool with_reduce = true);
int launch_splitk_reduce(const ConvArgs& a, hipStream_t s);
// Reduction of a split-K conv fused with the GroupNorm that consumes its output (Conv_0 -> GroupNorm_1 of a ResnetBlock,
// layerspp.py:262-265; the group structure G = min(Cout / 4, 32) is known when the plan is built).  One block per (group,
// sample) sums the slices of its H*W x Cout/G elements, adds bias / per-sample bias, and -- holding the whole group --
// computes its exact mean / variance.  apply = 1: writes act(GroupNorm(.)) to a.out (the pre-norm tensor is never
// stored; gamma / beta / silu as given); apply = 0: writes the pre-norm tensor to a.out and the per-(sample, channel)
// mean / scale = rstd * gamma to gn_mean / gn_scale ([B][Cout]) for a consumer that normalises on load.
// Requires H*W * (Cout / G) <= 32768 elements per block and at least 128 blocks (conv_reduce_gn_ok).
bool conv_reduce_gn_ok(int B, int HW, int Cout);
int launch_splitk_reduce_gn(const ConvArgs& a, const float* gamma, const float* beta, float eps, int silu, int apply,
                            float* gn_mean, float* gn_scale, hipStream_t s);
// number of K slices launch_conv will use for this shape (1 = no split) and the partial-buffer size in floats
int conv_ksplit(int B, int H, int W, int Cin, int Cout, int taps);
// flat pixels per statistics block of a split-K conv's output
int conv_splitk_stats_group(int HW);
// true when this 3x3 shape runs as slices of the F(4,3) kernel
bool conv_splitk_is_wino(int B, int H, int W, int Cin, int Cout, int taps);
// true when the 4-channel input conv runs on the matrix cores (then it also emits fused GroupNorm statistics,
// H*W/128 partial blocks per sample)
bool conv_cin4_uses_mfma(int B, int H, int W, int Cout, int taps);
// statistics blocks per sample the Combine kernel (conv1x1 4 -> C + in-place residual) writes with its output (0: none)
int conv_cin4_stats_blocks(int B, int H, int W, int Cout);

// direct conv for 4 input channels (input layer, Combine): VALU, HBM-bound
int launch_conv_cin4(const ConvArgs& a, hipStream_t s);

// ------------------------------------------------------------------ GroupNorm
// stats over (C/G channels) x H x W for a (possibly concatenated) NHWC tensor
int gn_partial_blocks(int HW, int C);
int gn_pixels_per_block(int HW, int nblk);      // ceil(HW / nblk): the block size every producer uses
// dt / in_dt / out_dt: storage types of the activation tensors (DT_*); out_dt is DT_F32 or in_dt
int launch_gn_stats(const void* in1, int C1, const void* in2, int C2, int B, int HW,
                    float* partial /*[B][nblk][C][2]*/, int nblk, hipStream_t s, int dt = DT_F32);
// statistics may come as two partial sets (channel concat of two tensors whose partials were produced
// separately, e.g. by their conv epilogues): set 1 covers channels [0,C1), set 2 channels [C1, C1+C2)
int launch_gn_finalize(const float* partial1, int nblk1, int C1, const float* partial2, int nblk2, int C2, int B,
                       int HW, int G, const float* gamma, float eps, float* mean /*[B][C]*/,
                       float* scale /*[B][C]*/, hipStream_t s);

// finalize + apply in one launch (small images: HW * C / G elements per block)
int launch_gn_finalize_apply(const void* in1, const float* partial1, int nblk1, int C1, const void* in2,
                             const float* partial2, int nblk2, int C2, int B, int HW, int G, const float* gamma,
                             const float* beta, float eps, int silu, void* out, hipStream_t s, int in_dt = DT_F32,
                             int out_dt = DT_F32);
int launch_gn_apply(const void* in1, int C1, const void* in2, int C2, int B, int HW,
                    GnParams gn, int silu, void* out, hipStream_t s, int in_dt = DT_F32, int out_dt = DT_F32);

// ------------------------------------------------------------------ FIR resampling ([1,3,3,1] x [1,3,3,1] / 64)
// down: out[B][H/2][W/2][C]; up: out[B][2H][2W][C] (gain 4).  Optional fused GN(+SiLU) on the input
// (gn.mean == null -> raw), optional elementwise `add` tensor (same shape as out) summed into the result.
// out2 (optional): the same resampling of the raw (un-normalised) input, written from the same read.
// dt: storage type of in / out / out2 / add (DT_*).
int launch_fir_down(const void* in, int B, int H, int W, int C, GnParams gn, int silu, void* out,
                    hipStream_t s, void* out2 = nullptr, int dt = DT_F32);
int launch_fir_up(const void* in, int B, int H, int W, int C, GnParams gn, int silu, const void* add,
                  void* out, hipStream_t s, void* out2 = nullptr, int dt = DT_F32);
// generic NCHW upfirdn2d (drop-in for the reference's op/upfirdn2d ABI), kernel up to 8x8
int launch_upfirdn2d_nchw(const float* in, const float* kernel, int planes, int in_h, int in_w, int kh,
                          int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                          int pad_y0, int pad_y1, float* out, int out_h, int out_w, hipStream_t s);

// ------------------------------------------------------------------ attention
// qkv: [B][L][3C] (q | k | v per token), out: [B][L][C]; softmax(q k^T * C^-1/2) v, single head.  dt = storage type of qkv AND
// out: fp32 (v_mfma_f32_32x32x2_f32) or bf16 / half (v_mfma_f32_32x32x16_*, fp32 softmax state and accumulation)
int launch_attention(const void* qkv, int B, int L, int C, void* out, hipStream_t s, int dt = DT_F32);

// ------------------------------------------------------------------ small ops
// Per-call arguments of the network's boundary kernels (feature pack, time embedding, head).  They live in DEVICE
// memory (one block per model handle, rewritten by a one-thread kernel whose argument is the new value) so that the
// launch list of a forward pass contains no per-call pointer or scalar: it can be captured once as a hipGraph and
// replayed for every solver step / every call.
struct CallBlock {
    const float* x;     // complex64 [B,1,F,T]
    const float* y;
    const float* t;     // float32 [B]
    float* out;
    int mode;           // see launch_head
    float dt;
    // mode 3 (one stage of a fixed-step Runge-Kutta step, flowse_rk_sample): with v = the network output,
    //   out     = x0     + a * v   (the next stage's input; skipped when out == null)
    //   acc_out = acc_in + b * v   (the running combination of slopes; skipped when acc_out == null)
    const float* x0 = nullptr;
    const float* acc_in = nullptr;
    float* acc_out = nullptr;
    float a = 0.f, b = 0.f;
};
int launch_set_call(CallBlock* d_cb, const CallBlock& value, hipStream_t s);
// d_ts[i * B + b] = ts[i] for i < N (ts: host values, passed to the kernel by value in chunks)
int launch_fill_times(float* d_ts, const float* ts, int N, int B, hipStream_t s);
// `cb` != null: x / y / t / out / mode / dt are read from the device-resident call block instead of the arguments
int launch_pack_input(const float* x_c64, const float* y_c64, int B, int F, int T, float* out4, hipStream_t s,
                      const CallBlock* cb = nullptr);
// GaussianFourierProjection(log t): out[b][0:E] = sin, out[b][E:2E] = cos
int launch_gfp(const float* t, const float* Wf, int B, int E, float* out, hipStream_t s, const CallBlock* cb = nullptr);
// out[b][r] = act(sum_k W[r][k] in[b][k] + bias[r]); act: 0 none, 1 SiLU
int launch_linear(const float* in, int B, int K, const float* W, const float* bias, int R, int act,
                  float* out, int out_stride, hipStream_t s);
// head: v = Wout (pyr / t[b]) + bout  (1x1 conv 4 -> 2, complex pack)
//   mode 0: out = v            (NCSNpp.forward)
//   mode 1: out = -v           (VFModel.forward)
//   mode 2: out = x + dt * v   (Euler update  x + VF * (-dt), VF = -v); out may alias x
//   mode 3: Runge-Kutta stage (call block only): out = x0 + a * v and / or acc_out = acc_in + b * v
int launch_head(const float* pyr4, const float* t, const float* Wout /*[2][4]*/, const float* bout /*[2]*/,
                int B, int F, int T, int mode, const float* x_c64, float dt, float* out_c64, hipStream_t s,
                const CallBlock* cb = nullptr);
// fused STFT + compression (-> complex64 [B,1,256,Tpad], frames >= T zeroed) and decompression + iSTFT
int launch_stft_compress(const float* sig, int B, int L, float scale_in, float* out_c64, int T, int Tpad, float factor,
                         float exponent, hipStream_t s);
int launch_istft_decompress(const float* spec_c64, int B, int T, int Tpad, float factor, float exponent, float* out,
                            int Lout, float scale_out, hipStream_t s);
// out = y + sigma * z   (complex64 as float pairs)
int launch_axpy(const float* y, const float* z, float sigma, int64_t n, float* out, hipStream_t s);

}  // namespace flowse
