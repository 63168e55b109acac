// FIR resampling by 2 with the separable kernel outer([1,3,3,1]) / 64 on NHWC fp32 tensors.
//
// Replaces (reference): upsample_2d / downsample_2d (flowmse/backbones/ncsnpp_utils/up_or_down_sampling.py:195-257)
// -> upfirdn2d (op/upfirdn2d.py:145-200) -> the CUDA kernels of op/upfirdn2d_kernel.cu:107-207 (modes 3 and 5).
//
//   down (pad (1,1), stride 2):      out[oy][ox] = sum_{i,j<4} k[i] k[j] / 64 * in[2oy-1+i][2ox-1+j]
//   up   (zero-insert x2, gain 4, pad (2,1)):
//        even o = 2a:   (1 * in[a-1] + 3 * in[a]) / 4        odd o = 2a+1:  (3 * in[a] + 1 * in[a+1]) / 4
//        per axis (the polyphase form of the same 4-tap filter); samples outside the image are zero.
//
// Depth-wise and identical for every channel, so in NHWC each thread owns one output pixel x one channel quad
// and every access is a coalesced float4.  Optional fusions: GroupNorm(+SiLU) applied to the input samples on
// load (ResnetBlockBigGANpp resamples act(GroupNorm_0(x)), layerspp.py:246-259) and an elementwise `add`
// (output pyramid: pyramid = upsample(pyramid) + pyramid_h, ncsnpp.py:354-359).
#include "common.h"

namespace flowse {

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }

struct GnQuad {
    float4 mu, sc, be;
    int on, silu;
};

__device__ __forceinline__ float4 apply_tx(float4 v, const GnQuad& g) {
    if (g.on) {
        v.x = fmaf(v.x - g.mu.x, g.sc.x, g.be.x);
        v.y = fmaf(v.y - g.mu.y, g.sc.y, g.be.y);
        v.z = fmaf(v.z - g.mu.z, g.sc.z, g.be.z);
        v.w = fmaf(v.w - g.mu.w, g.sc.w, g.be.w);
        if (g.silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    }
    return v;
}

__device__ __forceinline__ GnQuad gn_quad(const GnParams& gn, int silu, int b, int C, int c) {
    GnQuad g;
    g.on = gn.mean != nullptr;
    g.silu = silu;
    if (g.on) {
        g.mu = *reinterpret_cast<const float4*>(gn.mean + (int64_t)b * C + c);
        g.sc = *reinterpret_cast<const float4*>(gn.scale + (int64_t)b * C + c);
        g.be = *reinterpret_cast<const float4*>(gn.beta + c);
    }
    return g;
}

// grid: (ceil(OW * C/4 / 256), OH, B); one thread = one output pixel x one channel quad (32-bit index math only)
// out2 (optional): the same resampling of the RAW input (the ResnetBlock's shortcut branch resamples x while the
// main branch resamples act(GroupNorm(x)), layerspp.py:251-259): both from one read of the input
__global__ __launch_bounds__(256) void fir_down_kernel(const float* __restrict__ in, int H, int W, int C, GnParams gn,
                                                       int silu, float* __restrict__ out, float* __restrict__ out2) {
    const unsigned Q = C >> 2, OW = W >> 1, OH = H >> 1;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= OW * Q) return;
    const unsigned ox = idx / Q, cq = idx - ox * Q;
    const int oy = blockIdx.y, b = blockIdx.z;
    const int c = cq * 4;
    const float k1[4] = {1.f, 3.f, 3.f, 1.f};
    const GnQuad g = gn_quad(gn, silu, b, C, c);
    const float* base = in + (int64_t)b * H * W * C + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        const int y = 2 * oy - 1 + ty;
        if ((unsigned)y >= (unsigned)H) continue;
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f), row2 = row;
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) {
            const int x = 2 * (int)ox - 1 + tx;
            if ((unsigned)x >= (unsigned)W) continue;
            const float4 r = *reinterpret_cast<const float4*>(base + ((int64_t)y * W + x) * C);
            const float4 v = apply_tx(r, g);
            row.x = fmaf(k1[tx], v.x, row.x); row.y = fmaf(k1[tx], v.y, row.y);
            row.z = fmaf(k1[tx], v.z, row.z); row.w = fmaf(k1[tx], v.w, row.w);
            if (out2) {
                row2.x = fmaf(k1[tx], r.x, row2.x); row2.y = fmaf(k1[tx], r.y, row2.y);
                row2.z = fmaf(k1[tx], r.z, row2.z); row2.w = fmaf(k1[tx], r.w, row2.w);
            }
        }
        acc.x = fmaf(k1[ty], row.x, acc.x); acc.y = fmaf(k1[ty], row.y, acc.y);
        acc.z = fmaf(k1[ty], row.z, acc.z); acc.w = fmaf(k1[ty], row.w, acc.w);
        if (out2) {
            acc2.x = fmaf(k1[ty], row2.x, acc2.x); acc2.y = fmaf(k1[ty], row2.y, acc2.y);
            acc2.z = fmaf(k1[ty], row2.z, acc2.z); acc2.w = fmaf(k1[ty], row2.w, acc2.w);
        }
    }
    const float s = 1.f / 64.f;
    const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + c;
    *reinterpret_cast<float4*>(out + off) = make_float4(acc.x * s, acc.y * s, acc.z * s, acc.w * s);
    if (out2) *reinterpret_cast<float4*>(out2 + off) = make_float4(acc2.x * s, acc2.y * s, acc2.z * s, acc2.w * s);
}

// grid: (ceil(2W * C/4 / 256), 2H, B)
__global__ __launch_bounds__(256) void fir_up_kernel(const float* __restrict__ in, int H, int W, int C, GnParams gn,
                                                     int silu, const float* __restrict__ add, float* __restrict__ out,
                                                     float* __restrict__ out2) {
    const unsigned Q = C >> 2, OW = W * 2, OH = H * 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= OW * Q) return;
    const unsigned ox = idx / Q, cq = idx - ox * Q;
    const int oy = blockIdx.y, b = blockIdx.z;
    const int c = cq * 4;
    const GnQuad g = gn_quad(gn, silu, b, C, c);
    const float* base = in + (int64_t)b * H * W * C + c;
    // per axis: two taps (position, weight); even: (a-1, 1), (a, 3); odd: (a, 3), (a+1, 1)
    const int ay = oy >> 1, ax = ox >> 1;
    const int y0 = (oy & 1) ? ay : ay - 1, x0 = (ox & 1) ? ax : ax - 1;
    const float wy0 = (oy & 1) ? 3.f : 1.f, wy1 = (oy & 1) ? 1.f : 3.f;
    const float wx0 = (ox & 1) ? 3.f : 1.f, wx1 = (ox & 1) ? 1.f : 3.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
        const int y = y0 + ty;
        if ((unsigned)y >= (unsigned)H) continue;
        const float wy = ty ? wy1 : wy0;
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f), row2 = row;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
            const int x = x0 + tx;
            if ((unsigned)x >= (unsigned)W) continue;
            const float wx = tx ? wx1 : wx0;
            const float4 r = *reinterpret_cast<const float4*>(base + ((int64_t)y * W + x) * C);
            const float4 v = apply_tx(r, g);
            row.x = fmaf(wx, v.x, row.x); row.y = fmaf(wx, v.y, row.y);
            row.z = fmaf(wx, v.z, row.z); row.w = fmaf(wx, v.w, row.w);
            if (out2) {
                row2.x = fmaf(wx, r.x, row2.x); row2.y = fmaf(wx, r.y, row2.y);
                row2.z = fmaf(wx, r.z, row2.z); row2.w = fmaf(wx, r.w, row2.w);
            }
        }
        acc.x = fmaf(wy, row.x, acc.x); acc.y = fmaf(wy, row.y, acc.y);
        acc.z = fmaf(wy, row.z, acc.z); acc.w = fmaf(wy, row.w, acc.w);
        if (out2) {
            acc2.x = fmaf(wy, row2.x, acc2.x); acc2.y = fmaf(wy, row2.y, acc2.y);
            acc2.z = fmaf(wy, row2.z, acc2.z); acc2.w = fmaf(wy, row2.w, acc2.w);
        }
    }
    const float s = 1.f / 16.f;
    float4 o = make_float4(acc.x * s, acc.y * s, acc.z * s, acc.w * s);
    const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + c;
    if (add) {
        const float4 r = *reinterpret_cast<const float4*>(add + off);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(out + off) = o;
    if (out2) *reinterpret_cast<float4*>(out2 + off) = make_float4(acc2.x * s, acc2.y * s, acc2.z * s, acc2.w * s);
}

static int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int launch_fir_down(const float* in, int B, int H, int W, int C, GnParams gn, int silu, float* out, hipStream_t s,
                    float* out2) {
    if ((C & 3) || (H & 1) || (W & 1) || H / 2 > 65535 || B > 65535) {
        set_error("fir_down: unsupported shape B=%d H=%d W=%d C=%d", B, H, W, C);
        return ERR_SHAPE;
    }
    const unsigned per_row = (unsigned)(W / 2) * (C / 4);
    hipLaunchKernelGGL(fir_down_kernel, dim3((per_row + 255) / 256, H / 2, B), dim3(256), 0, s, in, H, W, C, gn, silu,
                       out, out2);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_fir_up(const float* in, int B, int H, int W, int C, GnParams gn, int silu, const float* add, float* out,
                  hipStream_t s, float* out2) {
    if ((C & 3) || H * 2 > 65535 || B > 65535) {
        set_error("fir_up: unsupported shape B=%d H=%d C=%d", B, H, C);
        return ERR_SHAPE;
    }
    const unsigned per_row = (unsigned)(W * 2) * (C / 4);
    hipLaunchKernelGGL(fir_up_kernel, dim3((per_row + 255) / 256, H * 2, B), dim3(256), 0, s, in, H, W, C, gn, silu, add,
                       out, out2);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Generic NCHW upfirdn2d: the drop-in for the reference's only native ABI,
//   upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
// (op/upfirdn2d.cpp:12-22; input viewed [N*C, H, W, 1], op/upfirdn2d.py:99).  Semantics follow
// upfirdn2d_native (op/upfirdn2d.py:159-200): zero-insert, pad (negative pad crops), correlate with the
// FLIPPED kernel, decimate.  One thread per output sample; HBM-bound.
__global__ __launch_bounds__(256) void upfirdn2d_nchw_kernel(const float* __restrict__ in,
                                                             const float* __restrict__ kernel, int planes, int in_h,
                                                             int in_w, int kh, int kw, int up_x, int up_y,
                                                             int down_x, int down_y, int pad_x0, int pad_y0,
                                                             float* __restrict__ out, int out_h, int out_w,
                                                             int64_t total) {
    __shared__ float ks[64];
    if (threadIdx.x < kh * kw) ks[threadIdx.x] = kernel[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % out_w);
        int64_t r = i / out_w;
        const int oy = (int)(r % out_h);
        const int p = (int)(r / out_h);
        const float* src = in + (int64_t)p * in_h * in_w;
        float acc = 0.f;
        // out[oy][ox] = sum_{i,j} kflip[i][j] * xup_pad[oy*down + i][ox*down + j]
        for (int ky = 0; ky < kh; ++ky) {
            const int uy = oy * down_y + ky - pad_y0;      // coordinate in the zero-inserted image
            if (uy < 0 || uy % up_y != 0) continue;
            const int y = uy / up_y;
            if (y >= in_h) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int ux = ox * down_x + kx - pad_x0;
                if (ux < 0 || ux % up_x != 0) continue;
                const int x = ux / up_x;
                if (x >= in_w) continue;
                acc = fmaf(ks[(kh - 1 - ky) * kw + (kw - 1 - kx)], src[(int64_t)y * in_w + x], acc);
            }
        }
        out[i] = acc;
    }
}

int launch_upfirdn2d_nchw(const float* in, const float* kernel, int planes, int in_h, int in_w, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                          float* out, int out_h, int out_w, hipStream_t s) {
    if (kh * kw > 64 || kh < 1 || kw < 1 || up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) {
        set_error("upfirdn2d: unsupported kernel %dx%d / factors", kh, kw);
        return ERR_SHAPE;
    }
    const int eh = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    const int ew = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (eh != out_h || ew != out_w) {
        set_error("upfirdn2d: output shape %dx%d, expected %dx%d", out_h, out_w, eh, ew);
        return ERR_SHAPE;
    }
    const int64_t total = (int64_t)planes * out_h * out_w;
    hipLaunchKernelGGL(upfirdn2d_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, kernel, planes, in_h, in_w,
                       kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out, out_h, out_w, total);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
