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
// Depth-wise and identical for every channel: NHWC, channel quads per thread, 128-byte runs per pixel.  Optional fusions: GroupNorm(+SiLU) applied to the input samples on
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

// Both kernels work on LDS tiles: a block stages the input window of its output tile ONCE -- with the fused
// GroupNorm(+SiLU) applied on the way in, so every input element is normalised once per block instead of once per tap
// (4x per element for the down filter, 16x for the up filter in a thread-per-output formulation) -- and every output
// is then a handful of conflict-free ds_read_b128.  A block covers CC = 32 channels (8 quads): thread = (pixel lane,
// channel quad), global accesses are 128-byte runs per pixel.  Samples outside the image are zero (also after the
// fused activation).  ST = storage type of `in` / `out` / `out2` / `add` (float, bf16, half; arithmetic is fp32).
// out2 (optional): the same resampling of the RAW input (the ResnetBlock's shortcut branch resamples x while the main
// branch resamples act(GroupNorm(x)), layerspp.py:251-259): both from one read of the input.
constexpr int FIR_CC = 32;                 // channels per block

// Stage the input window of a block into LDS (fused GroupNorm(+SiLU) on the way in; `raw` also takes the untransformed
// values).  ALL of a thread's pixels are requested before the first one is used: the plain loop (load -> transform -> store
// per pixel, five or six dependent round trips per thread) made both kernels latency-bound -- the 16-bit modes, with half
// the bytes, were no faster than fp32 (fir_down at 256 x 256: 91 us vs 99 us).
template <class ST, int NPX, int IX>
__device__ __forceinline__ void fir_stage(float (*tile)[FIR_CC], float (*raw)[FIR_CC], const ST* base, const GnQuad& g, int pl,
                                          int q, int iy0, int ix0, int H, int W, int C, bool cok) {
    constexpr int NIT = (NPX + 31) / 32;
    float4 r[NIT];
    bool in[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int p = pl + 32 * k;
        const int py = p / IX, px = p - py * IX;
        const int y = iy0 + py, x = ix0 + px;
        in[k] = cok && p < NPX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        r[k] = St<ST>::ld4(base + (in[k] ? ((int64_t)y * W + x) * C : 0));          // (clamped address: unconditional load)
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int p = pl + 32 * k;
        if (p >= NPX) break;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 rv = in[k] ? r[k] : z;
        const float4 v = in[k] ? apply_tx(rv, g) : z;
        *reinterpret_cast<float4*>(&tile[p][q * 4]) = v;
        if (raw) *reinterpret_cast<float4*>(&raw[p][q * 4]) = rv;
    }
}

// down: output tile 4 x 8, input window 10 x 18.  grid (tiles_x * tiles_y, ceil(C / 32), B)
template <class ST>
__global__ __launch_bounds__(256) void fir_down_kernel(const ST* __restrict__ in, int H, int W, int C, GnParams gn,
                                                       int silu, ST* __restrict__ out, ST* __restrict__ out2,
                                                       int tiles_x) {
    constexpr int TY = 4, TX = 8, IY = 2 * TY + 2, IX = 2 * TX + 2, NPX = IY * IX;
    __shared__ __attribute__((aligned(16))) float tile[2][NPX][FIR_CC];      // [0] transformed, [1] raw (out2 only)
    const int OH = H >> 1, OW = W >> 1;
    const int t = threadIdx.x, q = t & 7, pl = t >> 3;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.z, c = blockIdx.y * FIR_CC + q * 4;
    const bool cok = c < C;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const GnQuad g = cok ? gn_quad(gn, silu, b, C, c) : GnQuad{};
    const ST* base = in + (int64_t)b * H * W * C + (cok ? c : 0);
    fir_stage<ST, NPX, IX>(tile[0], out2 ? tile[1] : nullptr, base, g, pl, q, iy0, ix0, H, W, C, cok);
    __syncthreads();
    const int oyl = pl >> 3, oxl = pl & 7;                                   // 32 pixel lanes = the 4 x 8 outputs
    const int oy = oy0 + oyl, ox = ox0 + oxl;
    if (!cok || oy >= OH || ox >= OW) return;
    const float k1[4] = {1.f, 3.f, 3.f, 1.f};
    const float s = 1.f / 64.f;
    const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + c;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && !out2) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) {
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int fx = 0; fx < 4; ++fx) {
                const float4 v = *reinterpret_cast<const float4*>(&tile[which][(2 * oyl + fy) * IX + 2 * oxl + fx][q * 4]);
                row.x = fmaf(k1[fx], v.x, row.x); row.y = fmaf(k1[fx], v.y, row.y);
                row.z = fmaf(k1[fx], v.z, row.z); row.w = fmaf(k1[fx], v.w, row.w);
            }
            acc.x = fmaf(k1[fy], row.x, acc.x); acc.y = fmaf(k1[fy], row.y, acc.y);
            acc.z = fmaf(k1[fy], row.z, acc.z); acc.w = fmaf(k1[fy], row.w, acc.w);
        }
        St<ST>::st4((which ? out2 : out) + off, make_float4(acc.x * s, acc.y * s, acc.z * s, acc.w * s));
    }
}

// up: input tile 8 x 8 (+1 halo) -> output tile 16 x 16.  grid (tiles_x * tiles_y, ceil(C / 32), B)
template <class ST>
__global__ __launch_bounds__(256) void fir_up_kernel(const ST* __restrict__ in, int H, int W, int C, GnParams gn,
                                                     int silu, const ST* __restrict__ add, ST* __restrict__ out,
                                                     ST* __restrict__ out2, int tiles_x) {
    constexpr int TY = 8, TX = 8, IY = TY + 2, IX = TX + 2, NPX = IY * IX;
    __shared__ __attribute__((aligned(16))) float tile[2][NPX][FIR_CC];
    const int OH = 2 * H, OW = 2 * W;
    const int t = threadIdx.x, q = t & 7, pl = t >> 3;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.z, c = blockIdx.y * FIR_CC + q * 4;
    const bool cok = c < C;
    const int iy0 = ty * TY - 1, ix0 = tx * TX - 1;
    const GnQuad g = cok ? gn_quad(gn, silu, b, C, c) : GnQuad{};
    const ST* base = in + (int64_t)b * H * W * C + (cok ? c : 0);
    fir_stage<ST, NPX, IX>(tile[0], out2 ? tile[1] : nullptr, base, g, pl, q, iy0, ix0, H, W, C, cok);
    __syncthreads();
    if (!cok) return;
    // per axis: two taps (position, weight); even o = 2a: (a-1, 1), (a, 3); odd o = 2a+1: (a, 3), (a+1, 1)
    const float sc = 1.f / 16.f;
    const int oxl = pl & 15;                                                 // 16 columns x 2 rows per pass
    const int ox = 2 * tx * TX + oxl;
    if (ox >= OW) return;
    const int axl = oxl >> 1;
    const int lx0 = (oxl & 1) ? axl + 1 : axl;                               // local column of the first tap (+1 halo)
    const float wx0 = (oxl & 1) ? 3.f : 1.f, wx1 = (oxl & 1) ? 1.f : 3.f;
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
        const int oyl = pass * 2 + (pl >> 4);
        const int oy = 2 * ty * TY + oyl;
        if (oy >= OH) break;
        const int ayl = oyl >> 1;
        const int ly0 = (oyl & 1) ? ayl + 1 : ayl;
        const float wy0 = (oyl & 1) ? 3.f : 1.f, wy1 = (oyl & 1) ? 1.f : 3.f;
        const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + c;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            if (which == 1 && !out2) break;
            const float4 v00 = *reinterpret_cast<const float4*>(&tile[which][ly0 * IX + lx0][q * 4]);
            const float4 v01 = *reinterpret_cast<const float4*>(&tile[which][ly0 * IX + lx0 + 1][q * 4]);
            const float4 v10 = *reinterpret_cast<const float4*>(&tile[which][(ly0 + 1) * IX + lx0][q * 4]);
            const float4 v11 = *reinterpret_cast<const float4*>(&tile[which][(ly0 + 1) * IX + lx0 + 1][q * 4]);
            float4 r0, r1, o;
            r0.x = fmaf(wx1, v01.x, fmaf(wx0, v00.x, 0.f)); r0.y = fmaf(wx1, v01.y, fmaf(wx0, v00.y, 0.f));
            r0.z = fmaf(wx1, v01.z, fmaf(wx0, v00.z, 0.f)); r0.w = fmaf(wx1, v01.w, fmaf(wx0, v00.w, 0.f));
            r1.x = fmaf(wx1, v11.x, fmaf(wx0, v10.x, 0.f)); r1.y = fmaf(wx1, v11.y, fmaf(wx0, v10.y, 0.f));
            r1.z = fmaf(wx1, v11.z, fmaf(wx0, v10.z, 0.f)); r1.w = fmaf(wx1, v11.w, fmaf(wx0, v10.w, 0.f));
            o.x = fmaf(wy1, r1.x, fmaf(wy0, r0.x, 0.f)) * sc; o.y = fmaf(wy1, r1.y, fmaf(wy0, r0.y, 0.f)) * sc;
            o.z = fmaf(wy1, r1.z, fmaf(wy0, r0.z, 0.f)) * sc; o.w = fmaf(wy1, r1.w, fmaf(wy0, r0.w, 0.f)) * sc;
            if (which == 0 && add) {
                const float4 r = St<ST>::ld4(add + off);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            St<ST>::st4((which ? out2 : out) + off, o);
        }
    }
}

static int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int launch_fir_down(const void* in, int B, int H, int W, int C, GnParams gn, int silu, void* out, hipStream_t s,
                    void* out2, int dt) {
    const int tiles_x = (W / 2 + 7) / 8, tiles_y = (H / 2 + 3) / 4, cg = (C + FIR_CC - 1) / FIR_CC;
    if ((C & 3) || (H & 1) || (W & 1) || cg > 65535 || B > 65535) {
        set_error("fir_down: unsupported shape B=%d H=%d W=%d C=%d", B, H, W, C);
        return ERR_SHAPE;
    }
    FLOWSE_DT_SWITCH(dt, ST, hipLaunchKernelGGL(fir_down_kernel<ST>, dim3(tiles_x * tiles_y, cg, B), dim3(256), 0, s,
                                                static_cast<const ST*>(in), H, W, C, gn, silu, static_cast<ST*>(out),
                                                static_cast<ST*>(out2), tiles_x));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_fir_up(const void* in, int B, int H, int W, int C, GnParams gn, int silu, const void* add, void* out,
                  hipStream_t s, void* out2, int dt) {
    const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8, cg = (C + FIR_CC - 1) / FIR_CC;
    if ((C & 3) || cg > 65535 || B > 65535) {
        set_error("fir_up: unsupported shape B=%d H=%d C=%d", B, H, C);
        return ERR_SHAPE;
    }
    FLOWSE_DT_SWITCH(dt, ST, hipLaunchKernelGGL(fir_up_kernel<ST>, dim3(tiles_x * tiles_y, cg, B), dim3(256), 0, s,
                                                static_cast<const ST*>(in), H, W, C, gn, silu,
                                                static_cast<const ST*>(add), static_cast<ST*>(out),
                                                static_cast<ST*>(out2), tiles_x));
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
