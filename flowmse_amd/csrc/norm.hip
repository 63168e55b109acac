// GroupNorm (+SiLU) for NHWC fp32 tensors, optionally over the channel concat of two tensors.
//
// Replaces (reference): nn.GroupNorm(num_groups=min(C//4,32), C, eps=1e-6) followed by SiLU as used in
// ResnetBlockBigGANpp (flowmse/backbones/ncsnpp_utils/layerspp.py:219,231,246,265), AttnBlockpp (:67,77) and the
// progressive-output heads (flowmse/backbones/ncsnpp.py:210,222,347,356), incl. the GroupNorm that spans the
// concat [h, skip] of ncsnpp.py:337 (a group may straddle the two source tensors, e.g. 384 = 256 + 128).
//
// Three kernels:
//   gn_stats     per-(sample, block, channel) partial sum / sum of squares (fp32, <= ~1k terms per partial);
//                threads own a channel quad (float4) and stride over pixels -> coalesced 16 B/lane streams,
//                LDS tree over the pixel lanes.  Deterministic (no atomics).
//   gn_finalize  per-(sample, group) reduction of the partials in fp64 -> mean, rstd; writes per-(b,c)
//                mean and scale = rstd * gamma (biased variance, as torch).
//   gn_apply     y = (x - mean) * scale + beta, optional SiLU; writes the (concatenated) tensor.
// The conv / FIR kernels can also consume GnParams directly (fused apply).
#include "common.h"

namespace flowse {

constexpr int GN_THREADS = 256;

int gn_pixels_per_block(int HW, int nblk) { return (HW + nblk - 1) / nblk; }

int gn_partial_blocks(int HW, int C) {
    // <= 256 pixels per block (enough blocks in flight to stream at HBM rate), at most 512 per sample
    int nblk = (HW + 255) / 256;
    if (nblk < 1) nblk = 1;
    if (nblk > 512) nblk = 512;
    (void)C;
    return nblk;
}

// grid (nblk, B).  Q = C/4 channel quads; PR = 256 / Q pixel lanes (Q <= 256).  Block blk covers pixels
// [blk * per, min(HW, (blk + 1) * per)) and writes (mean, M2) per channel (see Stat4 in common.h).
template <class ST>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const ST* __restrict__ in1, int C1,
                                                              const ST* __restrict__ in2, int C2, int HW,
                                                              float* __restrict__ partial, int nblk, int per) {
    __shared__ float red[GN_THREADS * 8];
    const int C = C1 + C2, Q = C >> 2;
    const int PR = GN_THREADS / Q;
    const int tid = threadIdx.x;
    const int pr = tid / Q, cq = tid - pr * Q;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int p0 = blk * per, p1 = min(HW, p0 + per);
    Stat4 st;
    st.init();
    if (pr < PR) {
        const int c = cq * 4;
        const ST* src;
        int cs, cc;
        if (c < C1) { src = in1; cs = C1; cc = c; } else { src = in2; cs = C2; cc = c - C1; }
        src += (int64_t)b * HW * cs + cc;
#pragma unroll 4
        for (int p = p0 + pr; p < p1; p += PR) st.add(St<ST>::ld4(src + (int64_t)p * cs));
    }
    st.finish(red + tid * 8);
    __syncthreads();
    if (pr == 0) {
        float acc8[8], nacc = (float)st.n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = red[tid * 8 + j];
        const int len = p1 - p0;
        for (int r = 1; r < PR; ++r) {
            const int cnt = r < len ? (len - r + PR - 1) / PR : 0;
            chan_merge4(nacc, acc8, (float)cnt, red + (r * Q + cq) * 8);
        }
        float* dst = partial + (((int64_t)b * nblk + blk) * C + cq * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[2 * j] = acc8[j]; dst[2 * j + 1] = acc8[4 + j]; }
    }
}

// (mean, 1/sqrt(var + eps)) of group g of sample b from the per-(block, channel) partials; all 256 threads of the block
// take part and receive the result.  Channels [0,C1) take their partials from set 1 (ppb1 pixels per block), [C1,C1+C2)
// from set 2; a group may straddle the two sets (384 = 256 + 128: group 21).
//
// ONE pass over the partials, in fp64:  S1 = sum n_i mean_i,  S2 = sum (M2_i + n_i mean_i^2)  ->  mean = S1 / N,
// var = S2 / N - mean^2 (N = HW * cpg).  The products n_i mean_i^2 of fp32 values are exact in fp64 and the
// subtraction loses log2(mean^2 / var) of 53 bits -- at |mean| = 1e4 sigma still 1e-8 relative, far below the fp32
// result's own rounding (the fp32 partials themselves are pivoted (mean, M2) pairs, so nothing cancels before this point).
// Access pattern: the (mean, M2) pairs of a group's channels are CONTIGUOUS per block (nch x 8 bytes), consecutive
// threads take consecutive pairs and then the next block -- every 128-byte line that is touched is used by nch lanes at
// once (round 2 read one pair per lane at a stride of C x 8 bytes, twice: 64 lines per wave instruction; its launches
// reached 680 us at [32,1,256,1024] where a sample has 2048 tile partials).
__device__ __forceinline__ void gn_group_stats(const float* __restrict__ p1, int nblk1, int ppb1, int C1,
                                               const float* __restrict__ p2, int nblk2, int ppb2, int C2, int HW,
                                               int G, float eps, int g, int b, float& mean_out, float& rstd_out) {
    __shared__ double wsum[8];
    const int tid = threadIdx.x;
    const int C = C1 + C2;
    const int cpg = C / G;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const float* p = set ? p2 : p1;
        const int nblk = set ? nblk2 : nblk1, ppb = set ? ppb2 : ppb1, Cs = set ? C2 : C1;
        const int base = set ? C1 : 0;
        const int lo = max(c_lo, base) - base, hi = min(c_hi, base + Cs) - base;      // this set's channels of the group
        const int nch = hi - lo;
        if (nch <= 0 || !p) continue;
        const float2* q = reinterpret_cast<const float2*>(p) + (int64_t)b * nblk * Cs + lo;
        const int items = nblk * nch;
        // four independent requests per round (a thread's items are summed in the same order as one at a time: the loop
        // was a chain of ~1 us dependent latencies -- 16 rounds for the 1024 strip partials of a 256 x 256 sample); a remainder
        // of two or three items per thread is ONE predicated round on clamped addresses (round 5 took them one per round: two
        // extra round trips at 64 x 64), a single item stays a single request (eight predicated requests for everything, tried
        // in round 6, cost the single-utterance shape 1.2 %: most of its launches hold one item per thread)
        int it = tid;
        for (; it + 768 < items; it += 1024) {
            int blk[4];
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = it + 256 * u;
                blk[u] = k / nch;
                v[u] = q[(int64_t)blk[u] * Cs + (k - blk[u] * nch)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double n = (double)min(ppb, HW - blk[u] * ppb), m = (double)v[u].x;
                s1 += n * m;
                s2 += (double)v[u].y + n * m * m;
            }
        }
        if (it + 256 < items) {                              // two or three left
            int blk[3];
            float2 v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int k = it + 256 * u, kc = k < items ? k : items - 1;
                blk[u] = kc / nch;
                v[u] = q[(int64_t)blk[u] * Cs + (kc - blk[u] * nch)];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (it + 256 * u < items) {
                    const double n = (double)min(ppb, HW - blk[u] * ppb), m = (double)v[u].x;
                    s1 += n * m;
                    s2 += (double)v[u].y + n * m * m;
                }
            }
        } else if (it < items) {
            const int blk = it / nch, j = it - blk * nch;
            const float2 v = q[(int64_t)blk * Cs + j];
            const double n = (double)min(ppb, HW - blk * ppb), m = (double)v.x;
            s1 += n * m;
            s2 += (double)v.y + n * m * m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((tid & 63) == 0) {
        wsum[tid >> 6] = s1;
        wsum[4 + (tid >> 6)] = s2;
    }
    __syncthreads();
    s1 = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    s2 = (wsum[4] + wsum[5]) + (wsum[6] + wsum[7]);
    const double N = (double)HW * cpg;
    const double mu = s1 / N;
    const double var = fmax(s2 / N - mu * mu, 0.0);
    rstd_out = (float)(1.0 / sqrt(var + (double)eps));
    mean_out = (float)mu;
}

// grid (G, B), 256 threads.  Channels [0,C1) take their partials from set 1 (ppb1 pixels per block), [C1,C1+C2) from
// set 2.  Blocks and channels are merged with Chan's formula in fp64: mean and biased variance of the group.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ p1, int nblk1, int ppb1, int C1,
                                                         const float* __restrict__ p2, int nblk2, int ppb2, int C2,
                                                         int HW, int G, const float* __restrict__ gamma, float eps,
                                                         float* __restrict__ mean, float* __restrict__ scale) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int C = C1 + C2, cpg = C / G;
    float muf, rstd;
    gn_group_stats(p1, nblk1, ppb1, C1, p2, nblk2, ppb2, C2, HW, G, eps, g, b, muf, rstd);
    if (lane < cpg) {
        const int c = g * cpg + lane;
        mean[(int64_t)b * C + c] = muf;
        scale[(int64_t)b * C + c] = rstd * gamma[c];
    }
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }

// grid (ceil(HW * C/4 / 256), B): one float4 per thread, 32-bit index math only
template <class ST, class OT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const ST* __restrict__ in1, int C1,
                                                       const ST* __restrict__ in2, int C2, int HW, GnParams gn,
                                                       int silu, OT* __restrict__ out) {
    const unsigned C = C1 + C2, Q = C >> 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)HW * Q) return;
    const unsigned pix_s = idx / Q, cq = idx - pix_s * Q;
    const int c = cq * 4, b = blockIdx.y;
    const int64_t pix = (int64_t)b * HW + pix_s;
    float4 v;
    if (c < C1) v = St<ST>::ld4(in1 + pix * C1 + c);
    else v = St<ST>::ld4(in2 + pix * C2 + (c - C1));
    const float4 mu = *reinterpret_cast<const float4*>(gn.mean + (int64_t)b * C + c);
    const float4 sc = *reinterpret_cast<const float4*>(gn.scale + (int64_t)b * C + c);
    const float4 be = *reinterpret_cast<const float4*>(gn.beta + c);
    float4 o;
    o.x = fmaf(v.x - mu.x, sc.x, be.x);
    o.y = fmaf(v.y - mu.y, sc.y, be.y);
    o.z = fmaf(v.z - mu.z, sc.z, be.z);
    o.w = fmaf(v.w - mu.w, sc.w, be.w);
    if (silu) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
    St<OT>::st4(out + pix * C + c, o);
}

// Small images: finalize and apply in ONE launch.  grid (G, B, Z): the block reduces its group's partials exactly like
// gn_finalize, then normalises (+ SiLU) its slice z of the group's HW x cpg elements of cat[in1, in2] into `out`.  (Z > 1 for
// a single utterance: with G x B = 32 blocks the 32 x 32 level took 15.7 us for a 1 MB tensor -- a latency chain per thread on
// an eighth of the chip; every slice repeats the few-hundred-value reduction, which is cheaper than a second launch.)
template <class ST, class OT>
__global__ __launch_bounds__(256) void gn_finalize_apply_kernel(const ST* __restrict__ in1, const float* __restrict__ p1,
                                                               int nblk1, int ppb1, int C1,
                                                               const ST* __restrict__ in2, const float* __restrict__ p2,
                                                               int nblk2, int ppb2, int C2, int HW, int G,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               OT* __restrict__ out) {
    const int g = blockIdx.x, b = blockIdx.y;
    const int C = C1 + C2, cpg = C / G;
    float muf, rstd;
    gn_group_stats(p1, nblk1, ppb1, C1, p2, nblk2, ppb2, C2, HW, G, eps, g, b, muf, rstd);
    const int px0 = (int)((int64_t)HW * blockIdx.z / gridDim.z), px1 = (int)((int64_t)HW * (blockIdx.z + 1) / gridDim.z);
    // groups of whole channel quads that lie in ONE source tensor (every shape of the released net): 16- or 8-byte accesses, a
    // pixel's quads in neighbouring lanes -- the element-by-element loop below took 11 us for a [8, 16, 16, 256] tensor
    if ((cpg & 3) == 0 && (C1 & 3) == 0 && (C2 & 3) == 0 && ((g + 1) * cpg <= C1 || g * cpg >= C1)) {
        const int qpp = cpg >> 2, nq = px1 * qpp;
        const bool first = (g + 1) * cpg <= C1;
        const ST* src = first ? in1 : in2;
        const int Cs = first ? C1 : C2, cs0 = first ? g * cpg : g * cpg - C1;
        const float sc = rstd;
        for (int i = px0 * qpp + threadIdx.x; i < nq; i += 256) {
            const int pix = i / qpp, j = i - pix * qpp;
            const int64_t px = (int64_t)b * HW + pix;
            const float4 x = St<ST>::ld4(src + px * Cs + cs0 + 4 * j);
            const float4 ga = *reinterpret_cast<const float4*>(gamma + g * cpg + 4 * j);
            const float4 be = *reinterpret_cast<const float4*>(beta + g * cpg + 4 * j);
            float4 v;
            v.x = fmaf(x.x - muf, sc * ga.x, be.x);
            v.y = fmaf(x.y - muf, sc * ga.y, be.y);
            v.z = fmaf(x.z - muf, sc * ga.z, be.z);
            v.w = fmaf(x.w - muf, sc * ga.w, be.w);
            if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            St<OT>::st4(out + px * C + g * cpg + 4 * j, v);
        }
        return;
    }
    for (int i = px0 * cpg + threadIdx.x; i < px1 * cpg; i += 256) {
        const int pix = i / cpg, c = g * cpg + (i - pix * cpg);
        const int64_t px = (int64_t)b * HW + pix;
        const float x = c < C1 ? St<ST>::ld1(in1 + px * C1 + c) : St<ST>::ld1(in2 + px * C2 + (c - C1));
        float v = fmaf(x - muf, rstd * gamma[c], beta[c]);
        if (silu) v = silu_f(v);
        St<OT>::st1(out + px * C + c, v);
    }
}

int launch_gn_stats(const void* in1, int C1, const void* in2, int C2, int B, int HW, float* partial, int nblk,
                    hipStream_t s, int dt) {
    const int C = C1 + C2;
    if ((C1 & 3) || (C2 & 3) || C / 4 > GN_THREADS || C <= 0) {
        set_error("gn_stats: unsupported channels C1=%d C2=%d", C1, C2);
        return ERR_SHAPE;
    }
    FLOWSE_DT_SWITCH(dt, ST, hipLaunchKernelGGL(gn_stats_kernel<ST>, dim3(nblk, B), dim3(GN_THREADS), 0, s,
                                                static_cast<const ST*>(in1), C1, static_cast<const ST*>(in2), C2, HW,
                                                partial, nblk, gn_pixels_per_block(HW, nblk)));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_gn_finalize(const float* partial1, int nblk1, int C1, const float* partial2, int nblk2, int C2, int B,
                       int HW, int G, const float* gamma, float eps, float* mean, float* scale, hipStream_t s) {
    // every producer cuts the HW pixels of a sample into nblk equal blocks (the last one may be short)
    const int ppb1 = gn_pixels_per_block(HW, nblk1), ppb2 = nblk2 > 0 ? gn_pixels_per_block(HW, nblk2) : 1;
    const int C = C1 + C2;
    if (C % G != 0 || C / G > 64 || (C2 > 0 && !partial2)) {
        set_error("gn_finalize: unsupported C=%d G=%d", C, G);
        return ERR_SHAPE;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, s, partial1, nblk1, ppb1, C1, partial2, nblk2, ppb2,
                       C2, HW, G, gamma, eps, mean, scale);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// out_dt: DT_F32 or the input's storage type
template <class ST>
static void launch_gfa(const void* in1, const float* partial1, int nblk1, int ppb1, int C1, const void* in2,
                       const float* partial2, int nblk2, int ppb2, int C2, int B, int HW, int G, const float* gamma,
                       const float* beta, float eps, int silu, void* out, int out_dt, hipStream_t s) {
    // pixel slices per (group, sample): up to one block per CU, at least 128 pixels each
    int Z = 1;
    while (Z < 8 && G * B * Z * 2 <= 256 && HW / (Z * 2) >= 128) Z *= 2;
    if (out_dt == DT_F32)
        hipLaunchKernelGGL((gn_finalize_apply_kernel<ST, float>), dim3(G, B, Z), dim3(256), 0, s, static_cast<const ST*>(in1),
                           partial1, nblk1, ppb1, C1, static_cast<const ST*>(in2), partial2, nblk2, ppb2, C2, HW, G, gamma,
                           beta, eps, silu, static_cast<float*>(out));
    else
        hipLaunchKernelGGL((gn_finalize_apply_kernel<ST, ST>), dim3(G, B, Z), dim3(256), 0, s, static_cast<const ST*>(in1),
                           partial1, nblk1, ppb1, C1, static_cast<const ST*>(in2), partial2, nblk2, ppb2, C2, HW, G, gamma,
                           beta, eps, silu, static_cast<ST*>(out));
}

int launch_gn_finalize_apply(const void* in1, const float* partial1, int nblk1, int C1, const void* in2,
                             const float* partial2, int nblk2, int C2, int B, int HW, int G, const float* gamma,
                             const float* beta, float eps, int silu, void* out, hipStream_t s, int in_dt, int out_dt) {
    const int ppb1 = gn_pixels_per_block(HW, nblk1), ppb2 = nblk2 > 0 ? gn_pixels_per_block(HW, nblk2) : 1;
    const int C = C1 + C2;
    if (C % G != 0 || (C2 > 0 && (!partial2 || !in2))) {
        set_error("gn_finalize_apply: unsupported C=%d G=%d", C, G);
        return ERR_SHAPE;
    }
    if (out_dt != DT_F32 && out_dt != in_dt) {
        set_error("gn_finalize_apply: output type must be fp32 or the input type");
        return ERR_ARG;
    }
    FLOWSE_DT_SWITCH(in_dt, ST, launch_gfa<ST>(in1, partial1, nblk1, ppb1, C1, in2, partial2, nblk2, ppb2, C2, B, HW, G,
                                               gamma, beta, eps, silu, out, out_dt, s));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

template <class ST>
static void launch_ga(const void* in1, int C1, const void* in2, int C2, int B, int HW, GnParams gn, int silu, void* out,
                      int out_dt, unsigned gx, hipStream_t s) {
    if (out_dt == DT_F32)
        hipLaunchKernelGGL((gn_apply_kernel<ST, float>), dim3(gx, B), dim3(256), 0, s, static_cast<const ST*>(in1), C1,
                           static_cast<const ST*>(in2), C2, HW, gn, silu, static_cast<float*>(out));
    else
        hipLaunchKernelGGL((gn_apply_kernel<ST, ST>), dim3(gx, B), dim3(256), 0, s, static_cast<const ST*>(in1), C1,
                           static_cast<const ST*>(in2), C2, HW, gn, silu, static_cast<ST*>(out));
}

int launch_gn_apply(const void* in1, int C1, const void* in2, int C2, int B, int HW, GnParams gn, int silu,
                    void* out, hipStream_t s, int in_dt, int out_dt) {
    const int64_t per_sample = (int64_t)HW * ((C1 + C2) / 4);
    if (per_sample >= (1LL << 32) || B > 65535) {
        set_error("gn_apply: tensor too large");
        return ERR_SHAPE;
    }
    if (out_dt != DT_F32 && out_dt != in_dt) {
        set_error("gn_apply: output type must be fp32 or the input type");
        return ERR_ARG;
    }
    FLOWSE_DT_SWITCH(in_dt, ST, launch_ga<ST>(in1, C1, in2, C2, B, HW, gn, silu, out, out_dt,
                                              (unsigned)((per_sample + 255) / 256), s));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
