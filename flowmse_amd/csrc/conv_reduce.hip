// Deterministic second pass of the split-K convolutions: sums the K slices in slice order and applies the epilogue
// (bias, per-sample time-embedding bias, residual, scale: flowmse/backbones/ncsnpp_utils/layerspp.py:262-274), optionally
// with the GroupNorm partial statistics of the tensor written (splitk_reduce_stats) or with the whole consuming
// GroupNorm (+ SiLU) of a ResnetBlock (splitk_reduce_gn, layerspp.py:265).
#include "conv_common.h"

namespace flowse {

// out = (sum_s partial[s] + bias + bias2 + res) * scale, float4 streams; grid (ceil(HW * Cout/4 / 256), B)
template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a) {
    const unsigned Q = a.Cout >> 2;
    const unsigned HW = a.H * a.W;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= HW * Q) return;
    const int b = blockIdx.y;
    const int n = (idx % Q) * 4;
    const int64_t i4 = ((int64_t)b * HW * Q + idx) * 4;            // float offset of this quad
    const int64_t slice = (int64_t)a.B * HW * a.Cout;
    float4 v = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
    for (int s = 1; s < a.ksplit; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
#pragma unroll 4
    for (int s = 0; s < a.ksplit2; ++s) {                 // the shortcut conv's slices (ConvArgs::partial2)
        const float4 t = *reinterpret_cast<const float4*>(a.partial2 + s * slice + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias_x) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias_x + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias2) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.res) {
        const float4 t = St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
    St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, v);
}

// Same reduction, organised like gn_stats (grid (HW / PB, B); a thread owns one channel quad and strides over the
// block's PB pixels) so that it can also emit the GroupNorm partial sums of the tensor it writes:
// stats[((b * nblk + blk) * Cout + c) * 2 + {0,1}].
// pixels per block of the split-K reduction: small images want many blocks (parallelism), larger ones few
// partials (every partial is later read by gn_finalize)
template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(ConvArgs a, int PB) {
    __shared__ float red[256 * 8];
    const int Q = a.Cout >> 2, PR = 256 / Q;
    const int HW = a.H * a.W;
    const int tid = threadIdx.x;
    const int pr = tid / Q, cq = tid - pr * Q;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int n = cq * 4;
    const int64_t slice = (int64_t)a.B * HW * a.Cout;
    Stat4 st;
    st.init();
    if (pr < PR) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + n);
        if (a.bias2) {
            const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
            bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
        }
        if (a.bias_x) {
            const float4 t = *reinterpret_cast<const float4*>(a.bias_x + n);
            bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
        }
        for (int p = blk * PB + pr; p < (blk + 1) * PB; p += PR) {
            const int64_t i4 = ((int64_t)b * HW + p) * a.Cout + n;
            float4 v = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
            for (int s = 1; s < a.ksplit; ++s) {           // independent loads: keep many in flight
                const float4 t = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
#pragma unroll 4
            for (int s = 0; s < a.ksplit2; ++s) {          // the shortcut conv's slices (ConvArgs::partial2)
                const float4 t = *reinterpret_cast<const float4*>(a.partial2 + s * slice + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (a.res) {
                const float4 t = St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, v);
            st.add(St<OT>::rnd4(v));
        }
    }
    float* mine = red + tid * 8;
    st.finish(mine);
    __syncthreads();
    if (pr == 0) {
        float acc8[8], nacc = (float)st.n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = mine[j];
        for (int r = 1; r < PR; ++r) {
            const int cnt = r < PB ? (PB - r + PR - 1) / PR : 0;         // pixels lane r visited
            chan_merge4(nacc, acc8, (float)cnt, red + (r * Q + cq) * 8);
        }
        float* dst = a.stats + (((int64_t)b * a.stats_nblk + blk) * a.Cout + n) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[2 * j] = acc8[j];
            dst[2 * j + 1] = acc8[4 + j];
        }
    }
}

// Reduction fused with the consuming GroupNorm (see launch_splitk_reduce_gn in common.h).  grid (G, B), 256 threads;
// a thread owns up to RG_MAXQ channel quads of the group's H*W x cpg elements in REGISTERS (no second pass over the
// slices, no LDS tile), the group's sum / sum of squares are reduced in fp64 (exact products of fp32 values; the
// subtraction mean^2 loses log2(mean^2 / var) of 53 bits -- see gn_group_stats in norm.hip).
constexpr int RG_MAXQ = 32;
bool conv_reduce_gn_ok(int B, int HW, int Cout) {
    const int G = Cout / 4 < 32 ? Cout / 4 : 32;
    if (G <= 0 || (Cout % G) != 0) return false;
    // one block per (group, sample): with fewer than ~128 blocks the launch is slower than the two it replaces -- measured
    // at [1,1,256,256] (32 blocks, each pulling up to 512 KB of slices through one CU): 8.27 k vs 8.67 k frames/s
    if ((int64_t)B * G < 128) return false;
    const int cpg = Cout / G;
    return (cpg & 3) == 0 && (int64_t)HW * (cpg / 4) <= 256 * RG_MAXQ;
}

template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(ConvArgs a, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               int apply, float* __restrict__ gn_mean,
                                                               float* __restrict__ gn_scale) {
    __shared__ double wsum[8];
    const int G = gridDim.x, g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int HW = a.H * a.W, Cout = a.Cout, cpg = Cout / G, qpg = cpg >> 2;
    const int items = HW * qpg;                          // channel quads of this (sample, group)
    const int64_t slice = (int64_t)a.B * HW * Cout;
    float4 v[RG_MAXQ];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < RG_MAXQ; ++k) {
        const int it = tid + k * 256;
        if (it < items) {
            const int p = it / qpg, q = it - p * qpg;
            const int n = g * cpg + q * 4;
            const int64_t i4 = ((int64_t)b * HW + p) * Cout + n;
            float4 t = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
            for (int s = 1; s < a.ksplit; ++s) {         // independent loads: keep many in flight
                const float4 u = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (a.bias) {
                const float4 u = *reinterpret_cast<const float4*>(a.bias + n);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (a.bias2) {
                const float4 u = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            t.x *= a.scale; t.y *= a.scale; t.z *= a.scale; t.w *= a.scale;
            if (!apply) {                                // the consumer reads this tensor: statistics of what is stored
                St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, t);
                t = St<OT>::rnd4(t);
            }
            v[k] = t;
            s1 += ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
            s2 += ((double)t.x * t.x + (double)t.y * t.y) + ((double)t.z * t.z + (double)t.w * t.w);
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
    const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (!apply) {
        if (tid < cpg) {
            const int c = g * cpg + tid;
            gn_mean[(int64_t)b * Cout + c] = mean;
            gn_scale[(int64_t)b * Cout + c] = rstd * gamma[c];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < RG_MAXQ; ++k) {
        const int it = tid + k * 256;
        if (it < items) {
            const int p = it / qpg, q = it - p * qpg;
            const int n = g * cpg + q * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + n), be = *reinterpret_cast<const float4*>(beta + n);
            float4 t = v[k];
            t.x = fmaf(t.x - mean, rstd * ga.x, be.x);
            t.y = fmaf(t.y - mean, rstd * ga.y, be.y);
            t.z = fmaf(t.z - mean, rstd * ga.z, be.z);
            t.w = fmaf(t.w - mean, rstd * ga.w, be.w);
            if (silu) {
                t.x = t.x / (1.f + expf(-t.x)); t.y = t.y / (1.f + expf(-t.y));
                t.z = t.z / (1.f + expf(-t.z)); t.w = t.w / (1.f + expf(-t.w));
            }
            St<OT>::st4(reinterpret_cast<OT*>(a.out) + ((int64_t)b * HW + p) * Cout + n, t);
        }
    }
}

int launch_splitk_reduce_gn(const ConvArgs& a, const float* gamma, const float* beta, float eps, int silu, int apply,
                            float* gn_mean, float* gn_scale, hipStream_t s) {
    const int HW = a.H * a.W;
    // (the kernel sums ONE set of slices and adds neither a residual nor the shortcut's partial set / bias: reject them
    // instead of dropping their contribution silently)
    if (!a.partial || a.ksplit < 1 || a.res || a.partial2 || a.ksplit2 || a.bias_x ||
        !conv_reduce_gn_ok(a.B, HW, a.Cout) || !gamma || (apply && !beta) || (!apply && (!gn_mean || !gn_scale))) {
        set_error("splitk_reduce_gn: unsupported arguments (HW=%d Cout=%d ksplit=%d)", HW, a.Cout, a.ksplit);
        return ERR_ARG;
    }
    const int G = a.Cout / 4 < 32 ? a.Cout / 4 : 32;
    FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_gn_kernel<OT>, dim3(G, a.B), dim3(256), 0, s, a, gamma,
                                                      beta, eps, silu, apply, gn_mean, gn_scale));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_splitk_reduce(const ConvArgs& a, hipStream_t s) {
    const int HW = a.H * a.W;
    if (a.stats) {
        const int PB = sk_pixels_per_block(HW);
        if (a.stats_nblk != HW / PB || (HW % PB) != 0 || a.Cout / 4 > 256) {
            set_error("splitk_reduce: inconsistent fused-stats geometry");
            return ERR_ARG;
        }
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_stats_kernel<OT>, dim3(HW / PB, a.B), dim3(256), 0, s,
                                                          a, PB));
        FLOWSE_LAUNCH_CHECK();
        return OK;
    }
    const unsigned per_sample = (unsigned)HW * (a.Cout / 4);
    FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_kernel<OT>, dim3((per_sample + 255) / 256, a.B),
                                                      dim3(256), 0, s, a));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
