// Small HBM-bound / latency-bound kernels around the network body.
#include "common.h"

namespace flowse {

static int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---------------------------------------------------------------------------------------------------
// Feature pack: complex x, y [B,1,F,T] (interleaved re,im) -> real NHWC [B][F][T][4] = (x.re, x.im, y.re, y.im).
// Replaces (reference): torch.cat([x, y], 1) of VFModel.forward (flowmse/model.py:166) + the real/imag
// split of NCSNpp.forward (flowmse/backbones/ncsnpp.py:252-254).
__global__ __launch_bounds__(256) void pack_input_kernel(const float2* __restrict__ x, const float2* __restrict__ y,
                                                         int64_t n, float4* __restrict__ out,
                                                         const CallBlock* __restrict__ cb) {
    if (cb) {
        x = reinterpret_cast<const float2*>(cb->x);
        y = reinterpret_cast<const float2*>(cb->y);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float2 a = x[i], b = y[i];
        out[i] = make_float4(a.x, a.y, b.x, b.y);
    }
}

int launch_pack_input(const float* x, const float* y, int B, int F, int T, float* out4, hipStream_t s,
                      const CallBlock* cb) {
    const int64_t n = (int64_t)B * F * T;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const float2*>(x),
                       reinterpret_cast<const float2*>(y), n, reinterpret_cast<float4*>(out4), cb);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Gaussian Fourier time embedding.  Replaces (reference): torch.log(t) (ncsnpp.py:259) +
// GaussianFourierProjection.forward (layerspp.py:39-41): x_proj = ((log t * W) * 2) * pi in fp32,
// out = [sin(x_proj), cos(x_proj)].  log / sin / cos are evaluated in fp64 from the fp32 operands and
// rounded once, so the result is the correctly rounded fp32 value of the reference's expression.
__global__ void gfp_kernel(const float* __restrict__ t, const float* __restrict__ Wf, int E, float* __restrict__ out,
                           const CallBlock* __restrict__ cb) {
    if (cb) t = cb->t;
    const int b = blockIdx.x;
    const float lt = (float)log((double)t[b]);
    for (int j = threadIdx.x; j < E; j += blockDim.x) {
        const float p = __fmul_rn(__fmul_rn(__fmul_rn(lt, Wf[j]), 2.0f), 3.14159274101257324f);
        out[(int64_t)b * 2 * E + j] = (float)sin((double)p);
        out[(int64_t)b * 2 * E + E + j] = (float)cos((double)p);
    }
}

int launch_gfp(const float* t, const float* Wf, int B, int E, float* out, hipStream_t s, const CallBlock* cb) {
    hipLaunchKernelGGL(gfp_kernel, dim3(B), dim3(128), 0, s, t, Wf, E, out, cb);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Row-parallel linear layer for the time-embedding MLP (ncsnpp.py:262-267) and the 49 per-block
// Dense_0(SiLU(temb)) projections (layerspp.py:262-263), all of which depend only on t: they are evaluated
// once per solver step into one bias table that the Conv_0 epilogues read.
//   out[b*out_stride + r] = act( sum_k W[r][k] * in[b][k] + bias[r] ),   one wave per row r, all b.
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, int B, int K,
                                                     const float* __restrict__ W, const float* __restrict__ bias,
                                                     int R, int act, float* __restrict__ out, int out_stride) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    const float* wr = W + (int64_t)r * K;
    // eight samples per pass over the row (one sample per pass re-read the row and waited a memory round trip per sample:
    // 40 us for the [10 752][512] table at batch 8); per (sample, row) the same fmaf chain and shuffle tree as before
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float w = wr[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, in[(int64_t)min(b0 + j, B - 1) * K + k], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && b0 + j < B) {
                v += bias ? bias[r] : 0.f;
                if (act) v = silu_f(v);
                out[(int64_t)(b0 + j) * out_stride + r] = v;
            }
        }
    }
}

int launch_linear(const float* in, int B, int K, const float* W, const float* bias, int R, int act, float* out,
                  int out_stride, hipStream_t s) {
    hipLaunchKernelGGL(linear_kernel, dim3((R + 3) / 4), dim3(256), 0, s, in, B, K, W, bias, R, act, out, out_stride);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Output head, fused with the solver update.  Replaces (reference): h / t (ncsnpp.py:398), output_layer
// conv1x1 4->2 (:401), permute + view_as_complex (:402-403), the negation of VFModel.forward (model.py:169)
// and EulerODEsolver.update_fn  x + VF * (-stepsize)  (sampling/odesolvers.py:42-47).
__global__ __launch_bounds__(256) void head_kernel(const float4* __restrict__ pyr, const float* __restrict__ t,
                                                   const float* __restrict__ Wout, const float* __restrict__ bout,
                                                   int64_t n, int64_t per_sample, int mode,
                                                   const float2* __restrict__ x, float dt, float2* __restrict__ out,
                                                   const CallBlock* __restrict__ cb) {
    const float2* acc_in = nullptr;
    float2* acc_out = nullptr;
    float ca = 0.f, cbk = 0.f;
    if (cb) {
        t = cb->t;
        x = reinterpret_cast<const float2*>(cb->x);
        out = reinterpret_cast<float2*>(cb->out);
        mode = cb->mode;
        dt = cb->dt;
        if (mode == 3) {                 // Runge-Kutta stage: x = the step's base point, not the network input
            x = reinterpret_cast<const float2*>(cb->x0);
            acc_in = reinterpret_cast<const float2*>(cb->acc_in);
            acc_out = reinterpret_cast<float2*>(cb->acc_out);
            ca = cb->a;
            cbk = cb->b;
        }
    }
    const float w00 = Wout[0], w01 = Wout[1], w02 = Wout[2], w03 = Wout[3];
    const float w10 = Wout[4], w11 = Wout[5], w12 = Wout[6], w13 = Wout[7];
    const float b0 = bout[0], b1 = bout[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float tb = t[i / per_sample];
        float4 p = pyr[i];
        p.x = p.x / tb; p.y = p.y / tb; p.z = p.z / tb; p.w = p.w / tb;
        float re = fmaf(w03, p.w, fmaf(w02, p.z, fmaf(w01, p.y, w00 * p.x))) + b0;
        float im = fmaf(w13, p.w, fmaf(w12, p.z, fmaf(w11, p.y, w10 * p.x))) + b1;
        if (mode == 1) {
            re = -re;
            im = -im;
        } else if (mode == 2) {
            const float2 xv = x[i];
            re = xv.x + __fmul_rn(re, dt);
            im = xv.y + __fmul_rn(im, dt);
        } else if (mode == 3) {
            if (acc_out) {
                const float2 av = acc_in[i];
                acc_out[i] = make_float2(av.x + __fmul_rn(re, cbk), av.y + __fmul_rn(im, cbk));
            }
            if (out) {
                const float2 xv = x[i];
                out[i] = make_float2(xv.x + __fmul_rn(re, ca), xv.y + __fmul_rn(im, ca));
            }
            continue;
        }
        out[i] = make_float2(re, im);
    }
}

int launch_head(const float* pyr4, const float* t, const float* Wout, const float* bout, int B, int F, int T, int mode,
                const float* x, float dt, float* out, hipStream_t s, const CallBlock* cb) {
    const int64_t n = (int64_t)B * F * T;
    hipLaunchKernelGGL(head_kernel, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const float4*>(pyr4), t, Wout,
                       bout, n, (int64_t)F * T, mode, reinterpret_cast<const float2*>(x), dt,
                       reinterpret_cast<float2*>(out), cb);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Per-call state kept in device memory (see CallBlock in common.h).  The new value travels as a kernel argument,
// i.e. it is copied at launch time: no host buffer has to outlive the call and nothing synchronises.
__global__ void set_call_kernel(CallBlock* cb, CallBlock v) { *cb = v; }

int launch_set_call(CallBlock* d_cb, const CallBlock& value, hipStream_t s) {
    hipLaunchKernelGGL(set_call_kernel, dim3(1), dim3(1), 0, s, d_cb, value);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

struct TimeChunk { float v[64]; };
__global__ void fill_times_kernel(float* __restrict__ d_ts, TimeChunk c, int n, int B) {
    for (int i = threadIdx.x; i < n * B; i += blockDim.x) d_ts[i] = c.v[i / B];
}

int launch_fill_times(float* d_ts, const float* ts, int N, int B, hipStream_t s) {
    for (int i0 = 0; i0 < N; i0 += 64) {
        TimeChunk c;
        const int n = N - i0 < 64 ? N - i0 : 64;
        for (int i = 0; i < 64; ++i) c.v[i] = i < n ? ts[i0 + i] : 0.f;
        hipLaunchKernelGGL(fill_times_kernel, dim3(1), dim3(256), 0, s, d_ts + (size_t)i0 * B, c, n, B);
        FLOWSE_LAUNCH_CHECK();
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Prior sample x_T = y + sigma(1) * z.  Replaces (reference): FLOWMATCHING.prior_sampling (flowmse/odes.py:93-100)
// with the noise z supplied by the caller.
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                   float sigma, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = y[i] + __fmul_rn(z[i], sigma);
}

int launch_axpy(const float* y, const float* z, float sigma, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, z, sigma, n, out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
