// STFT + magnitude compression, and its inverse, as two fused kernels.
//
// Replaces (reference): SpecsDataModule.stft -> spec_fwd and spec_back -> istft (flowmse/data_module.py:149-175,
// 199-205; model.py:190-203), i.e. torch.stft(n_fft=510, hop=128, periodic hann, center=True [reflect], onesided)
// followed by  c * |z|^e * exp(j arg z)  (e = 0.5, c = 0.15), the zero padding of pad_spec (util/other.py:83-90),
// and the inverse chain ending in torch.istft(..., length).  The steps on either side of the sampler
// (SURVEY.md section 8(f), rank 1).  O(n_fft^2) direct DFTs with an exact 510-entry twiddle table: 0.13 MFLOP per
// frame -- microseconds per utterance, no FFT library, no intermediate tensors.
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace flowse {

constexpr int NFFT = 510, HOP = 128, NBIN = 256, PADC = NFFT / 2;   // 255 samples of centre padding

// [3][NFFT]: window, cos(2 pi k / NFFT), sin(2 pi k / NFFT) -- one copy per device, created under a lock on the
// first call made with that device current (a pointer of one device must never be handed to another's kernels)
static std::mutex g_tab_mu;
static std::map<int, float*> g_tab_of_device;

static int ensure_tables(float** out) {
    int dev = 0;
    FLOWSE_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tab_mu);
    auto it = g_tab_of_device.find(dev);
    if (it != g_tab_of_device.end()) {
        *out = it->second;
        return OK;
    }
    std::vector<float> h(3 * NFFT);
    for (int k = 0; k < NFFT; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)NFFT;
        h[k] = (float)(0.5 - 0.5 * cos(a));           // periodic Hann
        h[NFFT + k] = (float)cos(a);
        h[2 * NFFT + k] = (float)sin(a);
    }
    float* d = nullptr;
    FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(float)));
    FLOWSE_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    g_tab_of_device[dev] = d;
    *out = d;
    return OK;
}

// grid (Tpad, B), 256 threads = 256 bins.  Frames t >= T are the zero padding of pad_spec.
__global__ __launch_bounds__(256) void stft_compress_kernel(const float* __restrict__ sig, int L, float scale_in,
                                                            const float* __restrict__ tab, float2* __restrict__ out,
                                                            int T, int Tpad, float factor, float exponent) {
    __shared__ float xw[NFFT], ct[NFFT], st[NFFT];
    const int t = blockIdx.x, b = blockIdx.y, f = threadIdx.x;
    float2* dst = out + ((int64_t)b * NBIN + f) * Tpad + t;
    if (t >= T) {
        *dst = make_float2(0.f, 0.f);
        return;
    }
    for (int n = threadIdx.x; n < NFFT; n += 256) {
        int j = t * HOP + n - PADC;                    // centre=True, reflect padding
        if (j < 0) j = -j;
        if (j >= L) j = 2 * (L - 1) - j;
        xw[n] = sig[(int64_t)b * L + j] * scale_in * tab[n];
        ct[n] = tab[NFFT + n];
        st[n] = tab[2 * NFFT + n];
    }
    __syncthreads();
    float re = 0.f, im = 0.f;
    int idx = 0;                                       // (f * n) mod NFFT
    for (int n = 0; n < NFFT; ++n) {
        re = fmaf(xw[n], ct[idx], re);
        im = fmaf(-xw[n], st[idx], im);
        idx += f;
        if (idx >= NFFT) idx -= NFFT;
    }
    // spec_fwd: factor * |z|^e * exp(j arg z) = z * factor * |z|^(e-1)
    const float mag = sqrtf(re * re + im * im);
    float s = factor;
    if (exponent != 1.f) s = mag > 0.f ? factor * powf(mag, exponent - 1.f) : 0.f;
    *dst = make_float2(re * s, im * s);
}

// grid (ceil(Lout / 128), B), 256 threads: sample = tid & 127, the two halves split the bins.
__global__ __launch_bounds__(256) void istft_decompress_kernel(const float2* __restrict__ spec, int T, int Tpad,
                                                               float factor, float exponent,
                                                               const float* __restrict__ tab, float* __restrict__ out,
                                                               int Lout, float scale_out) {
    constexpr int NF = 5;                               // frames that can overlap a block of HOP samples
    __shared__ float2 Xs[NF][NBIN];
    __shared__ float ct[NFFT], st[NFFT], win[NFFT];
    __shared__ float part[256];
    const int b = blockIdx.y, n0 = blockIdx.x * HOP;
    const int tid = threadIdx.x;
    // frame t covers output samples [128 t - 255, 128 t + 254]
    int t_lo = (n0 - (NFFT - 1 - PADC) + HOP - 1) / HOP;          // ceil((n0 - 254) / 128), may be negative
    if (n0 - (NFFT - 1 - PADC) < 0) t_lo = 0;
    for (int i = tid; i < NFFT; i += 256) {
        win[i] = tab[i];
        ct[i] = tab[NFFT + i];
        st[i] = tab[2 * NFFT + i];
    }
    for (int i = tid; i < NF * NBIN; i += 256) {
        const int fr = i / NBIN, f = i - fr * NBIN;
        const int t = t_lo + fr;
        float2 z = make_float2(0.f, 0.f);
        if (t < T) {
            z = spec[((int64_t)b * NBIN + f) * Tpad + t];
            // spec_back: (|z| / factor)^(1/e) * exp(j arg z)
            z.x /= factor;
            z.y /= factor;
            if (exponent != 1.f) {
                const float mag = sqrtf(z.x * z.x + z.y * z.y);
                const float s = mag > 0.f ? powf(mag, 1.f / exponent - 1.f) : 0.f;
                z.x *= s;
                z.y *= s;
            }
        }
        Xs[fr][f] = z;
    }
    __syncthreads();
    const int n = n0 + (tid & 127), half = tid >> 7;
    float acc = 0.f, env = 0.f;
    for (int fr = 0; fr < NF; ++fr) {
        const int t = t_lo + fr;
        const int k = n + PADC - t * HOP;               // position inside frame t
        if (t >= T || k < 0 || k >= NFFT) continue;
        const float w = win[k];
        env = fmaf(w, w, env);
        // irfft: 1/N [Re X0 + (-1)^k Re X_{N/2} + 2 sum_{f=1}^{N/2-1} (Re X_f cos - Im X_f sin)(2 pi f k / N)]
        const int f0 = half * 128;
        int idx = (int)(((int64_t)f0 * k) % NFFT);
        float s = 0.f;
        for (int f = f0; f < f0 + 128; ++f) {
            const float2 z = Xs[fr][f];
            const float c = (f == 0 || f == NBIN - 1) ? 1.f : 2.f;
            const float zi = (f == 0 || f == NBIN - 1) ? 0.f : z.y;
            s += c * (z.x * ct[idx] - zi * st[idx]);
            idx += k;
            if (idx >= NFFT) idx -= NFFT;
        }
        acc = fmaf(s * (1.f / NFFT), w, acc);
    }
    part[tid] = acc;
    __syncthreads();
    if (half == 0 && n < Lout) {
        const float v = part[tid] + part[tid + 128];
        out[(int64_t)b * Lout + n] = env > 1e-11f ? v / env * scale_out : 0.f;
    }
}

int launch_stft_compress(const float* sig, int B, int L, float scale_in, float* out_c64, int T, int Tpad, float factor,
                         float exponent, hipStream_t s) {
    if (L <= PADC || T != L / HOP + 1 || Tpad < T || B < 1 || B > 65535) {
        set_error("stft: need L > %d, T == L / %d + 1 (got L=%d T=%d Tpad=%d B=%d)", PADC, HOP, L, T, Tpad, B);
        return ERR_SHAPE;
    }
    float* tab = nullptr;
    int rc = ensure_tables(&tab);
    if (rc != OK) return rc;
    hipLaunchKernelGGL(stft_compress_kernel, dim3(Tpad, B), dim3(256), 0, s, sig, L, scale_in, tab,
                       reinterpret_cast<float2*>(out_c64), T, Tpad, factor, exponent);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_istft_decompress(const float* spec_c64, int B, int T, int Tpad, float factor, float exponent, float* out,
                            int Lout, float scale_out, hipStream_t s) {
    if (T < 1 || Tpad < T || Lout < 1 || Lout > (T - 1) * HOP + NFFT - PADC || B < 1 || B > 65535 || factor == 0.f) {
        set_error("istft: bad shape T=%d Tpad=%d Lout=%d B=%d", T, Tpad, Lout, B);
        return ERR_SHAPE;
    }
    float* tab = nullptr;
    int rc = ensure_tables(&tab);
    if (rc != OK) return rc;
    hipLaunchKernelGGL(istft_decompress_kernel, dim3((Lout + HOP - 1) / HOP, B), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(spec_c64), T, Tpad, factor, exponent, tab, out, Lout,
                       scale_out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
