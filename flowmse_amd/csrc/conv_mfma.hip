// 3x3 / 1x1 convolution as an implicit GEMM on the CDNA4 fp32 matrix cores.
//
//   out[m][n] = ( sum_{tap, ci} A[m][tap, ci] * Wp[n][tap][ci] + bias[n] + bias2[b(m)][n] + res[m][n] ) * scale
//
//   m = flattened NHWC pixel (b, y, x), n = output channel, K = taps * Cin with the channel axis contiguous
//   in both operands.  Exact fp32: v_mfma_f32_32x32x2_f32 is bitwise an fmaf chain (no TF32 on gfx950).
//
// Replaces (reference): ddpm_conv3x3 / ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-124) and
// NIN (layers.py:546-555) as used by ResnetBlockBigGANpp (layerspp.py:245-274), AttnBlockpp (:75-91), the
// progressive-output heads (ncsnpp.py:345-366) incl. the channel concat of ncsnpp.py:337 (two-source A operand)
// and the per-sample time-embedding bias of layerspp.py:262-263 (bias2) and the (x + h)/sqrt(2) skip (res, scale).
//
// Tiling: block = 4 waves (256 threads); wave tile = TM x TN MFMA tiles of 32x32; block tile BM x BN =
// (WM*TM*32) x (WN*TN*32).  K is walked in steps of (tap, 32-channel chunk): the A tile [BM][32] is gathered
// straight from the shifted NHWC pixels (zero outside the image = conv padding), the B tile [BN][32] from the
// packed weights; both are register-staged into double-buffered LDS with a row stride of 36 floats, which
// makes the fragment ds_read_b128 conflict-free.  Each lane reads 4 consecutive k of its row once and feeds
// 4 successive MFMAs with them (lanes 0-31 carry k = 8j+e, lanes 32-63 carry k = 8j+4+e, for A and B alike).
#include "common.h"

namespace flowse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;          // channels per K step
constexpr int LDS_ROW = 36;     // floats per LDS tile row (KC + 4 pad)

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_kernel(ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    constexpr int A_LOADS = BM * 8 / NT, B_LOADS = (BN * 8 + NT - 1) / NT;
    static_assert(BM * 8 % NT == 0, "A tile must split evenly");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDS_ROW]
    float* Bs = smem + 2 * BM * LDS_ROW;      // [2][BN][LDS_ROW]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int Cin = a.C1 + a.C2;
    const int taps = a.taps;
    const int n_ntiles = (a.Cout + BN - 1) / BN;
    const int mt = blockIdx.x / n_ntiles, nt = blockIdx.x - mt * n_ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread gather bookkeeping: this thread always loads channel quad `col4` of rows row0 + 32q
    const int col4 = tid & 7, row0 = tid >> 3;
    int pm[A_LOADS], py[A_LOADS], px[A_LOADS];
#pragma unroll
    for (int q = 0; q < A_LOADS; ++q) {
        const int m = m0 + row0 + 32 * q;
        pm[q] = m;
        if (m < M) {
            const int rem = m % HW;
            py[q] = rem / W;
            px[q] = rem - py[q] * W;
        } else {
            py[q] = -(1 << 20);
            px[q] = 0;
        }
    }

    float4 ra[A_LOADS], rb[B_LOADS];
    unsigned okmask = 0;        // bit q: ra[q] valid, bit 16+q: rb[q] valid (applied when staging into LDS)

    // Loads are UNCONDITIONAL on a clamped (always valid) address; the zero-masking happens in lstore(), i.e.
    // AFTER the MFMA block, so the loads stay in flight under the matrix work.  (A load under a runtime branch,
    // or a select right behind it, makes hipcc wait vmcnt(0) per element / ahead of the MFMAs.)
    auto gload = [&](int s) {
        const int chunk = s / taps, tap = s - chunk * taps;
        int dy = 0, dx = 0;
        if (taps == 9) {
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
        const int c = chunk * KC + col4 * 4;
        const bool cvalid = c < Cin;
        const bool second = cvalid && c >= a.C1;
        const float* src = second ? a.in2 : a.in1;
        const int cs = second ? a.C2 : a.C1;
        const int cc = cvalid ? (second ? c - a.C1 : c) : 0;
        const int shift = dy * W + dx;
        okmask = 0;
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q) {
            const int yy = py[q] + dy, xx = px[q] + dx;
            const bool ok = cvalid && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const int64_t off = ok ? (int64_t)(pm[q] + shift) * cs + cc : 0;
            ra[q] = *reinterpret_cast<const float4*>(src + off);
            okmask |= ok ? (1u << q) : 0u;
        }
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) {
            const int r = row0 + 32 * q;
            const int n = n0 + r;
            const bool ok = r < BN && cvalid && n < a.Cout;
            const int64_t off = ok ? ((int64_t)n * taps + tap) * Cin + c : 0;
            rb[q] = *reinterpret_cast<const float4*>(a.w + off);
            okmask |= ok ? (1u << (16 + q)) : 0u;
        }
    };
    auto lstore = [&](int buf) {
        float* Ab = As + buf * BM * LDS_ROW;
        float* Bb = Bs + buf * BN * LDS_ROW;
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q) {
            const bool ok = (okmask >> q) & 1u;
            float4 v = ra[q];
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            *reinterpret_cast<float4*>(Ab + (row0 + 32 * q) * LDS_ROW + col4 * 4) = v;
        }
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) {
            const int r = row0 + 32 * q;
            const bool ok = (okmask >> (16 + q)) & 1u;
            float4 v = rb[q];
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            if (r < BN) *reinterpret_cast<float4*>(Bb + r * LDS_ROW + col4 * 4) = v;
        }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, kh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (Cin + KC - 1) / KC;
    const int S = nchunks * taps;

    gload(0);
    lstore(0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) gload(s + 1);          // global loads stay in flight under the MFMAs
        const float* Ab = As + buf * BM * LDS_ROW + (wm * TM * 32 + li) * LDS_ROW + kh * 4;
        const float* Bb = Bs + buf * BN * LDS_ROW + (wn * TN * 32 + li) * LDS_ROW + kh * 4;
#pragma unroll
        for (int j = 0; j < KC / 8; ++j) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_ROW + j * 8);
#pragma unroll
            for (int i = 0; i < TN; ++i)
                bf[i] = *reinterpret_cast<const float4*>(Bb + i * 32 * LDS_ROW + j * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[jn].x, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[jn].y, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[jn].z, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[jn].w, acc[i][jn], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the staging (and its vmcnt wait) behind the MFMA block
        if (s + 1 < S) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  The accumulators go through LDS (C/D layout of the 32x32 MFMA: col = lane & 31,
    // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) so that bias / per-sample bias / residual are read and the
    // result is written as coalesced float4 rows of the NHWC output.
    constexpr int CROW = BN + 4;
    static_assert(BM * CROW <= 2 * (BM + BN) * LDS_ROW, "C tile must fit in the staging buffers");
    float* Cs = smem;                          // [BM][CROW]; safe: the last loop iteration ended with a barrier
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                Cs[row * CROW + (wn * TN + jn) * 32 + li] = acc[i][jn][r];
            }
    __syncthreads();
    constexpr int C4 = BN / 4;                 // float4 per tile row
    constexpr int RPP = NT / C4;               // rows per pass
    const int ec4 = tid % C4, er0 = tid / C4;
    const int n = n0 + ec4 * 4;
    if (n < a.Cout) {                          // Cout % 4 == 0: a quad is entirely inside or outside
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + n);
        const bool has_b2 = a.bias2 != nullptr, has_res = a.res != nullptr;
#pragma unroll 4
        for (int rr = er0; rr < BM; rr += RPP) {
            const int m = m0 + rr;
            if (m >= M) break;
            float4 v = *reinterpret_cast<const float4*>(Cs + rr * CROW + ec4 * 4);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (has_b2) {
                const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)(m / HW) * a.bias2_stride + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const int64_t o = (int64_t)m * a.Cout + n;
            if (has_res) {
                const float4 t = *reinterpret_cast<const float4*>(a.res + o);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            *reinterpret_cast<float4*>(a.out + o) = v;
        }
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * LDS_ROW * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        FLOWSE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, TM, TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    if ((a.C1 & 3) || (a.C2 & 3) || (a.Cout & 3) || (a.bias2 && (a.bias2_stride & 3)) || (a.taps != 1 && a.taps != 9) ||
        a.C1 <= 0 || (a.in2 == nullptr && a.C2 != 0)) {
        set_error("conv: unsupported channel counts C1=%d C2=%d taps=%d", a.C1, a.C2, a.taps);
        return ERR_SHAPE;
    }
    if ((int64_t)a.B * a.H * a.W >= (1LL << 31) / 4) {
        set_error("conv: too many pixels for 32-bit pixel indices");
        return ERR_SHAPE;
    }
    if (a.Cout <= 32) return launch_cfg<4, 1, 1, 1>(a, s);
    if (a.Cout <= 64) return launch_cfg<2, 2, 2, 1>(a, s);
    return launch_cfg<2, 2, 2, 2>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// Direct VALU convolution for 4 input channels: the input layer conv3x3 4->nf (ncsnpp.py:159,285) and the
// Combine conv1x1 4->C (layerspp.py:44-59).  K = 36 / 4 is too short for the matrix cores; these layers are
// bound by the HBM write of the output.  One thread = one pixel x 4 output channels; weights live in LDS.
template <int TAPS>
__global__ __launch_bounds__(256) void conv_cin4_kernel(ConvArgs a, int Q) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cout][TAPS][4]
    const int tid = threadIdx.x;
    const int nw4 = a.Cout * TAPS;
    for (int i = tid; i < nw4; i += 256)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(a.w)[i];
    __syncthreads();
    const int ppb = 256 / Q;
    const int H = a.H, W = a.W, HW = H * W;
    const int64_t M = (int64_t)a.B * HW;
    const int64_t m = (int64_t)blockIdx.x * ppb + tid / Q;
    const int cq = tid % Q;
    if (m >= M) return;
    const int rem = (int)(m % HW);
    const int y = rem / W, x = rem - y * W;
    float4 in[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int dy = TAPS == 9 ? t / 3 - 1 : 0, dx = TAPS == 9 ? t % 3 - 1 : 0;
        const int yy = y + dy, xx = x + dx;
        in[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            in[t] = *reinterpret_cast<const float4*>(a.in1 + (m + dy * W + dx) * 4);
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = cq * 4 + j;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + (n * TAPS + t) * 4);
            acc = fmaf(in[t].x, w4.x, acc);
            acc = fmaf(in[t].y, w4.y, acc);
            acc = fmaf(in[t].z, w4.z, acc);
            acc = fmaf(in[t].w, w4.w, acc);
        }
        if (a.bias) acc += a.bias[n];
        if (a.bias2) acc += a.bias2[(m / HW) * a.bias2_stride + n];
        o[j] = acc;
    }
    const int64_t off = m * a.Cout + cq * 4;
    if (a.res) {
        const float4 r = *reinterpret_cast<const float4*>(a.res + off);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    *reinterpret_cast<float4*>(a.out + off) =
        make_float4(o[0] * a.scale, o[1] * a.scale, o[2] * a.scale, o[3] * a.scale);
}

int launch_conv_cin4(const ConvArgs& a, hipStream_t s) {
    const int Q = a.Cout / 4;
    if (a.C1 != 4 || a.C2 != 0 || (a.Cout & 3) || Q > 256 || (256 % Q) != 0 ||
        (size_t)a.Cout * a.taps * 16 > 64 * 1024) {
        return launch_conv(a, s);      // generic path handles any shape
    }
    const int ppb = 256 / Q;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)((M + ppb - 1) / ppb);
    const size_t lds = (size_t)a.Cout * a.taps * 16;
    if (a.taps == 9)
        hipLaunchKernelGGL(conv_cin4_kernel<9>, dim3(grid), dim3(256), lds, s, a, Q);
    else
        hipLaunchKernelGGL(conv_cin4_kernel<1>, dim3(grid), dim3(256), lds, s, a, Q);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
