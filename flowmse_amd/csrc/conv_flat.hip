// Flat-tiled fp32 implicit-GEMM convolution kernels (v_mfma_f32_32x32x2_f32): 1x1 shortcuts / NIN, 3x3 on images the
// LDS-halo kernels do not cover, and the K slices of split-K launches.  Replaces (reference) ddpm_conv1x1 / ddpm_conv3x3
// (flowmse/backbones/ncsnpp_utils/layers.py:100-124) and NIN (layers.py:546-555).
//   conv_mfma_fast_kernel   C1 % 32 == 0 and C2 % 32 == 0: every K step (tap, 32-channel chunk) lies in ONE source tensor and
//                           is full; everything that varies per step is wave-uniform (buffer descriptor + soffset), per
//                           lane only a 9-bit tap-validity mask and one byte offset per gathered row; the block addresses
//                           its inputs through a window descriptor based at pixel m0 - W - 1
//   conv_mfma_kernel        generic fallback (any channel count multiple of 4, chunks straddling the concat)
// Common: block = 4 waves; LDS rows of 36 floats (conflict-free ds_read_b128); each lane reads 4 consecutive k of its row
// once and feeds 4 successive MFMAs; epilogue through LDS (conv_common.h).
#include "conv_common.h"

namespace flowse {

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
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int split = blockIdx.y;
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
    const int S_all = nchunks * taps;
    const int per = (S_all + a.ksplit - 1) / a.ksplit;
    const int s_begin = split * per;
    const int S = min(S_all, s_begin + per);       // this slice walks steps [s_begin, S)

    gload(s_begin);
    lstore(0);
    __syncthreads();

    for (int s = s_begin; s < S; ++s) {
        const int buf = (s - s_begin) & 1;
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

    conv_epilogue<WM, WN, TM, TN>(a, acc, smem, m0, n0, M, HW, split);
}

template <int WM, int WN, int TM, int TN, class OT = float>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_mfma_fast_kernel(ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    constexpr int A_LOADS = BM * 8 / NT, B_LOADS = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tiles must split evenly");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_ROW;

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int taps = a.taps;
    const int n_ntiles = (a.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int split = blockIdx.y;
    const int m0 = mt * BM, n0 = nt * BN;

    const int col4 = tid & 7, row0 = tid >> 3;
    // per gathered row: byte offset inside the window for each source, and which taps fall inside the image
    unsigned avo1[A_LOADS], avo2[A_LOADS], tapmask[A_LOADS];
#pragma unroll
    for (int q = 0; q < A_LOADS; ++q) {
        const int r = row0 + 32 * q;
        const int m = m0 + r;
        avo1[q] = (unsigned)(r * C1 + col4 * 4) * 4u;
        avo2[q] = (unsigned)(r * C2 + col4 * 4) * 4u;
        unsigned mask = 0;
        if (m < M) {
            const int rem = m % HW;
            const int y = rem / W, x = rem - y * W;
            if (taps == 9) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) mask |= 1u << t;
                }
            } else {
                mask = 1u;
            }
        }
        tapmask[q] = mask;
    }
    unsigned bvo[B_LOADS];
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
        const int n = n0 + row0 + 32 * q;
        bvo[q] = n < a.Cout ? (unsigned)(n * taps * Cin + col4 * 4) * 4u : OOB;
    }
    // window descriptors (wave-uniform): base = pixel (m0 - W - 1), BM + 2W + 2 pixels long
    const int64_t wbase = (int64_t)m0 - W - 1;
    const int wpix = BM + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + wbase * C1), 0, wpix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + wbase * C2 : a.in1), 0, C2 ? wpix * C2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.Cout * taps * Cin * 4, 0x00020000);

    // Operand staging: registers -> LDS, double-buffered in LDS AND two steps deep in registers.  The A rows of a 1x1
    // shortcut at 256 x 256 come straight from HBM (first touch of the block's pixels; 2-4 us loaded latency) while one K
    // step is only 64 MFMAs per wave = 1.7 us: requested ONE step ahead (round 2) they were waited for at every step;
    // requested TWO steps ahead they have a full step of slack.  The register ring index is a compile-time constant
    // (the loop is unrolled by two), so nothing is indexed dynamically.
    u32x4 ra[2][A_LOADS], rb[2][B_LOADS];

    auto gload = [&](int s, auto ring) {
        constexpr int R = decltype(ring)::value;
        const int chunk = s / taps, tap = s - chunk * taps;
        int shift = W + 1;                                    // window origin is pixel m0 - W - 1
        if (taps == 9) shift += (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff_a = (unsigned)(second ? shift * C2 + (c0 - C1) : shift * C1 + c0) * 4u;
        const unsigned soff_b = (unsigned)(tap * Cin + c0) * 4u;
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q) {
            const bool ok = (tapmask[q] >> tap) & 1u;
            const unsigned vo = ok ? (second ? avo2[q] : avo1[q]) : OOB;
            ra[R][q] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo, soff_a, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo, soff_a, 0);
        }
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q)
            rb[R][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo[q], soff_b, 0);
    };
    auto lstore = [&](int buf, auto ring) {
        constexpr int R = decltype(ring)::value;
        float* Ab = As + buf * BM * LDS_ROW;
        float* Bb = Bs + buf * BN * LDS_ROW;
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q)
            *reinterpret_cast<u32x4*>(Ab + (row0 + 32 * q) * LDS_ROW + col4 * 4) = ra[R][q];
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q)
            *reinterpret_cast<u32x4*>(Bb + (row0 + 32 * q) * LDS_ROW + col4 * 4) = rb[R][q];
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

    const int S_all = (Cin / KC) * taps;
    const int per = (S_all + a.ksplit - 1) / a.ksplit;
    const int s_begin = split * per;
    const int S = min(S_all, s_begin + per);

    auto compute = [&](int buf) {
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
        __builtin_amdgcn_sched_barrier(0);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;

    gload(s_begin, R0{});
    if (s_begin + 1 < S) gload(s_begin + 1, R1{});
    lstore(0, R0{});
    __syncthreads();

    for (int s = s_begin; s < S; s += 2) {
        // step s: LDS buffer 0; ring 0 is free (stored), ring 1 holds step s + 1
        if (s + 2 < S) gload(s + 2, R0{});
        compute(0);
        if (s + 1 < S) lstore(1, R1{});
        __syncthreads();
        if (s + 1 >= S) break;
        // step s + 1: LDS buffer 1; ring 1 is free, ring 0 holds step s + 2
        if (s + 3 < S) gload(s + 3, R1{});
        compute(1);
        if (s + 2 < S) lstore(0, R0{});
        __syncthreads();
    }
    conv_epilogue<WM, WN, TM, TN, OT>(a, acc, smem, m0, n0, M, HW, split);
}


template <int WM, int WN, int TM, int TN>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
    // staging buffers; the epilogue's C tile + stats scratch (NT*8 floats) must fit as well
    const size_t lds_stage = 2 * (BM + BN) * LDS_ROW * sizeof(float);
    const size_t lds_epi = ((size_t)BM * (BN + 4) + 64 * WM * WN * 8) * sizeof(float);
    const size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    if (const int rc = allow_lds<&conv_mfma_kernel<WM, WN, TM, TN>>(lds)) return rc;
    if (const int rc = allow_lds<&conv_mfma_fast_kernel<WM, WN, TM, TN, float>>(lds)) return rc;
    if (const int rc = allow_lds<&conv_mfma_fast_kernel<WM, WN, TM, TN, bf16_t>>(lds)) return rc;
    if (const int rc = allow_lds<&conv_mfma_fast_kernel<WM, WN, TM, TN, f16_t>>(lds)) return rc;
    // fast path: every K step is a full 32-channel chunk of one source; window / weight offsets fit 31 bits
    const bool fast = (a.C1 % KC) == 0 && (a.C2 % KC) == 0 &&
                      (int64_t)(BM + 2 * a.W + 2) * (a.C1 > a.C2 ? a.C1 : a.C2) * 4 < (1LL << 31) &&
                      (int64_t)a.Cout * a.taps * (a.C1 + a.C2) * 4 < (1LL << 31) && !conv_force_generic();
    if (a.in_dt != DT_F32 || (a.out_dt != DT_F32 && !fast)) {
        set_error("conv: the fp32 flat kernels take fp32 inputs (16-bit outputs only on the 32-channel-aligned path)");
        return ERR_ARG;
    }
    if (fast) {
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL((conv_mfma_fast_kernel<WM, WN, TM, TN, OT>), dim3(grid, a.ksplit),
                                                          dim3(64 * WM * WN), lds, s, a));
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN>), dim3(grid, a.ksplit), dim3(64 * WM * WN), lds, s, a);
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_flat_fp32(const ConvArgs& a, hipStream_t s) {
    return a.Cout <= 32 ? launch_cfg<4, 1, 1, 1>(a, s)
         : a.Cout <= 64 ? launch_cfg<2, 2, 2, 1>(a, s)
         : conv_small_m((int64_t)a.B * a.H * a.W, a.Cout) ? launch_cfg<1, 4, 1, 1>(a, s) : launch_cfg<2, 2, 2, 2>(a, s);
}

}  // namespace flowse
