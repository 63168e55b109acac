// Replaces (reference): ddpm_conv3x3 / ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-124) and NIN (:546-555) on
// the small images of the network in the 16-bit STORAGE modes (BASELINE configs 2 / 4: bf16 / fp16 activations): at most
// 2048 pixels in the whole batch (the 16 x 16 ... 4 x 4 levels at batch 8).
#include "conv_common.h"

namespace flowse {

// The 16-bit twin of conv_smallm_kernel (conv_smallm.hip): a block owns a 32-pixel x (32 or 64)-channel output tile, its
// eight waves each take every eighth K step (tap, 32-channel chunk) and request their operands straight into the fragment
// registers of v_mfma_f32_32x32x16_{bf16,f16} -- A: two 16-byte pieces per lane and step (8 channels each) through a window
// descriptor, B: the fragment-order weights the producer / consumer kernel already keeps ([Cout/32][chunk][tap][half][lane][8],
// launch_pc16_weights; built for the 1x1 convs as well) -- through a branch-free ring six or eight steps deep.  A step is only
// two or four 32-cycle MFMAs: the kernel is a latency / L2-bandwidth exercise, the ring is what matters.  The eight fp32
// partial tiles meet once in LDS, are summed in a fixed order, and the same pass adds bias / per-sample bias / residual,
// rounds ONCE to the storage type, stores, and leaves the GroupNorm partial statistics of the ROUNDED values (what the
// consumer reads).  Rounds 2-4 ran these shapes through conv_flat16_kernel with fp32 slabs + a reduction launch.
template <class F, int... I>
__device__ __forceinline__ void sm16_unroll(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// OT: storage type of res / out -- the operands' 16-bit type, or float (the few convs that leave the 16-bit domain)
template <int NT2, bool F16, class OT>
__global__ __launch_bounds__(512, 2) void conv_smallm16b_kernel(ConvArgs a) {
    using T16 = typename std::conditional<F16, f16_t, bf16_t>::type;
    constexpr int BM = 32, BN = 32 * NT2, CROW = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [8][32][CROW]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int taps = a.taps;
    const int n_ntiles = a.Cout / BN;
    const int mt = blockIdx.x / n_ntiles, nt = blockIdx.x - mt * n_ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int m = m0 + li;
    unsigned tapmask = 0;
    if (m < M) {
        const int rem = m % HW;
        const int y = rem / W, x = rem - y * W;
        if (taps == 9) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) tapmask |= 1u << t;
            }
        } else {
            tapmask = 1u;
        }
    }
    const unsigned avo1 = (unsigned)(li * C1 + kh * 8) * 2u, avo2 = (unsigned)(li * C2 + kh * 8) * 2u;
    const int64_t wbase = (int64_t)m0 - W - 1;
    const int wpix = BM + 2 * W + 2;
    const T16* in1p = reinterpret_cast<const T16*>(a.in1);
    const T16* in2p = reinterpret_cast<const T16*>(a.in2);
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T16*>(in1p + wbase * C1), 0, wpix * C1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T16*>(C2 ? in2p + wbase * C2 : in1p), 0, C2 ? wpix * C2 * 2 : 0, 0x00020000);
    const int nchunks = Cin / KC;
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wfrag), 0, a.Cout * taps * Cin * 2, 0x00020000);
    const unsigned bvo = (unsigned)lane * 16u;
    const int S_all = nchunks * taps;
    constexpr int D = NT2 == 1 ? 8 : 6;
    u32x4 ra[D][2], rb[D][NT2][2];
    auto gload = [&](int s, auto ring) {
        constexpr int R = decltype(ring)::value;
        const bool live = s < S_all;                          // steps past the end: clamped addresses, A reads as zero
        s = live ? s : S_all - 1;
        const int chunk = s / taps, tap = s - chunk * taps;
        int shift = W + 1;                                    // window origin is pixel m0 - W - 1
        if (taps == 9) shift += (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff_a = (unsigned)(second ? shift * C2 + (c0 - C1) : shift * C1 + c0) * 2u;
        const bool ok = live && ((tapmask >> tap) & 1u);
        const unsigned vo = ok ? (second ? avo2 : avo1) : OOB;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
            ra[R][mh] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo + mh * 32, soff_a, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo + mh * 32, soff_a, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            const unsigned soff_b = (unsigned)((((n0 >> 5) + t) * nchunks + chunk) * taps + tap) * 2048u;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) rb[R][t][mh] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo + mh * 1024, soff_b, 0);
        }
    };
    f32x16 acc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto compute = [&](auto ring) {
        constexpr int R = decltype(ring)::value;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                if (F16)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[R][mh]),
                                                                    __builtin_bit_cast(f16x8, rb[R][t][mh]), acc[t], 0, 0, 0);
                else
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[R][mh]),
                                                                     __builtin_bit_cast(bf16x8, rb[R][t][mh]), acc[t], 0, 0, 0);
            }
    };
    // (loop structure: see conv_smallm_kernel -- no request under a runtime branch)
    const int nstep = (S_all + 7) >> 3;
    {
        auto pro = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) gload(wave + 8 * d, dc);
        };
        sm16_unroll(pro, std::make_integer_sequence<int, D>{});
    }
    int i = 0;
    for (; i + D <= nstep; i += D) {
        auto step = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            gload(wave + 8 * (i + d + D - 1), std::integral_constant<int, (d + D - 1) % D>{});
            __builtin_amdgcn_sched_barrier(0);            // (hipcc otherwise sinks the requests down to their first use)
            compute(dc);
            __builtin_amdgcn_sched_barrier(0);
        };
        sm16_unroll(step, std::make_integer_sequence<int, D>{});
    }
    {
        const int rem = nstep - i;
        auto tail = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) {
                if (d < rem) compute(dc);
            }
        };
        sm16_unroll(tail, std::make_integer_sequence<int, D>{});
    }

    float* Cs = smem;
    {
        float* Cw = Cs + wave * (BM * CROW);
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * kh) * CROW + t * 32 + li] = acc[t][r];
    }
    __syncthreads();
    constexpr int C4 = BN / 4;
    const int row = tid / C4, cq = tid - row * C4;
    const bool act = row < BM;
    const int mo = m0 + row;
    const int n = n0 + cq * 4;
    OT* outp = reinterpret_cast<OT*>(a.out);
    const OT* resp = reinterpret_cast<const OT*>(a.res);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(Cs + (w * BM + row) * CROW + cq * 4);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (a.bias) {
            const float4 t = *reinterpret_cast<const float4*>(a.bias + n);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (mo < M) {
            if (a.bias2) {
                const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)(mo / HW) * a.bias2_stride + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (a.res) {
                const float4 t = St<OT>::ld4(resp + (int64_t)mo * a.Cout + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            St<OT>::st4(outp + (int64_t)mo * a.Cout + n, v);
            v = St<OT>::rnd4(v);                              // statistics of what the consumer will read
        }
    }
    if (!a.stats) return;
    __syncthreads();
    if (act) *reinterpret_cast<float4*>(Cs + row * CROW + cq * 4) = v;
    __syncthreads();
    const int PB = HW < BM ? HW : BM;
    const int groups = BM / PB;
    if (tid < groups * BN) {
        const int g = tid / BN, c = tid - g * BN;
        const int mg = m0 + g * PB;
        if (mg < M) {
            float sum = 0.f;
            for (int r = 0; r < PB; ++r) sum += Cs[(g * PB + r) * CROW + c];
            const float mean = sum / (float)PB;
            float m2 = 0.f;
            for (int r = 0; r < PB; ++r) {
                const float d = Cs[(g * PB + r) * CROW + c] - mean;
                m2 = fmaf(d, d, m2);
            }
            const int bs = mg / HW, blk = (mg - bs * HW) / PB;
            float* dst = a.stats + (((int64_t)bs * a.stats_nblk + blk) * a.Cout + n0 + c) * 2;
            dst[0] = mean;
            dst[1] = m2;
        }
    }
}

// shapes the kernel takes in the 16-bit storage modes: at most 2048 pixels, 32-aligned channel counts, and not one of the
// launches the LDS-halo kernels cover anyway (those need >= 128 blocks of 128 pixels)
static const bool g_no_smallm16 = getenv("FLOWSE_NO_SMALLM") != nullptr || getenv("FLOWSE_FORCE_GENERIC_CONV") != nullptr;
bool conv16_smallm_ok(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (g_no_smallm16 || (taps != 1 && taps != 9) || (C1 % KC) || (C2 % KC) || (Cout % 32) || C1 <= 0) return false;
    const int64_t M = (int64_t)B * H * W;
    const int HW = H * W, PB = HW < 32 ? HW : 32;
    if (M > 2048 || M < 1 || PB <= 0 || (32 % PB) != 0 || (HW % PB) != 0) return false;
    if (conv16_uses_halo(B, H, W, C1, C2, Cout, taps)) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return (int64_t)(32 + 2 * W + 2) * cmax * 2 < (1LL << 31) && (int64_t)Cout * taps * (C1 + C2) * 2 < (1LL << 31);
}

// statistics blocks per sample of the 16-bit kernel's fused statistics: blocks of min(32, H W) pixels
int conv16_smallm_stats_blocks(int B, int H, int W) {
    const int HW = H * W, PB = HW < 32 ? HW : 32;
    (void)B;
    return (PB > 0 && (32 % PB) == 0 && (HW % PB) == 0) ? HW / PB : 0;
}

int launch_smallm16b(const ConvArgs& a, hipStream_t s) {
    if (!a.wfrag || a.in_dt == DT_F32 || (a.out_dt != a.in_dt && a.out_dt != DT_F32) || (a.wq_f16 ? DT_F16 : DT_BF16) != a.in_dt || a.gn.mean ||
        !conv16_smallm_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        set_error("conv_smallm16b: unsupported configuration (C1=%d C2=%d in_dt=%d out_dt=%d)", a.C1, a.C2, a.in_dt, a.out_dt);
        return ERR_ARG;
    }
    if (a.stats && a.stats_nblk != conv16_smallm_stats_blocks(a.B, a.H, a.W)) {
        set_error("conv_smallm16b: inconsistent fused-stats geometry (stats_nblk=%d)", a.stats_nblk);
        return ERR_ARG;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int mtiles = (int)((M + 31) / 32);
    const bool wide = (a.Cout % 64) == 0 && (int64_t)mtiles * (a.Cout / 64) >= 256;
#define FLOWSE_LSM16(NT2, F16, OT)                                                                            \
    {                                                                                                        \
        const size_t lds = (size_t)8 * 32 * (32 * NT2 + 4) * sizeof(float);                                  \
        if (const int rc = allow_lds<&conv_smallm16b_kernel<NT2, F16, OT>>(lds)) return rc;                  \
        hipLaunchKernelGGL((conv_smallm16b_kernel<NT2, F16, OT>), dim3(mtiles * (a.Cout / (32 * NT2))), dim3(512), lds, s, a); \
    }
    const bool of32 = a.out_dt == DT_F32;
    if (a.wq_f16) {
        if (wide) { if (of32) FLOWSE_LSM16(2, true, float) else FLOWSE_LSM16(2, true, f16_t) }
        else { if (of32) FLOWSE_LSM16(1, true, float) else FLOWSE_LSM16(1, true, f16_t) }
    } else {
        if (wide) { if (of32) FLOWSE_LSM16(2, false, float) else FLOWSE_LSM16(2, false, bf16_t) }
        else { if (of32) FLOWSE_LSM16(1, false, float) else FLOWSE_LSM16(1, false, bf16_t) }
    }
#undef FLOWSE_LSM16
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
