// Replaces (reference): ddpm_conv3x3 / ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-124) and NIN (:546-555) on
// the SMALL images of the network (ncsnpp.py:289-330, 335-385 at the 16 x 16 ... 4 x 4 levels; 32 x 32 for a single
// utterance): at most 2048 pixels in the whole batch.
#include "conv_common.h"

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// Small-M implicit GEMM with the K split INSIDE the block.
//
// With M = B H W <= 2048 pixels a 128 x 128 tiling has at most 16 tiles, so rounds 1-4 sliced K over extra blocks: every
// slice wrote a raw fp32 partial tile ([ksplit][M][Cout] slab: 33 MB for a 2 MB result at 16 x 16, batch 8) and a second
// launch read the slabs back, summed them and ran the epilogue -- 40 us per convolution against an MFMA floor of 8-15.
// Here a block owns a 32-pixel x (32 or 64)-channel output tile (256 blocks at 2048 pixels x 256 channels: one per CU)
// and its EIGHT waves each take every eighth K step (tap, 32-channel chunk): a wave requests its own operands straight
// into MFMA fragment registers -- A: 16 bytes per lane from the lane's pixel through a window descriptor (conv zero
// padding = an out-of-range offset, hardware returns 0), B: weights kept a second time in fragment order (one contiguous
// 1 KB line per wave-level request) -- two steps ahead of the MFMAs that consume them; no LDS, no barrier in the K loop.
// The eight partial accumulator tiles meet once in LDS ([wave][32][BN + 4]), are summed in a fixed order (bit-reproducible),
// and the same pass adds bias / per-sample bias / residual, scales, stores 16-byte quads and leaves the GroupNorm
// partial statistics of what it stored (per block of min(32, H W) pixels and channel: mean, M2).  No slab, no second launch.
constexpr int SM_BM = 32;

// f(integral_constant<int, 0>) ... f(integral_constant<int, D - 1>): register-ring slots are compile-time indices
template <class F, int... I>
__device__ __forceinline__ void sm_unroll(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int NT2>
__global__ __launch_bounds__(512, 2) void conv_smallm_kernel(ConvArgs a) {
    constexpr int BN = 32 * NT2, CROW = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [8][32][CROW] (+ statistics scratch behind it)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int taps = a.taps;
    const int n_ntiles = a.Cout / BN;
    const int mt = blockIdx.x / n_ntiles, nt = blockIdx.x - mt * n_ntiles;
    const int m0 = mt * SM_BM, n0 = nt * BN;

    // this lane's pixel: which taps fall inside the image (bit t), its byte offset inside the window of each source
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
    const unsigned avo1 = (unsigned)(li * C1 + kh * 4) * 4u, avo2 = (unsigned)(li * C2 + kh * 4) * 4u;
    // window descriptors (wave-uniform): base = pixel (m0 - W - 1), 32 + 2W + 2 pixels long
    const int64_t wbase = (int64_t)m0 - W - 1;
    const int wpix = SM_BM + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + wbase * C1), 0, wpix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + wbase * C2 : a.in1), 0, C2 ? wpix * C2 * 4 : 0, 0x00020000);
    const int nchunks = Cin / KC;
    // weights in fragment order: [Cout/32][tap][chunk][k-block 4][lane 64][4 floats]
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wsm), 0, a.Cout * taps * Cin * 4, 0x00020000);
    const unsigned bvo = (unsigned)lane * 16u;

    const int S_all = nchunks * taps;
    // Register ring, D steps deep: a step is only 16 NT2 MFMAs (0.5-0.9 us) while its operands come from L2 -- or, for the
    // weights of a 4 x 4 image that few blocks share, straight from HBM -- so D - 1 steps of requests stay in flight.
    constexpr int D = NT2 == 1 ? 4 : 2;
    u32x4 ra[D][4], rb[D][NT2][4];
    auto gload = [&](int s, auto ring) {
        constexpr int R = decltype(ring)::value;
        const bool live = s < S_all;                          // steps past the end: clamped addresses, A reads as zero
        s = live ? s : S_all - 1;
        const int chunk = s / taps, tap = s - chunk * taps;
        int shift = W + 1;                                    // window origin is pixel m0 - W - 1
        if (taps == 9) shift += (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff_a = (unsigned)(second ? shift * C2 + (c0 - C1) : shift * C1 + c0) * 4u;
        const bool ok = live && ((tapmask >> tap) & 1u);
        const unsigned vo = ok ? (second ? avo2 : avo1) : OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[R][j] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo + j * 32, soff_a, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo + j * 32, soff_a, 0);
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            const unsigned soff_b = (unsigned)((((n0 >> 5) + t) * taps + tap) * nchunks + chunk) * 4096u;
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[R][t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo + j * 1024, soff_b, 0);
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
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R][j].x), __uint_as_float(rb[R][t][j].x), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R][j].y), __uint_as_float(rb[R][t][j].y), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R][j].z), __uint_as_float(rb[R][t][j].z), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R][j].w), __uint_as_float(rb[R][t][j].w), acc[t], 0, 0, 0);
            }
    };
    // wave w walks steps w, w + 8, ...: n per wave (the last one masked for the waves that have one fewer), ring slot of
    // its i-th step = i % D.  No request sits under a runtime branch (hipcc drains vmcnt at every join: a conditional
    // request serialised the whole ring -- 1 us per step): a uniform loop of D-step groups, requests D - 1 steps ahead on
    // clamped addresses, and a tail of at most D - 1 steps whose operands are already in flight.
    const int nstep = (S_all + 7) >> 3;
    {
        auto pro = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) gload(wave + 8 * d, dc);
        };
        sm_unroll(pro, std::make_integer_sequence<int, D>{});
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
        sm_unroll(step, std::make_integer_sequence<int, D>{});
    }
    {
        const int rem = nstep - i;
        auto tail = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) {
                if (d < rem) compute(dc);
            }
        };
        sm_unroll(tail, std::make_integer_sequence<int, D>{});
    }

    // ---- the eight partial tiles meet in LDS: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 kh
    float* Cs = smem;
    {
        float* Cw = Cs + wave * (SM_BM * CROW);
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[((r & 3) + 8 * (r >> 2) + 4 * kh) * CROW + t * 32 + li] = acc[t][r];
    }
    __syncthreads();
    // thread = (tile row, channel quad): BN / 4 quads x 32 rows = 256 (BN = 32) or 512 (BN = 64) items
    constexpr int C4 = BN / 4;
    const int row = tid / C4, cq = tid - row * C4;
    const bool act = row < SM_BM;
    const int mo = m0 + row;
    const int n = n0 + cq * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(Cs + (w * SM_BM + row) * CROW + cq * 4);
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
                const float4 t = *reinterpret_cast<const float4*>(a.res + (int64_t)mo * a.Cout + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            *reinterpret_cast<float4*>(a.out + (int64_t)mo * a.Cout + n) = v;
        }
    }
    if (!a.stats) return;
    // GroupNorm partial statistics of what was stored: blocks of PB = min(32, H W) consecutive pixels of one sample
    // (launch guarantees: 32 % PB == 0, H W % PB == 0, M % PB == 0).  The finished tile goes back to LDS slice 0 and
    // one thread per (block, channel) takes mean and M2 in two passes over its PB values (exact, order fixed).
    __syncthreads();                                     // everyone has read the eight slices
    if (act) *reinterpret_cast<float4*>(Cs + row * CROW + cq * 4) = v;
    __syncthreads();
    const int PB = HW < SM_BM ? HW : SM_BM;
    const int groups = SM_BM / PB;
    for (int idx = tid; idx < groups * BN; idx += 512) {      // (H W < 4: more (block, channel) pairs than threads)
        const int g = idx / BN, c = idx - g * BN;
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

// ---------------------------------------------------------------------------------------------------
// The same scheme on 16 x 16 output tiles (v_mfma_f32_16x16x4_f32) for at most 256 pixels: a 32 x 32 tile of a 4 x 4 image
// (batch 8: 128 pixels -> 32 blocks; one utterance: 16 pixels -> 8 blocks, half of every tile padding) leaves 7/8 of the
// chip idle while each block grinds through the whole K on one CU (8 us of MFMA time whatever M is).  16 x 16 tiles give
// four times the blocks at a quarter of the MFMA work each.  Fragment layout of a 32-channel chunk: lane (row or column
// l & 15, k group l >> 4) holds channels 8 (l >> 4) .. + 7 as two float4 (MFMA e of the step uses element e of both
// operands); weights are kept a third time in that order ([Cout/16][tap][chunk][half][lane][4]).
__global__ __launch_bounds__(512, 2) void conv_smallm16_kernel(ConvArgs a) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int BM = 16, BN = 16, CROW = BN + 4;
    __shared__ __attribute__((aligned(16))) float Cs[8 * BM * CROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
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
    const unsigned avo1 = (unsigned)(li * C1 + kq * 8) * 4u, avo2 = (unsigned)(li * C2 + kq * 8) * 4u;
    const int64_t wbase = (int64_t)m0 - W - 1;
    const int wpix = BM + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + wbase * C1), 0, wpix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + wbase * C2 : a.in1), 0, C2 ? wpix * C2 * 4 : 0, 0x00020000);
    const int nchunks = Cin / KC;
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wsm16), 0, a.Cout * taps * Cin * 4, 0x00020000);
    const unsigned bvo = (unsigned)lane * 16u;
    const int S_all = nchunks * taps;
    // With 16-256 pixels there are only 16-256 blocks and each conv's weights come from HBM once: a wave keeps SEVEN of
    // its 9-18 steps in flight (16 registers per step) -- two latency rounds instead of five.
    constexpr int D = 8;
    u32x4 ra[D][2], rb[D][2];
    auto gload = [&](int s, auto ring) {
        constexpr int R = decltype(ring)::value;
        const bool live = s < S_all;
        s = live ? s : S_all - 1;
        const int chunk = s / taps, tap = s - chunk * taps;
        int shift = W + 1;
        if (taps == 9) shift += (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff_a = (unsigned)(second ? shift * C2 + (c0 - C1) : shift * C1 + c0) * 4u;
        const bool ok = live && ((tapmask >> tap) & 1u);
        const unsigned vo = ok ? (second ? avo2 : avo1) : OOB;
        const unsigned soff_b = (unsigned)(((n0 >> 4) * taps + tap) * nchunks + chunk) * 2048u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ra[R][h] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo + h * 16, soff_a, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo + h * 16, soff_a, 0);
            rb[R][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo + h * 1024, soff_b, 0);
        }
    };
    // two accumulators in turn (a dependent 16x16x4 MFMA waits 40 cycles, an independent one issues after 32)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    auto compute = [&](auto ring) {
        constexpr int R = decltype(ring)::value;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][0].x), __uint_as_float(rb[R][0].x), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][1].x), __uint_as_float(rb[R][1].x), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][0].y), __uint_as_float(rb[R][0].y), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][1].y), __uint_as_float(rb[R][1].y), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][0].z), __uint_as_float(rb[R][0].z), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][1].z), __uint_as_float(rb[R][1].z), acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][0].w), __uint_as_float(rb[R][0].w), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ra[R][1].w), __uint_as_float(rb[R][1].w), acc1, 0, 0, 0);
    };
    // (loop structure: see conv_smallm_kernel)
    const int nstep = (S_all + 7) >> 3;
    {
        auto pro = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) gload(wave + 8 * d, dc);
        };
        sm_unroll(pro, std::make_integer_sequence<int, D>{});
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
        sm_unroll(step, std::make_integer_sequence<int, D>{});
    }
    {
        const int rem = nstep - i;
        auto tail = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) {
                if (d < rem) compute(dc);
            }
        };
        sm_unroll(tail, std::make_integer_sequence<int, D>{});
    }

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r
    {
        float* Cw = Cs + wave * (BM * CROW);
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(4 * kq + r) * CROW + li] = acc0[r] + acc1[r];
    }
    __syncthreads();
    const int row = tid >> 2, cq = tid & 3;              // 16 rows x 4 channel quads = 64 threads
    const bool act = tid < 64;
    const int mo = m0 + row;
    const int n = n0 + cq * 4;
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
                const float4 t = *reinterpret_cast<const float4*>(a.res + (int64_t)mo * a.Cout + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            *reinterpret_cast<float4*>(a.out + (int64_t)mo * a.Cout + n) = v;
        }
    }
    if (!a.stats) return;
    // statistics block = the tile's 16 pixels (one sample: H W % 16 == 0): mean and M2 per channel, two passes
    __syncthreads();
    if (act) *reinterpret_cast<float4*>(Cs + row * CROW + cq * 4) = v;
    __syncthreads();
    if (tid < BN && m0 < M) {
        float sum = 0.f;
        for (int r = 0; r < BM; ++r) sum += Cs[r * CROW + tid];
        const float mean = sum * (1.f / BM);
        float m2 = 0.f;
        for (int r = 0; r < BM; ++r) {
            const float d = Cs[r * CROW + tid] - mean;
            m2 = fmaf(d, d, m2);
        }
        const int bs = m0 / HW, blk = (m0 - bs * HW) / BM;
        float* dst = a.stats + (((int64_t)bs * a.stats_nblk + blk) * a.Cout + n0 + tid) * 2;
        dst[0] = mean;
        dst[1] = m2;
    }
}

// [Cout][taps][Cin] -> [Cout/16][tap][Cin/32][half][lane][4]: lane = (n & 15) + 16 (k group), k group = (ci & 31) >> 3
__global__ __launch_bounds__(256) void smallm16_weights_kernel(const float* __restrict__ w, int Cout, int taps, int Cin,
                                                               float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (n, tap, ci)
    if (idx >= (int64_t)Cout * taps * Cin) return;
    const int ci = (int)(idx % Cin);
    const int tap = (int)((idx / Cin) % taps);
    const int64_t n = idx / ((int64_t)taps * Cin);
    const int nchunks = Cin >> 5;
    const int chunk = ci >> 5, kq = (ci >> 3) & 3, h = (ci >> 2) & 1, e = ci & 3;
    const int lane = kq * 16 + (int)(n & 15);
    out[(((((n >> 4) * taps + tap) * nchunks + chunk) * 2 + h) * 64 + lane) * 4 + e] = w[idx];
}

// [Cout][taps][Cin] -> fragment order [Cout/32][tap][Cin/32][k-block j][lane][4]
__global__ __launch_bounds__(256) void smallm_weights_kernel(const float* __restrict__ w, int Cout, int taps, int Cin,
                                                             float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (n, tap, ci)
    if (idx >= (int64_t)Cout * taps * Cin) return;
    const int ci = (int)(idx % Cin);
    const int tap = (int)((idx / Cin) % taps);
    const int64_t n = idx / ((int64_t)taps * Cin);
    const int nchunks = Cin >> 5;
    const int chunk = ci >> 5, j = (ci >> 3) & 3, kh = (ci >> 2) & 1, e = ci & 3;
    const int lane = kh * 32 + (int)(n & 31);
    out[(((((n >> 5) * taps + tap) * nchunks + chunk) * 4 + j) * 64 + lane) * 4 + e] = w[idx];
}

int launch_smallm_weights(const float* w_packed, int Cout, int taps, int Cin, float* out, hipStream_t s, bool tile16) {
    if ((Cout % 32) != 0 || (Cin % 32) != 0) {
        set_error("smallm_weights: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
        return ERR_SHAPE;
    }
    const int64_t n = (int64_t)Cout * taps * Cin;
    if (tile16)
        hipLaunchKernelGGL(smallm16_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_packed, Cout, taps, Cin, out);
    else
        hipLaunchKernelGGL(smallm_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_packed, Cout, taps, Cin, out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// at most 256 pixels (and whole 16-pixel blocks per sample): the 16 x 16-tile kernel (at 512 pixels its 512 blocks pull
// 151 MB of operands through L2 for a 0.6 GFLOP product: no faster than the 32 x 32 tiles' 128 blocks)
bool conv_smallm_tile16(int B, int H, int W) { return (int64_t)B * H * W <= 256 && ((H * W) % 16) == 0; }

// statistics blocks per sample of this kernel's fused statistics (0: the tile geometry does not allow them)
int conv_smallm_stats_blocks(int B, int H, int W) {
    if (conv_smallm_tile16(B, H, W)) return H * W / 16;
    const int HW = H * W, PB = HW < SM_BM ? HW : SM_BM;
    if (PB <= 0 || (SM_BM % PB) != 0 || (HW % PB) != 0) return 0;
    return HW / PB;
}

int launch_smallm(const ConvArgs& a, hipStream_t s) {
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int mtiles = (int)((M + SM_BM - 1) / SM_BM);
    // 64-channel tiles when that still gives every CU a block, else 32-channel tiles (twice the blocks)
    const bool wide = (a.Cout % 64) == 0 && (int64_t)mtiles * (a.Cout / 64) >= 256;
    if (a.stats && a.stats_nblk != conv_smallm_stats_blocks(a.B, a.H, a.W)) {
        set_error("conv_smallm: inconsistent fused-stats geometry (stats_nblk=%d)", a.stats_nblk);
        return ERR_ARG;
    }
    if (conv_smallm_tile16(a.B, a.H, a.W)) {
        if (!a.wsm16) {
            set_error("conv_smallm: the 16 x 16-tile form needs ConvArgs::wsm16");
            return ERR_ARG;
        }
        hipLaunchKernelGGL(conv_smallm16_kernel, dim3((unsigned)(((M + 15) / 16) * (a.Cout / 16))), dim3(512), 0, s, a);
    } else if (wide) {
        const size_t lds = (size_t)8 * SM_BM * (64 + 4) * sizeof(float);
        if (const int rc = allow_lds<&conv_smallm_kernel<2>>(lds)) return rc;
        hipLaunchKernelGGL((conv_smallm_kernel<2>), dim3(mtiles * (a.Cout / 64)), dim3(512), lds, s, a);
    } else {
        const size_t lds = (size_t)8 * SM_BM * (32 + 4) * sizeof(float);
        if (const int rc = allow_lds<&conv_smallm_kernel<1>>(lds)) return rc;
        hipLaunchKernelGGL((conv_smallm_kernel<1>), dim3(mtiles * (a.Cout / 32)), dim3(512), lds, s, a);
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
