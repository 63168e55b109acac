// Replaces (reference): ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-105) = the shortcut Conv_2 of
// ResnetBlockBigGANpp (layerspp.py:268-270) and NIN (layers.py:546-555) on the LARGE images of the fp32 mode.
#include "conv_common.h"

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// Streaming 1x1 convolution: out[m][n] = sum_c A[m][c] W[n][c], A = the NHWC activation tensor itself ([pixels][Cin],
// fully contiguous; the channel concat of ncsnpp.py:337 = two K ranges).
//
// Rounds 1-4 ran these through the flat 3x3 machinery (128 x 128 tiles, both operands through LDS, two block barriers per
// 32-channel step, C tile through LDS): 0.44-0.65 of the fp32 matrix peak for four rounds, 10.8 % of the fp32 step.  Round 3's
// probes said the time is not in the K loop's mechanics -- so this kernel has none: EVERY WAVE IS ITS OWN GEMM.  A wave owns
// 32 pixels x 128 output channels (four 32 x 32 accumulator tiles) over the whole K; per 8-channel k-block it requests ONE
// A fragment (16 bytes per lane straight from the activation tensor, through a per-wave buffer descriptor: rows past M read
// as zero) and FOUR B fragments (the fragment-order weight copy the small-image kernel uses, 1 KB per wave-level request,
// L2-resident: the whole matrix is 64-256 KB) and issues 16 MFMAs.  No LDS, no barrier, no VALU in the loop; requests run
// three k-blocks ahead through register rings with compile-time slots.  Output: straight from the accumulator layout (a lane
// holds one channel of 16 pixels per tile: 64 dword stores per wave, 128 contiguous bytes per half-wave; the output is 1/3 of
// the traffic and far from store-issue bound here), bias / per-sample bias / residual added on the way; GroupNorm partial
// statistics per block of 256 pixels (eight waves meet once in LDS).
constexpr int C1_PX = 32;            // pixels per wave
constexpr int C1_WAVES = 8;          // waves per block (256 pixels: the statistics block).  (Four-wave blocks, three per CU so that
                                     // output stages overlap other blocks' MFMAs, measured 6 % slower: 12 waves' requests per CU)

template <class F, int... I>
__device__ __forceinline__ void c1_unroll(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

__global__ __launch_bounds__(512, 2) void conv1x1_stream_kernel(ConvArgs a) {
    __shared__ float red[C1_WAVES * 128 * 2];            // per wave and channel: mean, M2 of its 32 pixels
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = a.Cout >> 7;                    // 128-channel blocks
    const int mb = blockIdx.x / n_ntiles, nt = blockIdx.x - mb * n_ntiles;
    const int m0 = (mb * C1_WAVES + wave) * C1_PX;       // this wave's first pixel
    const int n0 = nt * 128;
    const int nchunks = Cin / KC;
    // A: this wave's 32 pixels of each source; rows past M lie outside the descriptor and read as zero
    const int rows = M - m0 < C1_PX ? (M - m0 < 0 ? 0 : M - m0) : C1_PX;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in1 + (int64_t)(m0 < M ? m0 : 0) * C1), 0, rows * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + (int64_t)(m0 < M ? m0 : 0) * C2 : a.in1), 0, C2 ? rows * C2 * 4 : 0, 0x00020000);
    const unsigned avo1 = (unsigned)(li * C1 + kh * 4) * 4u, avo2 = (unsigned)(li * C2 + kh * 4) * 4u;
    // B: fragment order [Cout/32][tap = 1][chunk][k-block 4][lane 64][4 floats] (launch_smallm_weights)
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wsm), 0, a.Cout * Cin * 4, 0x00020000);
    const unsigned bvo = (unsigned)lane * 16u;
    const unsigned bt = (unsigned)nchunks * 4096u;       // bytes between two 32-channel blocks
    const unsigned b0 = (unsigned)(n0 >> 5) * bt;

    const int S = nchunks * 4;                           // k-blocks of 8 channels
    constexpr int D = 4;                                 // ring depth (k-blocks in flight; 8 measured 5 % slower: 232 registers)
    u32x4 ra[D], rb[D][4];
    auto gload = [&](int s, auto ring) {
        constexpr int R = decltype(ring)::value;
        s = s < S ? s : S - 1;                            // past the end: a harmless repeat (never consumed)
        const int chunk = s >> 2, j = s & 3;
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff_a = (unsigned)((second ? c0 - C1 : c0) + j * 8) * 4u;
        ra[R] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, avo2, soff_a, 0)
                       : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, avo1, soff_a, 0);
        const unsigned soff_b = b0 + (unsigned)chunk * 4096u + (unsigned)j * 1024u;
#pragma unroll
        for (int t = 0; t < 4; ++t) rb[R][t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo, soff_b + (unsigned)t * bt, 0);
    };
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto compute = [&](auto ring) {
        constexpr int R = decltype(ring)::value;
        // k-steps outer, channel tiles inner: four accumulators in turn (never two MFMAs in a row on one)
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].x), __uint_as_float(rb[R][0].x), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].x), __uint_as_float(rb[R][1].x), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].x), __uint_as_float(rb[R][2].x), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].x), __uint_as_float(rb[R][3].x), acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].y), __uint_as_float(rb[R][0].y), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].y), __uint_as_float(rb[R][1].y), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].y), __uint_as_float(rb[R][2].y), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].y), __uint_as_float(rb[R][3].y), acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].z), __uint_as_float(rb[R][0].z), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].z), __uint_as_float(rb[R][1].z), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].z), __uint_as_float(rb[R][2].z), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].z), __uint_as_float(rb[R][3].z), acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].w), __uint_as_float(rb[R][0].w), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].w), __uint_as_float(rb[R][1].w), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].w), __uint_as_float(rb[R][2].w), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ra[R].w), __uint_as_float(rb[R][3].w), acc[3], 0, 0, 0);
    };
    // a uniform loop of D-step groups, requests D - 1 steps ahead, none under a branch; a tail of at most D - 1 steps whose
    // operands are already in flight
    {
        auto pro = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) gload(d, dc);
        };
        c1_unroll(pro, std::make_integer_sequence<int, D>{});
    }
    int i = 0;
    for (; i + D <= S; i += D) {
        auto step = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            gload(i + d + D - 1, std::integral_constant<int, (d + D - 1) % D>{});
            __builtin_amdgcn_sched_barrier(0);            // (hipcc otherwise sinks the requests down to their first use)
            compute(dc);
            __builtin_amdgcn_sched_barrier(0);
        };
        c1_unroll(step, std::make_integer_sequence<int, D>{});
    }
    {
        const int rem = S - i;
        auto tail = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < D - 1) {
                if (d < rem) compute(dc);
            }
        };
        c1_unroll(tail, std::make_integer_sequence<int, D>{});
    }

    // ---- output straight from the C/D layout of the 32x32 MFMA: col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 kh.
    // Per-wave descriptors over this wave's rows of out / res (rows past M fall outside: dropped / zero); per lane ONE byte
    // offset (4 kh rows + channel), the register's row travels in the scalar offset: one store per value, no 64-bit math.
    const float scale = a.scale;
    const int Cout = a.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        a.out + (int64_t)(m0 < M ? m0 : 0) * Cout, 0, rows * Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res + (int64_t)(m0 < M ? m0 : 0) * Cout : a.out), 0, a.res ? rows * Cout * 4 : 0, 0x00020000);
    const bool has_res = a.res != nullptr, has_b2 = a.bias2 != nullptr;
    float s1[4], s2[4], piv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + t * 32 + li;
        const float bias = a.bias ? a.bias[n] : 0.f;
        const unsigned vo = (unsigned)((4 * kh) * Cout + n) * 4u;
        piv[t] = 0.f; s1[t] = 0.f; s2[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);       // + 4 kh (in vo)
            const unsigned so = (unsigned)(row * Cout) * 4u;
            float v = acc[t][r] + bias;
            if (has_b2) {
                const int mo = m0 + row + 4 * kh;
                v += a.bias2[(int64_t)((mo < M ? mo : 0) / HW) * a.bias2_stride + n];
            }
            if (has_res) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, vo, so, 0));
            v *= scale;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, vo, so, 0);
            if (r == 0) piv[t] = v;
            const float d = v - piv[t];
            s1[t] += d;
            s2[t] = fmaf(d, d, s2[t]);
        }
    }
    if (!a.stats) return;
    // statistics: 16 pixels per lane and channel -> the other pixel half (lane ^ 32, equal counts) -> this wave's 32
    // pixels -> the block's 256 pixels through LDS (launch guarantees: H W % 256 == 0, all rows valid)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float mean = piv[t] + s1[t] * (1.f / 16.f);
        float m2 = fmaxf(s2[t] - s1[t] * s1[t] * (1.f / 16.f), 0.f);
        const float mo = __shfl_xor(mean, 32), qo = __shfl_xor(m2, 32);
        const float d = mo - mean;
        m2 = m2 + qo + d * d * 8.f;
        mean = 0.5f * (mean + mo);
        if (kh == 0) {
            red[(wave * 128 + t * 32 + li) * 2] = mean;
            red[(wave * 128 + t * 32 + li) * 2 + 1] = m2;
        }
    }
    __syncthreads();
    if (tid < 128) {
        // equal-count partials (32 pixels each), merged pairwise in a fixed tree
        float mu[C1_WAVES], q[C1_WAVES];
#pragma unroll
        for (int w = 0; w < C1_WAVES; ++w) { mu[w] = red[(w * 128 + tid) * 2]; q[w] = red[(w * 128 + tid) * 2 + 1]; }
        float cnt = 32.f;
#pragma unroll
        for (int step = 1; step < C1_WAVES; step <<= 1) {
#pragma unroll
            for (int w = 0; w < C1_WAVES; w += 2 * step) {
                const float d = mu[w + step] - mu[w];
                q[w] = q[w] + q[w + step] + d * d * (0.5f * cnt);
                mu[w] = 0.5f * (mu[w] + mu[w + step]);
            }
            cnt *= 2.f;
        }
        const int mblk = mb * (C1_WAVES * C1_PX);
        const int bs = mblk / HW, blk = (mblk - bs * HW) >> 8;
        float* dst = a.stats + (((int64_t)bs * a.stats_nblk + blk) * a.Cout + n0 + tid) * 2;
        dst[0] = mu[0];
        dst[1] = q[0];
    }
}

// (Round 5, measured and removed: a persistent form of this kernel -- a wave walks several items, the finished item leaves
// through a second accumulator set, or through a wave-private LDS tile as one 16-byte store per k-block step of the next
// item's loop -- hides the output phase, 9.5 of a block's 38 us by s_memrealtime stamps, completely, and measures EQUAL:
// 109.3 vs 109.5 TFLOP/s on [8,256,256,256 -> 128], also with the stores removed altogether (113.8).  The K loop of this
// kernel already keeps the matrix pipes 100 % busy for 27 us of every 38; run back to back the same loop takes 16 % more
// cycles at a 6 % lower clock -- the launch is bound by what the part grants an MFMA-dense fp32 kernel, not by its schedule.)
// shapes the streaming kernel takes: fp32 1x1, more than 2048 pixels (below: conv_smallm.hip), whole 256-pixel blocks per
// sample, 32-aligned input channels, Cout a multiple of 128.  FLOWSE_NO_STREAM1X1=1: the flat kernel of rounds 1-4 (A-B hook)
static const bool g_no_stream = getenv("FLOWSE_NO_STREAM1X1") != nullptr || getenv("FLOWSE_FORCE_GENERIC_CONV") != nullptr;
bool conv1x1_stream_ok(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (g_no_stream || taps != 1 || (C1 % KC) || (C2 % KC) || (Cout % 128) || C1 <= 0) return false;
    const int64_t M = (int64_t)B * H * W;
    // at least one block per CU (256 pixels x 128 channels each); below that the flat kernel's K slices fill the chip better
    if (M <= 2048 || ((H * W) % 256) != 0 || (M / 256) * (Cout / 128) < 256) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return 32 * cmax * 4 < (1LL << 31) && (int64_t)Cout * (C1 + C2) * 4 < (1LL << 31) && M * Cout < (1LL << 40);
}
int conv1x1_stream_stats_blocks(int B, int H, int W) { (void)B; return H * W / 256; }

int launch_1x1_stream(const ConvArgs& a, hipStream_t s) {
    if (!a.wsm || a.in_dt != DT_F32 || a.out_dt != DT_F32 || a.gn.mean || a.partial || a.ksplit > 1 ||
        !conv1x1_stream_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        set_error("conv1x1_stream: unsupported configuration");
        return ERR_ARG;
    }
    if (a.stats && a.stats_nblk != conv1x1_stream_stats_blocks(a.B, a.H, a.W)) {
        set_error("conv1x1_stream: inconsistent fused-stats geometry (stats_nblk=%d)", a.stats_nblk);
        return ERR_ARG;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int mblocks = (int)((M + C1_WAVES * C1_PX - 1) / (C1_WAVES * C1_PX));
    hipLaunchKernelGGL(conv1x1_stream_kernel, dim3(mblocks * (a.Cout >> 7)), dim3(64 * C1_WAVES), 0, s, a);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
