// Direct (exact fmaf chain) LDS-halo 3x3 kernels in fp32 and the 4-channel boundary layers.
//   conv3x3_halo_kernel        8x16 pixel tile, halo of a 32-channel chunk staged once for all nine taps, fused GroupNorm+SiLU on
//                              the way into LDS: FLOWSE_NO_WINOGRAD=1, shapes the F(4,3) kernel does not take (Cout % 64 != 0)
//   conv3x3_head4_kernel       progressive-output heads C -> 4 (flowmse/backbones/ncsnpp.py:345-366) on v_mfma_f32_4x4x1
//   conv3x3_cin4_mfma_kernel   input layer 4 -> 128 (ncsnpp.py:285) on the matrix cores
//   conv_cin4_kernel           VALU conv for 4 input channels (Combine 1x1, layerspp.py:44-59; small / odd input layers)
#include "conv_common.h"

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// 3x3 convolution with an LDS-resident halo tile (the production kernel for H, W >= 64).
//
// The block owns an 8 x 16 pixel tile of one image and BN output channels.  For each 32-channel chunk the
// (8+2) x (16+2) input halo is staged into LDS ONCE and all nine taps read their shifted A fragments from it
// (tap = a constant row offset into the halo), so the activation crosses L2 -> LDS once per chunk instead of
// nine times, and only the per-tap weight tile is streamed per step.  Because every input element is staged
// exactly once per block, the GroupNorm + SiLU of the consuming ResnetBlock (layerspp.py:246,265: Conv(act(GN(x))))
// is applied right there, on the way into LDS -- the normalised tensor never exists in HBM.  Zero padding stays
// exact: out-of-image halo pixels are written as 0 AFTER the activation.
#define FLOWSE_HTAP 1
// GN: 0 = plain input, 1 = GroupNorm affine while staging, 2 = GroupNorm + SiLU
template <int WM, int WN, int TM, int TN, int GN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3x3_halo_kernel(ConvArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    static_assert(BM == 128 && NT == 256, "8x16 pixel tile, 4 waves");
    constexpr int HROWS = 180;                          // 10 x 18 halo pixels
    constexpr int H_LOADS = (HROWS * 8 + NT - 1) / NT;  // 6 float4 per thread
    constexpr int B_LOADS = (BN * 8 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                   // [HROWS][LDS_ROW]
    float* Bs = smem + HROWS * LDS_ROW;                 // [2][BN][LDS_ROW]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = (a.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 3);
    const int b = mt / tiles_img, tt = mt - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * 8, x0 = tx * 16, n0 = nt * BN;
    const int m_tl = (b * H + y0) * W + x0;              // top-left output pixel

    const int col4 = tid & 7, row0 = tid >> 3;
    unsigned hvo1[H_LOADS], hvo2[H_LOADS];
    unsigned hin = 0;                                    // bit q: halo row q of this thread lies inside the image
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HROWS && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        hvo1[q] = in ? (unsigned)((hy * W + hx) * C1 + col4 * 4) * 4u : OOB;
        hvo2[q] = in ? (unsigned)((hy * W + hx) * C2 + col4 * 4) * 4u : OOB;
        hin |= in ? (1u << q) : 0u;
    }
    unsigned bvo[B_LOADS];
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
        const int r = row0 + 32 * q;
        const int n = n0 + r;
        bvo[q] = (r < BN && n < a.Cout) ? (unsigned)(n * 9 * Cin + col4 * 4) * 4u : OOB;
    }
    const int64_t wbase = (int64_t)m_tl - W - 1;         // window origin = halo pixel (0,0)
    const int wpix = 9 * W + 18;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + wbase * C1), 0, wpix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + wbase * C2 : a.in1), 0, C2 ? wpix * C2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.Cout * 9 * Cin * 4, 0x00020000);

    u32x4 rh[H_LOADS], rb[B_LOADS];
    float4 g_mu, g_sc, g_be;                             // GroupNorm parameters of the staged channel quad

    auto gloadH = [&](int chunk) {
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff = (unsigned)(second ? c0 - C1 : c0) * 4u;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q)
            rh[q] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, hvo2[q], soff, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, hvo1[q], soff, 0);
        if (GN) {
            const int cg = c0 + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
    };
    // GroupNorm + SiLU on the staged registers (VALU only; runs under the partner wave's MFMAs), then the plain
    // LDS write once every wave has left the previous chunk's halo.  v_exp / v_rcp based SiLU: ~2 ulp.
    auto xform1 = [&](int q) {
        if (!GN) return;
        rh[q] = gn_quad<GN>(rh[q], g_mu, g_sc, g_be, (hin >> q) & 1u);
    };
    auto xformH = [&]() {
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) xform1(q);
    };
    auto lstoreH = [&]() {
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            if (hr < HROWS) *reinterpret_cast<u32x4*>(Hs + hr * LDS_ROW + col4 * 4) = rh[q];
        }
    };
    auto gloadB = [&](int s) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const unsigned soff_b = (unsigned)(tap * Cin + chunk * KC) * 4u;
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo[q], soff_b, 0);
    };
    auto lstoreB = [&](int buf) {
        float* Bb = Bs + buf * BN * LDS_ROW;
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) {
            const int r = row0 + 32 * q;
            if (r < BN) *reinterpret_cast<u32x4*>(Bb + r * LDS_ROW + col4 * 4) = rb[q];
        }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, kh = lane >> 5;
    // MFMA tile i of this wave covers tile rows 2*(wm*TM+i), +1; this lane's pixel inside it:
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int py = 2 * (wm * TM + i) + (li >> 4), px = li & 15;
        abase[i] = ((py + 1) * 18 + px + 1) * LDS_ROW + kh * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = Cin / KC;
    const int S_all = nchunks * 9;
    constexpr int HTAP = GN ? FLOWSE_HTAP : 7;           // tap at which the next chunk's halo is requested

    gloadH(0);
    gloadB(0);
    xformH();
    lstoreH();
    lstoreB(0);
    __syncthreads();

    // One K step with a literal TAP; the nine taps of a chunk are straight-line code and every load is unconditional
    // (clamped at the tail), so hipcc's s_waitcnt bookkeeping stays exact: the weight tile of step s+1 and the halo of
    // chunk c+1 (requested at tap 7, normalised in registers at tap 8, written after tap 8's barrier) stay in flight
    // under the MFMAs instead of being drained by a conservative vmcnt(0) at a control-flow join.
#define FLOWSE_STEP32(TAP)                                                                                           \
    {                                                                                                                \
        constexpr int tap = TAP;                                                                                     \
        const int s = chunk * 9 + tap;                                                                               \
        const int buf = s & 1;                                                                                       \
        gloadB(min(s + 1, S_all - 1));                                                                               \
        if (tap == HTAP) gloadH(min(chunk + 1, nchunks - 1));                                                        \
        if (HTAP == 7 && tap == 8) xformH();                                                                         \
        if (HTAP < 7 && tap > HTAP && tap - HTAP - 1 < H_LOADS) xform1(tap - HTAP - 1);                              \
        constexpr int tapoff = ((tap / 3 - 1) * 18 + (tap % 3 - 1)) * LDS_ROW;                                       \
        const float* Bb = Bs + buf * BN * LDS_ROW + (wn * TN * 32 + li) * LDS_ROW + kh * 4;                          \
        _Pragma("unroll") for (int j = 0; j < KC / 8; ++j) {                                                         \
            float4 af[TM], bf[TN];                                                                                   \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
                af[i] = *reinterpret_cast<const float4*>(Hs + abase[i] + tapoff + j * 8);                            \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                           \
                bf[i] = *reinterpret_cast<const float4*>(Bb + i * 32 * LDS_ROW + j * 8);                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) {       \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[jn].x, acc[i][jn], 0, 0, 0);           \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[jn].y, acc[i][jn], 0, 0, 0);           \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[jn].z, acc[i][jn], 0, 0, 0);           \
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[jn].w, acc[i][jn], 0, 0, 0);           \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        lstoreB(buf ^ 1); /* at the very last step: a spare tile into the idle buffer */                             \
        __syncthreads();                                                                                             \
        if (tap == 8) { /* everyone is done with this chunk's halo */                                                \
            lstoreH();                                                                                               \
            __syncthreads();                                                                                         \
        }                                                                                                            \
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        FLOWSE_STEP32(0) FLOWSE_STEP32(1) FLOWSE_STEP32(2) FLOWSE_STEP32(3) FLOWSE_STEP32(4)
        FLOWSE_STEP32(5) FLOWSE_STEP32(6) FLOWSE_STEP32(7) FLOWSE_STEP32(8)
    }
#undef FLOWSE_STEP32
    conv_epilogue<WM, WN, TM, TN>(a, acc, smem, m_tl, n0, M, HW, 0, W);
}

template <int WM, int WN, int TM, int TN>
static int launch_halo(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)(M / BM) * ((a.Cout + BN - 1) / BN);
    const size_t lds_stage = (180 + 2 * BN) * LDS_ROW * sizeof(float);
    const size_t lds_epi = ((size_t)BM * (BN + 4) + 64 * WM * WN * 8) * sizeof(float);
    const size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    if (const int rc = allow_lds<&conv3x3_halo_kernel<WM, WN, TM, TN, 0>>(lds)) return rc;
    if (const int rc = allow_lds<&conv3x3_halo_kernel<WM, WN, TM, TN, 1>>(lds)) return rc;
    if (const int rc = allow_lds<&conv3x3_halo_kernel<WM, WN, TM, TN, 2>>(lds)) return rc;
    if (a.gn.mean && a.gn_silu)
        hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, TM, TN, 2>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    else if (a.gn.mean)
        hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, TM, TN, 1>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    else
        hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, TM, TN, 0>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_halo_fp32(const ConvArgs& a, hipStream_t s) {
    if (a.Cout <= 32) return launch_halo<4, 1, 1, 1>(a, s);
    if (a.Cout <= 64) return launch_halo<2, 2, 2, 1>(a, s);
    // fewer than two 128x128 tiles per CU (single utterances): halve the N tile so that two blocks share every
    // CU and cover each other's barriers / prologues
    const int64_t tiles128 = ((int64_t)a.B * a.H * a.W / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 512) return launch_halo<2, 2, 2, 1>(a, s);
    return launch_halo<2, 2, 2, 2>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 convolution to FOUR output channels: the progressive-output heads (ncsnpp.py:345-366,
// pyramid = up(pyramid) + conv3x3(act(GroupNorm(h)))).  With N = 4 the 32-wide MFMA tiles would waste 7/8 of the
// matrix work, so this kernel uses v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 outer products per
// instruction = 64 pixels x 4 channels x one k, no padding anywhere.  Lane l supplies pixel l of the wave's 4 x 16
// pixel strip (A) and weight column l & 3 (B); accumulator register r of lane l holds pixel (l & ~3) + r, channel l & 3.
// Block = 16 x 16 pixels, 4 waves; per 32-channel chunk the 18 x 18 halo (GroupNorm + SiLU fused as in the other
// halo kernels) and the 4 x 9 x 32 weights sit in LDS; a lane's float4 fragment feeds four MFMAs.  HBM-read bound.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int GN, class ST = float>
__global__ __launch_bounds__(256, 3) void conv3x3_head4_kernel(ConvArgs a) {
    constexpr unsigned ES = sizeof(ST);                  // input element size (the 4-channel res / out stay fp32)
    constexpr int HPIX = 18 * 18;                        // halo pixels
    constexpr int H_LOADS = (HPIX * 8 + 255) / 256;      // 11 float4 per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                    // [HPIX][LDS_ROW]
    float* Ws = smem + HPIX * LDS_ROW;                   // [4][9][LDS_ROW]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int Cin = a.C1;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 4);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / tiles_img, tt = bid - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const int m_tl = (b * H + y0) * W + x0;

    const int col4 = tid & 7, row0 = tid >> 3;
    unsigned hvo[H_LOADS];
    unsigned hin = 0;
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HPIX && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        hvo[q] = in ? (unsigned)((hy * W + hx) * Cin + col4 * 4) * ES : OOB;
        hin |= in ? (1u << q) : 0u;
    }
    const int64_t wbase = (int64_t)m_tl - W - 1;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ST*>(reinterpret_cast<const ST*>(a.in1) + wbase * Cin), 0, (17 * W + 18) * Cin * (int)ES, 0x00020000);

    u32x4 rh[H_LOADS];
    float4 g_mu, g_sc, g_be, rw0, rw1;
    auto gload = [&](int chunk) {
        const unsigned soff = (unsigned)(chunk * KC) * ES;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) rh[q] = buf_ld_quad<ST>(rsrc1, hvo[q], soff);
        if (GN) {
            const int cg = chunk * KC + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
        // weights of this chunk: 4 x 9 rows of 8 float4 = 288 float4, thread t takes t and (t < 32) t + 256
        rw0 = *reinterpret_cast<const float4*>(a.w + (int64_t)(tid >> 3) * Cin + chunk * KC + col4 * 4);
        if (tid < 32) rw1 = *reinterpret_cast<const float4*>(a.w + (int64_t)(32 + (tid >> 3)) * Cin + chunk * KC + col4 * 4);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            if (GN) rh[q] = gn_quad<GN>(rh[q], g_mu, g_sc, g_be, (hin >> q) & 1u);
            if (hr < HPIX) *reinterpret_cast<u32x4*>(Hs + hr * LDS_ROW + col4 * 4) = rh[q];
        }
        *reinterpret_cast<float4*>(Ws + (tid >> 3) * LDS_ROW + col4 * 4) = rw0;
        if (tid < 32) *reinterpret_cast<float4*>(Ws + (32 + (tid >> 3)) * LDS_ROW + col4 * 4) = rw1;
    };

    const int lane = tid & 63, wave = tid >> 6;
    // lane = pixel of the wave's 4 x 16 strip: row lane >> 4, column lane & 15 (halo coordinates +1)
    const float* Ap = Hs + ((4 * wave + (lane >> 4)) * 18 + (lane & 15)) * LDS_ROW;
    const float* Bp = Ws + (lane & 3) * 9 * LDS_ROW;
    // two accumulator chains (channels 0, 2 / 1, 3 of a quad): a two-pass MFMA issued straight behind the one it accumulates
    // onto waits for it (pyramid_conv@256x256: 111 -> 102 us); summed once at the end
    f32x4v acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    const int nchunks = Cin / KC;
    gload(0);
    lstore();
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        gload(min(chunk + 1, nchunks - 1));              // next chunk into registers under this chunk's MFMAs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* At = Ap + ((tap / 3) * 18 + (tap % 3)) * LDS_ROW;
            const float* Bt = Bp + tap * LDS_ROW;
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                const float4 av = *reinterpret_cast<const float4*>(At + q * 4);
                const float4 bv = *reinterpret_cast<const float4*>(Bt + q * 4);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, bv.x, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, bv.y, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, bv.z, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, bv.w, acc1, 0, 0, 0);
            }
        }
        __syncthreads();                                 // everyone has left this chunk's tiles
        lstore();
        __syncthreads();
    }
    acc += acc1;
    // accumulator register r: pixel (lane & ~3) + r of the strip, channel lane & 3
    const int j = lane & 3;
    const float bj = a.bias ? a.bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = (lane & ~3) + r;
        const int64_t m = (int64_t)m_tl + (4 * wave + (p >> 4)) * W + (p & 15);
        float v = acc[r] + bj;
        if (a.res) v += a.res[m * 4 + j];
        a.out[m * 4 + j] = v * a.scale;
    }
}

// The same heads on 16-bit storage (BASELINE configs 2 / 4).  Rounds 2-4 widened the 16-bit input to fp32 and ran the fp32
// kernel above: half the bytes, the same 288 v_mfma_f32_4x4x1 per wave and chunk -- the same time as in fp32 (123 us at
// [8,256,256,128], 1.2 TB/s: matrix-issue bound, 2.6 % of config[4]).  Here the halo and the weights sit in LDS in the storage
// type (GroupNorm + SiLU in fp32 on the way in, ONE rounding, as for every other 16-bit conv operand) and one
// v_mfma_f32_4x4x4 takes four channels of a pixel at once: 72 matrix instructions per wave and chunk.  A lane's A operand =
// 4 consecutive channels of its pixel (8 bytes), B = the same 4 channels of weight row lane & 3; accumulator layout as above.
// Staging: 16-byte pieces (8 channels of a halo pixel), all requested before the first is used.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int H16_ROWB = 72;                         // bytes per halo pixel / weight row: 32 x 16 bit + 8 pad (conflict-free b64 reads)

template <int GN, class ST>
__global__ __launch_bounds__(256, 3) void conv3x3_head4_16_kernel(ConvArgs a) {
    constexpr bool F16 = St<ST>::dt == DT_F16;
    constexpr int HPIX = 18 * 18;
    constexpr int PIECES = (HPIX * 4 + 255) / 256;       // 16-byte pieces per thread and chunk: 6 (the last one for 16 threads)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Hs = reinterpret_cast<char*>(smem);            // [HPIX][H16_ROWB]
    char* Ws = Hs + HPIX * H16_ROWB;                     // [4][9][H16_ROWB]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int Cin = a.C1;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 4);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / tiles_img, tt = bid - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const int m_tl = (b * H + y0) * W + x0;

    const int oct = tid & 3, hp0 = tid >> 2;             // this thread's 8-channel group; halo pixels hp0 + 64 q
    unsigned hvo[PIECES];
    unsigned hin = 0;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int hr = hp0 + 64 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HPIX && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        hvo[q] = in ? (unsigned)((hy * W + hx) * Cin + oct * 8) * 2u : OOB;
        hin |= in ? (1u << q) : 0u;
    }
    const int64_t wbase = (int64_t)m_tl - W - 1;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ST*>(reinterpret_cast<const ST*>(a.in1) + wbase * Cin), 0, (17 * W + 18) * Cin * 2, 0x00020000);

    u32x4 rh[PIECES];
    float4 g_mu[2], g_sc[2], g_be[2], rw0, rw1;
    auto gload = [&](int chunk) {
        const unsigned soff = (unsigned)(chunk * KC) * 2u;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) rh[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc1, hvo[q], soff, 0);
        if (GN) {
            const int cg = chunk * KC + oct * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                g_mu[h] = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg + 4 * h);
                g_sc[h] = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg + 4 * h);
                g_be[h] = *reinterpret_cast<const float4*>(a.gn.beta + cg + 4 * h);
            }
        }
        // weights of this chunk (fp32 [4][9][Cin]): 4 x 9 rows of 8 float4 = 288 float4, thread t takes t and (t < 32) t + 256
        rw0 = *reinterpret_cast<const float4*>(a.w + (int64_t)(tid >> 3) * Cin + chunk * KC + (tid & 7) * 4);
        if (tid < 32) rw1 = *reinterpret_cast<const float4*>(a.w + (int64_t)(32 + (tid >> 3)) * Cin + chunk * KC + (tid & 7) * 4);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int hr = hp0 + 64 * q;
            u32x4 o = rh[q];
            if (GN) {
                const unsigned wsrc[4] = {rh[q].x, rh[q].y, rh[q].z, rh[q].w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) St<ST>::unpack2(wsrc[e], v[2 * e], v[2 * e + 1]);
                const float mu[8] = {g_mu[0].x, g_mu[0].y, g_mu[0].z, g_mu[0].w, g_mu[1].x, g_mu[1].y, g_mu[1].z, g_mu[1].w};
                const float sc[8] = {g_sc[0].x, g_sc[0].y, g_sc[0].z, g_sc[0].w, g_sc[1].x, g_sc[1].y, g_sc[1].z, g_sc[1].w};
                const float be[8] = {g_be[0].x, g_be[0].y, g_be[0].z, g_be[0].w, g_be[1].x, g_be[1].y, g_be[1].z, g_be[1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = fmaf(v[e] - mu[e], sc[e], be[e]);
                    if (GN == 2) y *= __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y));
                    v[e] = y;
                }
                const unsigned keep = ((hin >> q) & 1u) ? 0xffffffffu : 0u;      // zero padding AFTER the activation
                o.x = St<ST>::pack2(v[0], v[1]) & keep; o.y = St<ST>::pack2(v[2], v[3]) & keep;
                o.z = St<ST>::pack2(v[4], v[5]) & keep; o.w = St<ST>::pack2(v[6], v[7]) & keep;
            }
            if (hr < HPIX) {                                 // (rows are 8-byte aligned: two b64 halves)
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(Hs + hr * H16_ROWB + oct * 16) = u32x2{o.x, o.y};
                *reinterpret_cast<u32x2*>(Hs + hr * H16_ROWB + oct * 16 + 8) = u32x2{o.z, o.w};
            }
        }
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(Ws + (tid >> 3) * H16_ROWB + (tid & 7) * 8) = u32x2{St<ST>::pack2(rw0.x, rw0.y), St<ST>::pack2(rw0.z, rw0.w)};
        if (tid < 32)
            *reinterpret_cast<u32x2*>(Ws + (32 + (tid >> 3)) * H16_ROWB + (tid & 7) * 8) = u32x2{St<ST>::pack2(rw1.x, rw1.y), St<ST>::pack2(rw1.z, rw1.w)};
    };

    const int lane = tid & 63, wave = tid >> 6;
    const char* Ap = Hs + ((4 * wave + (lane >> 4)) * 18 + (lane & 15)) * H16_ROWB;
    const char* Bp = Ws + (lane & 3) * 9 * H16_ROWB;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    const int nchunks = Cin / KC;
    gload(0);
    lstore();
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        gload(min(chunk + 1, nchunks - 1));              // next chunk into registers under this chunk's MFMAs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* At = Ap + ((tap / 3) * 18 + (tap % 3)) * H16_ROWB;
            const char* Bt = Bp + tap * H16_ROWB;
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 av = *reinterpret_cast<const u32x2*>(At + q * 8);
                const u32x2 bv = *reinterpret_cast<const u32x2*>(Bt + q * 8);
                f32x4v& ac = (q & 1) ? acc1 : acc;
                if (F16) ac = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, av), __builtin_bit_cast(h16x4, bv), ac, 0, 0, 0);
                else ac = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, av), __builtin_bit_cast(s16x4, bv), ac, 0, 0, 0);
            }
        }
        __syncthreads();                                 // everyone has left this chunk's tiles
        lstore();
        __syncthreads();
    }
    acc += acc1;
    // accumulator register r: pixel (lane & ~3) + r of the strip, channel lane & 3
    const int j = lane & 3;
    const float bj = a.bias ? a.bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = (lane & ~3) + r;
        const int64_t m = (int64_t)m_tl + (4 * wave + (p >> 4)) * W + (p & 15);
        float v = acc[r] + bj;
        if (a.res) v += a.res[m * 4 + j];
        a.out[m * 4 + j] = v * a.scale;
    }
}

bool conv_supports_head4(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    return !conv_force_generic() && taps == 9 && Cout == 4 && C2 == 0 && (C1 % KC) == 0 && !(H & 15) && !(W & 15) &&
           (int64_t)B * (H >> 4) * (W >> 4) >= 64 && (int64_t)(17 * W + 18) * C1 * 4 < (1LL << 31);
}

// FLOWSE_HEAD4_FP32=1: the 16-bit modes widen the heads' input and run the fp32 kernel, as in rounds 2-4 (A-B hook)
static const bool g_head4_fp32 = getenv("FLOWSE_HEAD4_FP32") != nullptr;
int launch_head4(const ConvArgs& a, hipStream_t s) {
    const int grid = a.B * (a.H >> 4) * (a.W >> 4);
    const size_t lds = (size_t)(18 * 18 + 36) * LDS_ROW * sizeof(float);
    if (a.out_dt != DT_F32) {
        set_error("head4: the 4-channel output is fp32");
        return ERR_ARG;
    }
    if (a.in_dt != DT_F32 && !g_head4_fp32) {            // 16-bit storage: 16-bit operands on v_mfma_f32_4x4x4
        const size_t lds16 = (size_t)(18 * 18 + 36) * H16_ROWB;
        const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
#define FLOWSE_H16(GNF)                                                                                                     \
    if (a.in_dt == DT_BF16) hipLaunchKernelGGL((conv3x3_head4_16_kernel<GNF, bf16_t>), dim3(grid), dim3(256), lds16, s, a); \
    else hipLaunchKernelGGL((conv3x3_head4_16_kernel<GNF, f16_t>), dim3(grid), dim3(256), lds16, s, a);
        if (gn == 2) { FLOWSE_H16(2) } else if (gn == 1) { FLOWSE_H16(1) } else { FLOWSE_H16(0) }
#undef FLOWSE_H16
        FLOWSE_LAUNCH_CHECK();
        return OK;
    }
    if (a.gn.mean && a.gn_silu) {
        FLOWSE_DT_SWITCH(a.in_dt, ST, hipLaunchKernelGGL((conv3x3_head4_kernel<2, ST>), dim3(grid), dim3(256), lds, s, a));
    } else if (a.gn.mean) {
        FLOWSE_DT_SWITCH(a.in_dt, ST, hipLaunchKernelGGL((conv3x3_head4_kernel<1, ST>), dim3(grid), dim3(256), lds, s, a));
    } else {
        FLOWSE_DT_SWITCH(a.in_dt, ST, hipLaunchKernelGGL((conv3x3_head4_kernel<0, ST>), dim3(grid), dim3(256), lds, s, a));
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// ---------------------------------------------------------------------------------------------------
// Direct VALU convolution for 4 input channels: the input layer conv3x3 4->nf (ncsnpp.py:159,285) and the
// Combine conv1x1 4->C (layerspp.py:44-59).  K = 36 / 4 is too short for the matrix cores; these layers are
// bound by the HBM write of the output.  One thread = one pixel x 4 output channels; weights live in LDS.
template <int TAPS, class OT>
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
        const float4 r = St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + off);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    St<OT>::st4(reinterpret_cast<OT*>(a.out) + off,
                make_float4(o[0] * a.scale, o[1] * a.scale, o[2] * a.scale, o[3] * a.scale));
}


// Combine (conv1x1 4 -> C plus the in-place residual, layerspp.py:55-59) WITH the GroupNorm partial statistics of what it
// writes: a block owns PB = min(128, H W) consecutive pixels of one sample (its threads = Cout/4 channel quads x 256/(Cout/4)
// pixel rows walk them), accumulates pivoted (mean, M2) per thread and merges the pixel rows through LDS.  Saves the
// gn_stats launch that used to re-read the tensor right after this kernel wrote it.  OT = the storage type of res / out (the
// 4-channel input is always fp32); the statistics are those of the values as stored (rounded to OT).
template <class OT>
__global__ __launch_bounds__(256) void conv_cin4_stats_kernel(ConvArgs a, int Q, int PB) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cout][4] weights, then [R][Q][8] statistics scratch
    const int tid = threadIdx.x;
    for (int i = tid; i < a.Cout; i += 256) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(a.w)[i];
    __syncthreads();
    float* red = wl + a.Cout * 4;
    const int R = 256 / Q;
    const int HW = a.H * a.W;
    const int cq = tid % Q, pr = tid / Q;
    const int64_t m0 = (int64_t)blockIdx.x * PB;
    const int bs = (int)(m0 / HW);
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + cq * 4);
    if (a.bias2) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)bs * a.bias2_stride + cq * 4);
        bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
    }
    float4 w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const float4*>(wl + (cq * 4 + j) * 4);
    Stat4 st;
    st.init();
    // four pixels per round, all of their loads requested before the first is used (in place: `out` aliases `res`, so the
    // compiler keeps a load behind the store before it -- one pixel per round was a chain of 32 memory round trips per thread:
    // 24 us for a 32 x 32 level)
    for (int p0 = pr; p0 < PB; p0 += 4 * R) {
        float4 xv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * R;
            const int64_t m = m0 + (p < PB ? p : pr);
            xv[u] = *reinterpret_cast<const float4*>(a.in1 + m * 4);
            rv[u] = a.res ? St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + m * a.Cout + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * R;
            if (p >= PB) break;
            const float4 x = xv[u];
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
                acc = fmaf(x.x, w4[j].x, acc);
                acc = fmaf(x.y, w4[j].y, acc);
                acc = fmaf(x.z, w4[j].z, acc);
                acc = fmaf(x.w, w4[j].w, acc);
                o[j] = acc;
            }
            float4 v = make_float4(o[0] + bq.x, o[1] + bq.y, o[2] + bq.z, o[3] + bq.w);
            if (a.res) { v.x += rv[u].x; v.y += rv[u].y; v.z += rv[u].z; v.w += rv[u].w; }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            St<OT>::st4(reinterpret_cast<OT*>(a.out) + (m0 + p) * a.Cout + cq * 4, v);
            st.add(St<OT>::rnd4(v));
        }
    }
    st.finish(red + (pr * Q + cq) * 8);
    __syncthreads();
    if (pr == 0) {
        float acc8[8], nacc = (float)(PB / R);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = red[cq * 8 + j];
        for (int r = 1; r < R; ++r) chan_merge4(nacc, acc8, (float)(PB / R), red + (r * Q + cq) * 8);
        const int blk = (int)((m0 - (int64_t)bs * HW) / PB);
        float* dst = a.stats + (((int64_t)bs * a.stats_nblk + blk) * a.Cout + cq * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[2 * j] = acc8[j];
            dst[2 * j + 1] = acc8[4 + j];
        }
    }
}

// statistics blocks per sample of the Combine kernel's fused statistics (0: shape not covered)
int conv_cin4_stats_blocks(int B, int H, int W, int Cout) {
    const int HW = H * W, PB = HW < 128 ? HW : 128, Q = Cout / 4;
    (void)B;
    if ((Cout & 3) || Q > 256 || (256 % Q) != 0 || (HW % PB) != 0 || (PB % (256 / Q)) != 0) return 0;
    return HW / PB;
}

// Matrix-core form of the 4 -> 128 input convolution for full-size images: K = 9 taps x 4 channels = 36 (+4 zero),
// a lane's A operand is simply the float4 of one neighbouring pixel (taps 2q for lanes 0-31, 2q+1 for lanes 32-63),
// read straight from global memory; the whole 128 x 40 weight matrix sits in registers.  Block = 128 flat pixels x 128
// channels, wave = 32 pixels x 128 channels (4 accumulator tiles); output-write bound.  Shares the standard epilogue,
// i.e. also emits the GroupNorm partial statistics of its output.
template <class OT>
__global__ __launch_bounds__(256, 2) void conv3x3_cin4_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int m = m0 + wave * 32 + li;                  // this lane's pixel (A row)
    const int rem = m % HW;
    const int y = rem / W, x = rem - y * W;
    float4 af[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = 2 * q + kh;                        // tap 9 does not exist: zero operand
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        const bool ok = t < 9 && m < M && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
        const float4 v = *reinterpret_cast<const float4*>(a.in1 + (int64_t)(ok ? m + dy * W + dx : 0) * 4);
        af[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x16 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    if constexpr (sizeof(OT) == 2) {
        // 16-bit storage modes: the same K = 36 (+ 12 zero) as THREE v_mfma_f32_32x32x16 per channel tile instead of twenty
        // 32x32x2 -- in fp32 the matrix instructions of this layer took longer than the HBM write of its output (105 us for
        // 134 MB at [8,256,256]).  Lane half kh of MFMA m supplies k = 16 m + 8 kh .. + 7 = taps 4 m + 2 kh and 4 m + 2 kh + 1
        // (four channels each), i.e. the float4 pairs already loaded, rounded once to the storage type like every other
        // operand of these modes; the weights likewise.
        constexpr bool F16 = St<OT>::dt == DT_F16;
        typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
        auto pack8 = [](const float4& lo, const float4& hi) {
            u32x4 r;
            r.x = St<OT>::pack2(lo.x, lo.y); r.y = St<OT>::pack2(lo.z, lo.w);
            r.z = St<OT>::pack2(hi.x, hi.y); r.w = St<OT>::pack2(hi.z, hi.w);
            return r;
        };
        // (the fp32 form's af[] holds the taps of parity kh only; here a lane needs taps 4 m + 2 kh and 4 m + 2 kh + 1)
        float4 at[3][2];
#pragma unroll
        for (int mq = 0; mq < 3; ++mq)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int t = 4 * mq + 2 * kh + sl;
                const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
                const bool ok = t < 9 && m < M && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const float4 v = *reinterpret_cast<const float4*>(a.in1 + (int64_t)(ok ? m + dy * W + dx : 0) * 4);
                at[mq][sl] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = j * 32 + li;
#pragma unroll
            for (int mq = 0; mq < 3; ++mq) {
                float4 wv[2];
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int t = 4 * mq + 2 * kh + sl;
                    wv[sl] = t < 9 ? *reinterpret_cast<const float4*>(a.w + ((int64_t)n * 9 + t) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const u32x4 av = pack8(at[mq][0], at[mq][1]), bv = pack8(wv[0], wv[1]);
                if (F16) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hx8, av), __builtin_bit_cast(hx8, bv), acc[0][j], 0, 0, 0);
                else acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[0][j], 0, 0, 0);
            }
        }
        conv_epilogue<4, 1, 1, 4, OT>(a, acc, smem, m0, 0, M, HW, 0);
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = j * 32 + li;                       // B row = output channel (Cout == 128)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int t = 2 * q + kh;
            const float4 w4 = t < 9 ? *reinterpret_cast<const float4*>(a.w + ((int64_t)n * 9 + t) * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, w4.x, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, w4.y, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, w4.z, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, w4.w, acc[0][j], 0, 0, 0);
        }
    }
    conv_epilogue<4, 1, 1, 4, OT>(a, acc, smem, m0, 0, M, HW, 0);
}

bool conv_cin4_uses_mfma(int B, int H, int W, int Cout, int taps) {
    return taps == 9 && Cout == 128 && ((H * W) % 128) == 0 && (int64_t)B * H * W >= 128 * 256 && !conv_force_generic();
}

int launch_conv_cin4(const ConvArgs& a, hipStream_t s) {
    const int Q = a.Cout / 4;
    if (a.C1 == 4 && a.C2 == 0 && a.ksplit <= 1 && !a.gn.mean && conv_cin4_uses_mfma(a.B, a.H, a.W, a.Cout, a.taps)) {
        const size_t lds = ((size_t)128 * (128 + 4) + 256 * 8) * sizeof(float);
        if (const int rc = allow_lds<&conv3x3_cin4_mfma_kernel<float>>(lds)) return rc;
        if (const int rc = allow_lds<&conv3x3_cin4_mfma_kernel<bf16_t>>(lds)) return rc;
        if (const int rc = allow_lds<&conv3x3_cin4_mfma_kernel<f16_t>>(lds)) return rc;
        const int grid = (int)((int64_t)a.B * a.H * a.W / 128);
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(conv3x3_cin4_mfma_kernel<OT>, dim3(grid), dim3(256), lds, s, a));
        FLOWSE_LAUNCH_CHECK();
        return OK;
    }
    if (a.C1 != 4 || a.C2 != 0 || (a.Cout & 3) || Q > 256 || (256 % Q) != 0 ||
        (size_t)a.Cout * a.taps * 16 > 64 * 1024) {
        return launch_conv(a, s);      // generic path handles any shape
    }
    if (a.stats) {                                       // Combine with fused statistics (1x1)
        const int nblk = conv_cin4_stats_blocks(a.B, a.H, a.W, a.Cout);
        if (a.taps != 1 || a.in_dt != DT_F32 || nblk == 0 || a.stats_nblk != nblk) {
            set_error("conv_cin4: fused statistics need a 1x1 conv of an fp32 input on whole statistics blocks (stats_nblk=%d)", a.stats_nblk);
            return ERR_ARG;
        }
        const int PB = a.H * a.W / nblk;
        const size_t lds_s = (size_t)a.Cout * 16 + (size_t)256 * 8 * sizeof(float);
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(conv_cin4_stats_kernel<OT>, dim3((unsigned)((int64_t)a.B * nblk)), dim3(256), lds_s, s, a, Q, PB));
        FLOWSE_LAUNCH_CHECK();
        return OK;
    }
    const int ppb = 256 / Q;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)((M + ppb - 1) / ppb);
    const size_t lds = (size_t)a.Cout * a.taps * 16;
    if (a.taps == 9) {
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL((conv_cin4_kernel<9, OT>), dim3(grid), dim3(256), lds, s, a, Q));
    } else {
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL((conv_cin4_kernel<1, OT>), dim3(grid), dim3(256), lds, s, a, Q));
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
