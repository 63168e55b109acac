// 16-bit matrix-core convolution kernels (v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate): the precision modes
// bf16x3 (split operands, fp32 storage), bf16 and fp16 (BASELINE configs 2 / 4: 16-bit activation storage).
// Replaces (reference) the same ops as the fp32 kernels: ddpm_conv3x3 / ddpm_conv1x1
// (flowmse/backbones/ncsnpp_utils/layers.py:100-124) inside ResnetBlockBigGANpp (layerspp.py:245-274).
//   conv3x3_halo_bf16_kernel   LDS-halo 3x3, per-tap weight tile, two blocks per CU: the operand modes on fp32 storage
//                              (bf16x3, bf16 / fp16 on networks the storage modes do not take) and, in the storage modes,
//                              images whose height is a multiple of 8 but not of 16 (conv16_pc.hip takes the rest)
//   conv_flat16_kernel         flat 1x1 / small 3x3 on 16-bit activations, split-K with fp32 slabs
//   convert_kernel             storage conversion at the boundaries; pack_conv_bf16: host-side operand planes
#include "conv_common.h"

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// bf16 matrix-core variant of the LDS-halo 3x3 kernel (optional precision modes, off by default).
//
// Same tiling and data flow as conv3x3_halo_kernel; the operands are bf16 for v_mfma_f32_32x32x16_bf16 (16x the
// fp32 MFMA rate).  TERMS = 3 ("bf16x3"): every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi),
// 16 mantissa bits kept) and the product is accumulated as hi*hi + hi*lo + lo*hi in fp32 -- the dropped lo*lo term
// is 2^-16 relative, so results stay fp32-class (measured ~1e-5 rel-L2 end to end) at 3/16 of the fp32 MFMA cost.
// TERMS = 1: plain bf16 operands (BASELINE config 3).  Activations stay fp32 in HBM and are split while the halo is
// staged (after the fused GroupNorm+SiLU); weights are pre-split at upload.  LDS row = [hi: 32 x bf16][lo: 32 x bf16]
// + 16 B pad (stride 144 B, or 80 B for one plane): every fragment read is a conflict-free ds_read_b128.
// F16 = true: IEEE half operands (v_mfma_f32_32x32x16_f16, 11-bit mantissa; BASELINE config 5), TERMS must be 1.
// IT / OT: storage types of the input tensors and of res / out (float in the operand-only modes; the 16-bit type of
// the operands in the 16-bit storage modes).
template <int TERMS, bool GN, bool F16 = false, class IT = float, class OT = float>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(ConvArgs a) {
    static_assert(!F16 || TERMS == 1, "the half path has no split mode");
    constexpr unsigned ES = sizeof(IT);
    constexpr int BN = 128, NT = 256;
    constexpr int PLANES = TERMS == 1 ? 1 : 2;
    constexpr int ROWB = PLANES * 64 + 16;               // LDS row stride in bytes
    constexpr int HROWS = 180;
    constexpr int H_LOADS = 6;                            // fp32 halo: 180 rows x 8 float4 / 256 threads
    constexpr int CPR = PLANES * 4;                       // 16-byte columns per weight row
    constexpr int B_LOADS = BN * CPR / NT;                // 4 (two planes) or 2 (one plane)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Hs = reinterpret_cast<char*>(smem);             // [HROWS][ROWB]
    char* Bs = Hs + HROWS * ROWB;                         // [2][BN][ROWB]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int nchunks = Cin / KC;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 3);
    const int b = mt / tiles_img, tt = mt - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * 8, x0 = tx * 16, n0 = nt * BN;
    const int m_tl = (b * H + y0) * W + x0;

    const int col4 = tid & 7, row0 = tid >> 3;            // halo staging: 8 float4 columns x 32 rows per pass
    unsigned hvo1[H_LOADS], hvo2[H_LOADS];
    unsigned hin = 0;
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HROWS && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        hvo1[q] = in ? (unsigned)((hy * W + hx) * C1 + col4 * 4) * ES : OOB;
        hvo2[q] = in ? (unsigned)((hy * W + hx) * C2 + col4 * 4) * ES : OOB;
        hin |= in ? (1u << q) : 0u;
    }
    const int bcol = tid % CPR, brow0 = tid / CPR;        // weight staging
    constexpr int BRPP = NT / CPR;                        // rows per pass
    unsigned bvo[B_LOADS];
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
        const int n = n0 + brow0 + BRPP * q;
        bvo[q] = (unsigned)(n * 9 * nchunks * (PLANES * 64) + bcol * 16);
    }
    const int64_t wbase = (int64_t)m_tl - W - 1;
    const int wpix = 9 * W + 18;
    const IT* in1p = reinterpret_cast<const IT*>(a.in1);
    const IT* in2p = reinterpret_cast<const IT*>(a.in2);
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<IT*>(in1p + wbase * C1), 0, wpix * C1 * (int)ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<IT*>(C2 ? in2p + wbase * C2 : in1p), 0, C2 ? wpix * C2 * (int)ES : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.wq), 0, a.Cout * 9 * nchunks * (PLANES * 64), 0x00020000);

    u32x4 rh[H_LOADS], rb[B_LOADS];
    float4 g_mu, g_sc, g_be;

    auto gloadH = [&](int chunk) {
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned soff = (unsigned)(second ? c0 - C1 : c0) * ES;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q)
            rh[q] = second ? buf_ld_quad<IT>(rsrc2, hvo2[q], soff) : buf_ld_quad<IT>(rsrc1, hvo1[q], soff);
        if (GN) {
            const int cg = c0 + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
    };
    // GroupNorm + SiLU, then the bf16 split; afterwards rh[q] = {hi01, hi23, lo01, lo23} (packed bf16 pairs)
    auto xformH = [&]() {
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const bool in = (hin >> q) & 1u;
            float v[4] = {__uint_as_float(rh[q].x), __uint_as_float(rh[q].y), __uint_as_float(rh[q].z),
                          __uint_as_float(rh[q].w)};
            if (GN) {
                const float mu[4] = {g_mu.x, g_mu.y, g_mu.z, g_mu.w}, sc[4] = {g_sc.x, g_sc.y, g_sc.z, g_sc.w},
                            be[4] = {g_be.x, g_be.y, g_be.z, g_be.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf(v[e] - mu[e], sc[e], be[e]);
                    if (a.gn_silu) t = fast_silu(t);
                    v[e] = in ? t : 0.f;
                }
            }
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (F16) {
                    const _Float16 h = (_Float16)v[e];
                    hi[e] = __builtin_bit_cast(unsigned short, h);
                    lo[e] = 0;
                } else {
                    const __bf16 h = (__bf16)v[e];
                    hi[e] = __builtin_bit_cast(unsigned short, h);
                    const __bf16 l = (__bf16)(v[e] - (float)h);
                    lo[e] = __builtin_bit_cast(unsigned short, l);
                }
            }
            rh[q].x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);
            rh[q].y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
            rh[q].z = (unsigned)lo[0] | ((unsigned)lo[1] << 16);
            rh[q].w = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
        }
    };
    auto lstoreH = [&]() {
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            if (hr >= HROWS) continue;
            char* p = Hs + hr * ROWB + col4 * 8;
            *reinterpret_cast<uint2*>(p) = make_uint2(rh[q].x, rh[q].y);
            if (PLANES == 2) *reinterpret_cast<uint2*>(p + 64) = make_uint2(rh[q].z, rh[q].w);
        }
    };
    // weight tiles travel through TWO register sets: the 16-bit MFMA phase of a step (~0.4 us) is shorter than an
    // L2 round trip, so tile s+2 is requested while step s computes and tile s+1 (requested a step earlier) is
    // written to LDS at the end of step s
    u32x4 rb2[B_LOADS];
    auto gloadB = [&](int s, u32x4 (&R)[B_LOADS]) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const unsigned soff_b = (unsigned)((tap * nchunks + chunk) * (PLANES * 64));
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) R[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo[q], soff_b, 0);
    };
    auto lstoreB = [&](int buf, u32x4 (&R)[B_LOADS]) {
        char* Bb = Bs + buf * BN * ROWB;
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q)
            *reinterpret_cast<u32x4*>(Bb + (brow0 + BRPP * q) * ROWB + bcol * 16) = R[q];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = 2 * (wm * 2 + i) + (li >> 4), px = li & 15;
        abase[i] = ((py + 1) * 18 + px + 1) * ROWB + kh * 16;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int S_all = nchunks * 9;
    gloadH(0);
    gloadB(0, rb);
    gloadB(min(1, S_all - 1), rb2);
    xformH();
    lstoreH();
    lstoreB(0, rb);
    __syncthreads();

    // One K step.  TAP is a literal and the nine taps of a chunk are emitted as straight-line code: no load sits
    // under a branch, so hipcc's s_waitcnt bookkeeping stays exact (counted vmcnt, never a drain).  RL = the register
    // set that is free (gets tile s+2), RS = the set holding tile s+1.  The next chunk's halo is requested at tap 7,
    // normalised / split in registers at tap 8 and written after tap 8's barrier; at the last chunk the (clamped)
    // reload of the same halo is redundant but harmless.  (A macro, not a lambda taking the sets by reference:
    // register arrays passed through generic lambdas end up in scratch memory.)
#define FLOWSE_STEP16(TAP, RL, RS)                                                                                   \
    {                                                                                                                \
        constexpr int tap = TAP;                                                                                     \
        const int s = chunk * 9 + tap;                                                                               \
        const int buf = s & 1;                                                                                       \
        gloadB(min(s + 2, S_all - 1), RL);                                                                           \
        if (tap == 7) gloadH(min(chunk + 1, nchunks - 1));                                                           \
        __builtin_amdgcn_sched_barrier(0); /* requests go out before the MFMAs (hipcc sinks them otherwise) */      \
        if (tap == 8) xformH();                                                                                      \
        constexpr int tapoff = ((tap / 3 - 1) * 18 + (tap % 3 - 1)) * ROWB;                                          \
        const char* Bb = Bs + buf * BN * ROWB + (wn * 64 + li) * ROWB + kh * 16;                                     \
        _Pragma("unroll") for (int mh = 0; mh < 2; ++mh) {                                                           \
            bf16x8 ah[2], al[2], bh[2], bl[2];                                                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
                const char* p = Hs + abase[i] + tapoff + mh * 32;                                                    \
                ah[i] = *reinterpret_cast<const bf16x8*>(p);                                                         \
                if (TERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(p + 64);                                    \
            }                                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
                const char* p = Bb + j * 32 * ROWB + mh * 32;                                                        \
                bh[j] = *reinterpret_cast<const bf16x8*>(p);                                                         \
                if (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(p + 64);                                    \
            }                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {            \
                if (F16) {                                                                                           \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]),             \
                                                                      __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0); \
                } else {                                                                                             \
                    if (TERMS == 3) {                                                                                \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);       \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);       \
                    }                                                                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);           \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        lstoreB(buf ^ 1, RS); /* at the very last step: a spare tile into the idle buffer */                         \
        __syncthreads();                                                                                             \
        if (tap == 8) { /* everyone is done with this chunk's halo */                                                \
            lstoreH();                                                                                               \
            __syncthreads();                                                                                         \
        }                                                                                                            \
    }
    // 9 steps per chunk, so the set parity alternates from chunk to chunk: two chunks per loop iteration
    for (int c2 = 0; c2 < nchunks; c2 += 2) {
        {
            const int chunk = c2;
            FLOWSE_STEP16(0, rb, rb2) FLOWSE_STEP16(1, rb2, rb) FLOWSE_STEP16(2, rb, rb2) FLOWSE_STEP16(3, rb2, rb)
            FLOWSE_STEP16(4, rb, rb2) FLOWSE_STEP16(5, rb2, rb) FLOWSE_STEP16(6, rb, rb2) FLOWSE_STEP16(7, rb2, rb)
            FLOWSE_STEP16(8, rb, rb2)
        }
        if (c2 + 1 < nchunks) {
            const int chunk = c2 + 1;
            FLOWSE_STEP16(0, rb2, rb) FLOWSE_STEP16(1, rb, rb2) FLOWSE_STEP16(2, rb2, rb) FLOWSE_STEP16(3, rb, rb2)
            FLOWSE_STEP16(4, rb2, rb) FLOWSE_STEP16(5, rb, rb2) FLOWSE_STEP16(6, rb2, rb) FLOWSE_STEP16(7, rb, rb2)
            FLOWSE_STEP16(8, rb2, rb)
        }
    }
#undef FLOWSE_STEP16
    conv_epilogue<2, 2, 2, 2, OT>(a, acc, smem, m_tl, n0, M, HW, 0, W);
}

// ---------------------------------------------------------------------------------------------------
// Flat-tiled 16-bit kernel for activations STORED as bf16 / half (BASELINE configs 3 / 5): the 1x1 shortcut
// convolutions and every 3x3 the halo kernel does not take (W < 16, split-K shapes).  128 flat pixels x 128 output
// channels per block, 4 waves x (2 x 2) tiles of v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation.  A K step is one
// (tap, 32-channel chunk) exactly as in the fp32 flat kernel -- same window descriptor, per-row tap masks and hardware
// zero fill -- but TWO steps are staged per barrier (a 16-bit MFMA phase is 16x shorter than an fp32 one) and both
// operands cross L2 -> LDS as 16-byte columns of eight channels.  Rows of 32 channels + 16 B pad (80 B) keep every
// fragment read a conflict-free ds_read_b128.  Weights: the packed [Cout][taps][Cin] matrix in the same 16-bit type
// (ConvArgs::wq).  gridDim.y = K slices (fp32 partial slabs + splitk_reduce, as for the fp32 kernel).
template <bool F16, class OT>
__global__ __launch_bounds__(256, 2) void conv_flat16_kernel(ConvArgs a) {
    using T16 = typename std::conditional<F16, f16_t, bf16_t>::type;
    constexpr int BM = 128, BN = 128, ROWB = 80, SLOT = 128 * ROWB;      // one (operand, step) tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);                              // [2 buffers][2 steps][BM][ROWB]
    char* Bs = As + 4 * SLOT;                                              // [2 buffers][2 steps][BN][ROWB]

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

    const int col = tid & 3, row0 = tid >> 2;                              // 16-byte column (8 channels), rows row0 + 64 q
    unsigned avo1[2], avo2[2], tapmask[2], bvo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = row0 + 64 * q;
        const int m = m0 + r;
        avo1[q] = (unsigned)(r * C1 + col * 8) * 2u;
        avo2[q] = (unsigned)(r * C2 + col * 8) * 2u;
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
        const int n = n0 + r;
        bvo[q] = n < a.Cout ? (unsigned)(n * taps * Cin + col * 8) * 2u : OOB;
    }
    const int64_t wbase = (int64_t)m0 - W - 1;
    const int wpix = BM + 2 * W + 2;
    const T16* in1p = reinterpret_cast<const T16*>(a.in1);
    const T16* in2p = reinterpret_cast<const T16*>(a.in2);
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T16*>(in1p + wbase * C1), 0, wpix * C1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T16*>(C2 ? in2p + wbase * C2 : in1p), 0, C2 ? wpix * C2 * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wq), 0, a.Cout * taps * Cin * 2, 0x00020000);

    const int S_all = (Cin / KC) * taps;                                   // K steps
    const int stages_all = (S_all + 1) >> 1;
    const int per = (stages_all + a.ksplit - 1) / a.ksplit;
    const int g_begin = split * per, g_end = min(stages_all, g_begin + per);

    u32x4 ra[2][2], rb[2][2];                                              // [step of the stage][row group]
    auto gload = [&](int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s = 2 * stage + j;
            const bool live = s < S_all;                                   // odd step count: the last slot is all zeros
            const int chunk = s / taps, tap = s - chunk * taps;
            int shift = W + 1;
            if (taps == 9) shift += (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
            const int c0 = chunk * KC;
            const bool second = c0 >= C1;
            const unsigned soff_a = (unsigned)(second ? shift * C2 + (c0 - C1) : shift * C1 + c0) * 2u;
            const unsigned soff_b = (unsigned)(tap * Cin + c0) * 2u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = live && ((tapmask[q] >> tap) & 1u);
                const unsigned vo = ok ? (second ? avo2[q] : avo1[q]) : OOB;
                ra[j][q] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo, live ? soff_a : 0u, 0)
                                  : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo, live ? soff_a : 0u, 0);
                rb[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, live ? bvo[q] : OOB, live ? soff_b : 0u, 0);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int off = (buf * 2 + j) * SLOT + (row0 + 64 * q) * ROWB + col * 16;
                *reinterpret_cast<u32x4*>(As + off) = ra[j][q];
                *reinterpret_cast<u32x4*>(Bs + off) = rb[j][q];
            }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(g_begin);
    lstore(0);
    __syncthreads();
    for (int g = g_begin; g < g_end; ++g) {
        const int buf = (g - g_begin) & 1;
        if (g + 1 < g_end) gload(g + 1);                                   // next stage in flight under the MFMAs
        const char* Ab = As + buf * 2 * SLOT + (wm * 64 + li) * ROWB + kh * 16;
        const char* Bb = Bs + buf * 2 * SLOT + (wn * 64 + li) * ROWB + kh * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const bf16x8*>(Ab + j * SLOT + i * 32 * ROWB + mh * 32);
                    bf[i] = *reinterpret_cast<const bf16x8*>(Bb + j * SLOT + i * 32 * ROWB + mh * 32);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        if (F16)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]),
                                                                               __builtin_bit_cast(f16x8, bf[jn]),
                                                                               acc[i][jn], 0, 0, 0);
                        else
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[jn], acc[i][jn], 0, 0, 0);
                    }
            }
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < g_end) lstore(buf ^ 1);
        __syncthreads();
    }
    conv_epilogue<2, 2, 2, 2, OT>(a, acc, smem, m0, n0, M, HW, split);
}

// 16-bit path policies.  The halo kernel takes a 3x3 when its tiling applies and yields at least ~one block per two
// CUs; everything else goes to the flat kernel, sliced along K (two-step stages, >= 2 stages per slice) until ~512
// blocks exist.
bool conv16_uses_halo(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (taps != 9 || (H & 7) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout % 128)) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    if ((int64_t)(9 * W + 18) * cmax * 4 >= (1LL << 31) || (int64_t)Cout * 9 * (C1 + C2) * 4 >= (1LL << 31)) return false;
    return ((int64_t)B * H * W / 128) * (Cout / 128) >= 128;
}

int conv16_ksplit(int B, int H, int W, int Cin, int Cout, int taps) {
    if (conv16_smallm_ok(B, H, W, Cin, 0, Cout, taps)) return 1;        // K is split inside the block (conv16_smallm.hip)
    const int64_t M = (int64_t)B * H * W;
    const int64_t tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
    const int stages = ((Cin / KC) * taps + 1) / 2;
    if (tiles >= 256 || stages < 4) return 1;
    int64_t ks = (512 + tiles - 1) / tiles;           // (swept in round 4: 128 ... 1024 blocks, 2 / 4 stages per slice: flat,
    if (ks > stages / 2) ks = stages / 2;             //  profiles/r04_policy_sweep.md)
    if (ks < 1) ks = 1;
    const int per = (int)((stages + ks - 1) / ks);
    return (int)((stages + per - 1) / per);
}

int conv16_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps) {
    const int HW = H * W;
    if (Cout & 3) return 0;
    if (taps == 9 && conv16_uses_halo(B, H, W, Cin, 0, Cout, taps)) return HW / 128;     // C1/C2 split is irrelevant here
    if (conv16_smallm_ok(B, H, W, Cin, 0, Cout, taps)) return conv16_smallm_stats_blocks(B, H, W);
    if (conv16_ksplit(B, H, W, Cin, Cout, taps) != 1) {
        const int PB = sk_pixels_per_block(HW);
        return ((HW % PB) == 0 && Cout / 4 <= 256) ? HW / PB : 0;
    }
    return (HW % 128) == 0 ? HW / 128 : 0;
}

int launch_flat16(const ConvArgs& a, hipStream_t s) {
    if ((a.C1 % KC) || (a.C2 % KC) || !a.wq || a.terms != 1 || a.in_dt == DT_F32 || (a.wq_f16 ? DT_F16 : DT_BF16) != a.in_dt ||
        (a.out_dt != DT_F32 && a.out_dt != a.in_dt) || a.gn.mean ||
        (int64_t)(128 + 2 * a.W + 2) * (a.C1 > a.C2 ? a.C1 : a.C2) * 2 >= (1LL << 31) ||
        (int64_t)a.Cout * a.taps * (a.C1 + a.C2) * 2 >= (1LL << 31)) {
        set_error("flat16: unsupported configuration (C1=%d C2=%d in_dt=%d out_dt=%d)", a.C1, a.C2, a.in_dt, a.out_dt);
        return ERR_ARG;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)((M + 127) / 128) * ((a.Cout + 127) / 128);
    const size_t lds = 8 * 128 * 80;                                       // > the epilogue's C tile + stats scratch
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
#define FLOWSE_L16(F16, OT)                                                                          \
    {                                                                                                \
        if (const int rc = allow_lds<&conv_flat16_kernel<F16, OT>>(lds)) return rc;                  \
        hipLaunchKernelGGL((conv_flat16_kernel<F16, OT>), dim3(grid, ks), dim3(256), lds, s, a);     \
    }
    if (a.wq_f16) {
        if (a.out_dt == DT_F32) FLOWSE_L16(true, float) else FLOWSE_L16(true, f16_t)
    } else {
        if (a.out_dt == DT_F32) FLOWSE_L16(false, float) else FLOWSE_L16(false, bf16_t)
    }
#undef FLOWSE_L16
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// elementwise storage conversion (any pair of types), 4 elements per thread
template <class SI, class SO>
__global__ __launch_bounds__(256) void convert_kernel(const SI* __restrict__ src, SO* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        St<SO>::st4(dst + 4 * i, St<SI>::ld4(src + 4 * i));
}

int launch_convert(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, hipStream_t s) {
    if (n & 3) {
        set_error("convert: element count must be a multiple of 4");
        return ERR_ARG;
    }
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    FLOWSE_DT_SWITCH(src_dt, SI, FLOWSE_DT_SWITCH(dst_dt, SO, hipLaunchKernelGGL((convert_kernel<SI, SO>), dim3((unsigned)blocks),
                                                                                 dim3(256), 0, s, static_cast<const SI*>(src),
                                                                                 static_cast<SO*>(dst), n4)));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

static unsigned short bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static unsigned short f16_rne(float f) {          // IEEE binary16, round to nearest even, overflow -> inf
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                 // >= 65520 rounds to inf
    if (u < 0x33000001u) return (unsigned short)sign;                              // < 2^-25 rounds to 0
    int e = (int)(u >> 23) - 127;
    unsigned m = (u & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? (-14 - e) + 13 : 13;                                    // subnormal halves shift further
    unsigned half_m = m >> shift;
    const unsigned rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) ++half_m;
    unsigned out = e < -14 ? half_m : (((unsigned)(e + 15) << 10) + (half_m - 0x400u));
    return (unsigned short)(sign | out);
}

void pack_conv_bf16(const float* w, int Cout, int Cin, int terms, uint16_t* dst, bool f16) {
    const int planes = terms == 1 ? 1 : 2, nchunks = Cin / KC;
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < 9; ++t)
            for (int ch = 0; ch < nchunks; ++ch) {
                uint16_t* row = dst + (((int64_t)co * 9 + t) * nchunks + ch) * planes * 32;
                for (int k = 0; k < 32; ++k) {
                    const float v = w[((int64_t)co * Cin + ch * 32 + k) * 9 + t];
                    if (f16) {
                        row[k] = f16_rne(v);
                        continue;
                    }
                    const unsigned short hi = bf16_rne(v);
                    row[k] = hi;
                    if (planes == 2) row[32 + k] = bf16_rne(v - bf16_to_f(hi));
                }
            }
}

bool conv_supports_bf16(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    return (Cout % 128) == 0 && conv_supports_fused_gn(B, H, W, C1, C2, Cout, taps) &&
           conv_ksplit(B, H, W, C1 + C2, Cout, taps) == 1;              // the 16-bit kernel has no split form
}

template <int TERMS, bool F16 = false>
static int launch_halo_bf16(const ConvArgs& a, hipStream_t s) {
    constexpr int PLANES = TERMS == 1 ? 1 : 2, ROWB = PLANES * 64 + 16;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)(M / 128) * (a.Cout / 128);
    const size_t lds_stage = (size_t)(180 + 2 * 128) * ROWB;
    const size_t lds_epi = ((size_t)128 * 132 + 256 * 8) * sizeof(float);
    const size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    if (a.in_dt != a.out_dt) {
        set_error("halo16: input and output storage types must agree");
        return ERR_ARG;
    }
    if (a.in_dt == DT_F32) {
        if (const int rc = allow_lds<&conv3x3_halo_bf16_kernel<TERMS, false, F16>>(lds)) return rc;
        if (const int rc = allow_lds<&conv3x3_halo_bf16_kernel<TERMS, true, F16>>(lds)) return rc;
        if (a.gn.mean)
            hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<TERMS, true, F16>), dim3(grid), dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<TERMS, false, F16>), dim3(grid), dim3(256), lds, s, a);
    } else {
        using T16 = typename std::conditional<F16, f16_t, bf16_t>::type;
        if (TERMS != 1 || a.in_dt != St<T16>::dt) {
            set_error("halo16: 16-bit storage needs the matching single-plane operand mode");
            return ERR_ARG;
        }
        if (const int rc = allow_lds<&conv3x3_halo_bf16_kernel<1, false, F16, T16, T16>>(lds)) return rc;
        if (const int rc = allow_lds<&conv3x3_halo_bf16_kernel<1, true, F16, T16, T16>>(lds)) return rc;
        if (a.gn.mean)
            hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<1, true, F16, T16, T16>), dim3(grid), dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<1, false, F16, T16, T16>), dim3(grid), dim3(256), lds, s, a);
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_halo_bf16x3(const ConvArgs& a, hipStream_t s) { return launch_halo_bf16<3>(a, s); }

// 16-bit operands (terms = 1), fp32 or 16-bit storage, on launches the producer / consumer kernel (conv16_pc.hip) does not
// take: fewer than 256 of its items, H not a multiple of 16, fp32 storage (networks whose channel counts do not tile)
int launch_halo16_any(const ConvArgs& a, hipStream_t s) {
    return a.wq_f16 ? launch_halo_bf16<1, true>(a, s) : launch_halo_bf16<1>(a, s);
}

}  // namespace flowse
