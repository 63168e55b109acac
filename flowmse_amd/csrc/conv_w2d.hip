// Replaces (reference): ddpm_conv3x3 (flowmse/backbones/ncsnpp_utils/layers.py:118-124) inside ResnetBlockBigGANpp
// (layerspp.py:245-274) incl. the GroupNorm + SiLU in front of it, the channel concat of ncsnpp.py:337, the per-sample
// time-embedding bias (layerspp.py:262-263), the (x + h)/sqrt(2) skip and the statistics of the NEXT GroupNorm -- the same
// contract as conv_f43.hip, evaluated in the TWO-dimensional Winograd form F(4,3) (vertical) x F(2,3) (horizontal).
#include "conv_common.h"

#ifdef FLOWSE_MEASURE_W2D
// measurement build only (tools/w2d_ts.py): 100 MHz time stamps of block 0 .. 255, wave 0: [block][64] slots
namespace flowse { __device__ unsigned long long g_w2d_ts[256 * 64]; }
extern "C" int flowse_debug_w2d_ts(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(flowse::g_w2d_ts), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#ifdef FLOWSE_MEASURE_W2D_WAVES
/* per-wave arrival at the slot barrier of slots 0..7: [block][wave 8][slot 8]; slot 0 of wave w also holds nothing else */
#define W2_TS(K) if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (K) >= 4 && ((K) - 4) % 3 == 0 && ((K) - 4) / 3 < 8) \
        flowse::g_w2d_ts[blockIdx.x * 64 + (threadIdx.x >> 6) * 8 + ((K) - 4) / 3] = __builtin_amdgcn_s_memrealtime();
#else
#define W2_TS(K) if (threadIdx.x == 0 && blockIdx.x < 256 && (K) < 64) flowse::g_w2d_ts[blockIdx.x * 64 + (K)] = __builtin_amdgcn_s_memrealtime();
#endif
#else
#define W2_TS(K)
#endif

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// F(4x2, 3x3): a 4 x 2 output patch from a 6 x 4 input patch through 6 x 4 = 24 products per channel pair, where the
// direct form spends 72 -- the matrix cores execute ONE THIRD of the direct-convolution FLOPs (F(4,3) alone: one half).
//   vertical   (as conv_f43.hip, points 0, +-1, +-2, inf): v = B4^T d over the six rows, out = A4^T m
//   horizontal (points 0, +-1, inf):  h0 = c0 - c2   h1 = c1 + c2   h2 = c2 - c1   h3 = c1 - c3
//              weights  w0 = g0   w1 = (g0 + g1 + g2)/2   w2 = (g0 - g1 + g2)/2   w3 = g2
//              outputs  x0 = m0 + m1 + m2               x1 = m1 - m2 - m3
// The horizontal transforms only use +-1 and 1/2, so the fp32 error stays at ~1.5x the 1-D form's.
//
// Block = 16 x 16 pixels x 64 output channels, EIGHT waves, one block per CU (the accumulators of a 2-D Winograd
// tile are 3x the outputs: 24 components x 32 units x 64 channels = 96 registers per lane over 512 lanes).  The tile is
// 4 x 8 = 32 UNITS of 4 x 2 pixels = one 32-row MFMA tile; wave (CH, h) owns the vertical half CH (components 0..2 from
// rows d0..d4, or 5, 3, 4 from d1..d5 -- exactly the roles of conv_f43.hip) of ONE horizontal component h for both
// 32-channel tiles: 3 x 2 accumulators.  Per 32-channel chunk the 18 x 18 halo is staged into LDS once (double buffered,
// GroupNorm + SiLU on the way in); a wave reads TWO columns per input row (the two pixels its h combines: c_a +- c_b, ten
// ds_read_b128 per 8-channel k-block), adds them, runs the vertical transform and issues 24 MFMAs; transformed weights
// arrive in fragment order from L2 as in conv_f43.hip.  There is no horizontal tap loop any more: a chunk is 4 phases.
//
// LDS halo: [18 rows][18 pixels][32 ch + 4 pad], row pitch 656 floats (4 rows = 0 mod 64 banks).  A lane's unit is
// (ur, uc) = (li >> 3, li & 7): sixteen consecutive lanes = two unit rows x eight unit columns at a pixel pitch of 72
// floats -> bank 8 uc for both rows.  The 16-byte channel quad q of halo row hy is therefore stored at position
// q ^ ((hy >> 2) & 1): the two unit rows of a 16-lane group land 4 banks apart and every ds_read_b128 is conflict free.
constexpr int W2_HROW = 18 * LDS_ROW + 8;
constexpr int W2_HBUF = 18 * W2_HROW;                    // floats per halo buffer (47 232 bytes)
constexpr int W2_XDEST = 4 * 3 * 256;                    // floats per destination wave in the output exchange (12 KB)

struct W2Tile {
    int y0, x0;
    unsigned hin;                                        // bit q: this thread's halo quad q lies inside the image
    unsigned woff;                                       // pixel offset of the tile's window inside the sample's descriptor
};

// ---- output stage.  out = A4^T [m] A2 needs all 24 components of a (unit, channel); they live in eight waves.
// Destination wave D = (j, g) finishes the 32 channels of tile j for unit row g (4 image rows x 16 pixels): accumulator
// registers r = 4g .. 4g+3 of every source.  Sources pre-reduce vertically inside their half --
//     CH 0: (m0 + m1 + m2, m1 - m2, m1 + m2)          CH 1: (m3 + m4, m3 - m4, m5)
// -- and hand each destination three float4 (the four registers of its unit row).  Two rounds through ONE 96 KB region
// [dest 8][h 4][3][lane 64][4] (the CH 0 sources, then the CH 1 sources: 2 x 96 KB would not fit beside the next tile's
// prefetched halo); a destination folds the four h of a round in a fixed order (x0 = (h0 + h1) + h2, x1 = (h1 - h2) - h3:
// bit-reproducible) and finishes
//     o0 = A + D      o1 = B + 2 E      o2 = C + 4 D      o3 = B + 8 E + F          (A..F = the six partials, per column)
// Then, wave-private: transposition of its 64 pixels x 32 channels through its own 12 KB slice, bias / per-sample bias /
// residual / scale, 16-byte stores, GroupNorm partial statistics per (4 x 16 pixel strip, channel).
// NJ = 1 (32-channel blocks, see the kernel): only waves 0-3 are destinations; the others give and keep the barriers.
template <int CH, int NJ>
__device__ __forceinline__ void w2d_out(f32x16 (&acc)[3][NJ], float* X, int b, int y0, int x0, int n0) {
    // The arguments of the output stage are read from the kernel-argument segment HERE (scalar loads, once per tile): kept
    // live across the main loop they were ~20 SGPRs of a kernel that was spilling 58 of them to VGPR lanes
    const ConvArgs* ap = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();   // (C cast: constant -> generic address space)
    asm volatile("" : "+s"(ap));
    const ConvArgs& a = *ap;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));                       // keep lane-only address arithmetic out of the caller's main loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hq = wave & 3;
    const int jD = wave >> 2, gD = wave & 3;             // this wave as a destination
    const bool dest = jD < NJ;
    const int li = lane & 31, kh = lane >> 5;
    const int W = a.W, Cout = a.Cout;
    const int ch0 = n0 + jD * 32;
    const int pl = lane >> 3, cq = lane & 7;             // row pass: pixel lane, channel quad
    const bool has_res = a.res != nullptr;
    const int64_t pix0 = ((int64_t)b * a.H + y0 + 4 * gD) * W + x0;
    const float* resb = a.res + pix0 * Cout + ch0 + cq * 4;
    float* outb = a.out + pix0 * Cout + ch0 + cq * 4;
    int roff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pp = i * 8 + pl;                       // pixel of the 4 x 16 strip, row-major
        roff[i] = ((pp >> 4) * W + (pp & 15)) * Cout;
    }
    // residual quads are requested first: in flight during the whole exchange
    float4 rres[8];
    if (has_res && dest) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rres[i] = *reinterpret_cast<const float4*>(resb + roff[i]);
    }
    float* Xd = X + wave * W2_XDEST;
    auto give = [&]() {
#pragma unroll
        for (int D = 0; D < 4 * NJ; ++D) {
            const int jd = D >> 2, gd = D & 3;
            float4 v0, v1, v2;
            float* e0 = &v0.x; float* e1 = &v1.x; float* e2 = &v2.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * gd + e;
                const float s12 = acc[1][jd][r] + acc[2][jd][r];
                if (CH == 0) {
                    e0[e] = acc[0][jd][r] + s12;
                    e1[e] = acc[1][jd][r] - acc[2][jd][r];
                    e2[e] = s12;
                } else {                                 // acc[0] = m5, acc[1] = m3, acc[2] = m4
                    e0[e] = s12;
                    e1[e] = acc[1][jd][r] - acc[2][jd][r];
                    e2[e] = acc[0][jd][r];
                }
            }
            float* dst = X + D * W2_XDEST + (hq * 3) * 256 + lane * 4;
            *reinterpret_cast<float4*>(dst) = v0;
            *reinterpret_cast<float4*>(dst + 256) = v1;
            *reinterpret_cast<float4*>(dst + 512) = v2;
        }
    };
    // S[x][c][e]: column x, partial c, register e of this wave's unit row
    auto take = [&](float (&S)[2][3][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(Xd + (s * 3 + c) * 256 + lane * 4);
                const float* q = &t.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (s == 0) S[0][c][e] = q[e];
                    if (s == 1) { S[0][c][e] += q[e]; S[1][c][e] = q[e]; }
                    if (s == 2) { S[0][c][e] += q[e]; S[1][c][e] -= q[e]; }
                    if (s == 3) S[1][c][e] -= q[e];
                }
            }
        }
    };
    float SA[2][3][4], SB[2][3][4];
    if (CH == 0) give();
    __syncthreads();
    if (dest) take(SA);
    __syncthreads();
    if (CH == 1) give();
    __syncthreads();
    if (!dest) return;
    // (bias quads: L2 hits, requested here -- under the second round's reads and sums -- not before the exchange, where
    // eight more live registers spill)
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), bq2 = bq;
    if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + ch0 + cq * 4);
    if (a.bias2) bq2 = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + ch0 + cq * 4);
    take(SB);
    // ---- transpose through this wave's own slice (nobody else reads it), finish, store, statistics
    float* T = Xd;                                       // [64 pixels][32 channels]
    __builtin_amdgcn_wave_barrier();                     // LDS is in-order per wave: take()'s reads precede these writes
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float A = SA[x][0][e], Bv = SA[x][1][e], Cv = SA[x][2][e];
            const float D = SB[x][0][e], E = SB[x][1][e], Fv = SB[x][2][e];
            const float o0 = A + D, o1 = fmaf(2.f, E, Bv), o2 = fmaf(4.f, D, Cv), o3 = fmaf(8.f, E, Bv) + Fv;
            const int col = 2 * (e + 4 * kh) + x;        // unit column uc = e + 4 kh
            T[(0 * 16 + col) * 32 + li] = o0;
            T[(1 * 16 + col) * 32 + li] = o1;
            T[(2 * 16 + col) * 32 + li] = o2;
            T[(3 * 16 + col) * 32 + li] = o3;
        }
    __builtin_amdgcn_wave_barrier();
    bq.x += bq2.x; bq.y += bq2.y; bq.z += bq2.z; bq.w += bq2.w;
    const float scale = a.scale;
    float4 piv = make_float4(0.f, 0.f, 0.f, 0.f), s1 = piv, s2 = piv;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pp = i * 8 + pl;
        float4 v = *reinterpret_cast<const float4*>(T + pp * 32 + cq * 4);
        v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
        if (has_res) { v.x += rres[i].x; v.y += rres[i].y; v.z += rres[i].z; v.w += rres[i].w; }
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(outb + roff[i]) = v;
        if (i == 0) piv = v;
        const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
    }
    if (!a.stats) return;
    // 8 values per lane and channel -> the 8 pixel lanes of a channel quad (equal-count Chan merges) -> 64 pixels
    float mean[4] = {piv.x + s1.x * 0.125f, piv.y + s1.y * 0.125f, piv.z + s1.z * 0.125f, piv.w + s1.w * 0.125f};
    float m2[4] = {fmaxf(s2.x - s1.x * s1.x * 0.125f, 0.f), fmaxf(s2.y - s1.y * s1.y * 0.125f, 0.f),
                   fmaxf(s2.z - s1.z * s1.z * 0.125f, 0.f), fmaxf(s2.w - s1.w * s1.w * 0.125f, 0.f)};
    float cnt = 8.f;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mo = __shfl_xor(mean[j], off), qo = __shfl_xor(m2[j], off);
            const float d = mo - mean[j];
            m2[j] = m2[j] + qo + d * d * (0.5f * cnt);
            mean[j] = 0.5f * (mean[j] + mo);
        }
        cnt *= 2.f;
    }
    if (pl == 0) {
        // statistics blocks of this kernel: 4 x 16 pixel strips, row-major over the sample (stats_nblk = H W / 64)
        const int strip = ((y0 >> 2) + gD) * (W >> 4) + (x0 >> 4);
        float* dst = a.stats + (((int64_t)b * a.stats_nblk + strip) * Cout + ch0 + cq * 4) * 2;
        *reinterpret_cast<float4*>(dst) = make_float4(mean[0], m2[0], mean[1], m2[1]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
    }
}

template <int GN, int CH, int NJ>
__device__ __forceinline__ void conv3x3_w2d_body(const ConvArgs& a, float* smem, int tpb) {
    constexpr int BN = 32 * NJ;                          // output channels per block
    constexpr int H_LOADS = 6;                           // halo quads per thread and chunk
    float* Hs = smem;                                    // [2][18][W2_HROW]

    const int tid = threadIdx.x;
    W2_TS(0)
    const int H = a.H, W = a.W, HW = H * W;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // a block owns `tpb` consecutive pixel tiles (in walk order) of ONE 64-channel block; the channel blocks of a pixel
    // tile are neighbours in launch order (same XCD, same time: the second one finds the input in L2)
    const int mg = bid / n_ntiles;
    const int nt = bid - mg * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 4);
    const int n0 = nt * BN;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Staging map: thread = (pixel slot p = tid >> 3, channel quad tid & 7).  The 18 x 18 halo is walked in SIX rounds of
    // three COLUMNS x 18 rows = 54 pixels (slots 54..63 idle): slot p holds row p % 18 of column p / 18 + 3 q in round q.
    // Its row -- and with it the LDS swizzle bit -- is the same in every round, so the global offset and the LDS offset
    // are one register each and a round only adds scalars (3 pixels / 3 LDS pixel pitches): no per-quad address tables.
    const int col4 = tid & 7, slot = tid >> 3;
    const bool s_on = slot < 54;
    const int s_hx = slot >= 36 ? 2 : slot >= 18 ? 1 : 0, s_hy = slot - 18 * s_hx;
    const unsigned s_pix = (unsigned)(s_hy * W + s_hx);  // pixel offset of round 0 inside the window
    const int s_lds = s_hy * W2_HROW + s_hx * LDS_ROW + 4 * (col4 ^ ((s_hy >> 2) & 1));
    int s_st = s_lds + W2_HBUF;                          // staging store base: the buffer NOT being read
    int dflip = W2_HBUF;                                 // read bases += dflip, store base -= dflip at every slot barrier
    // tiles of an image are walked in vertical strips of 4 tiles (64 pixels), top to bottom (as conv_f43.hip)
    const int bsmp = (mg * tpb) / tiles_img;
    const int b = bsmp;
    const int64_t sbase = (int64_t)bsmp * HW - W - 1;
    const int spix = HW + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + sbase * C1), 0, spix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + sbase * C2 : a.in1), 0, C2 ? spix * C2 * 4 : 0, 0x00020000);
    auto make_tile = [&](int mt) {
        W2Tile t;
        const int tt = mt - bsmp * tiles_img;
        int ty, tx;
        if ((tiles_x & 3) == 0) {
            const int per_strip = 4 * (H >> 4);
            const int strip = tt / per_strip, w = tt - strip * per_strip;
            ty = w >> 2;
            tx = strip * 4 + (w & 3);
        } else {
            ty = tt / tiles_x;
            tx = tt - ty * tiles_x;
        }
        t.y0 = ty * 16;
        t.x0 = tx * 16;
        const bool rowin = s_on && (unsigned)(t.y0 - 1 + s_hy) < (unsigned)H;
        t.hin = 0;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q)
            t.hin |= (rowin && (unsigned)(t.x0 - 1 + s_hx + 3 * q) < (unsigned)W) ? (1u << q) : 0u;
        t.woff = (unsigned)(t.y0 * W + t.x0);
        return t;
    };
    const W2Tile first = make_tile(mg * tpb);
    int cur_y0 = first.y0, cur_x0 = first.x0;            // the tile being computed (the staged one runs ahead: stg)
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wino2), 0, a.Cout * 24 * Cin * 4, 0x00020000);

    // The halo of the next chunk is staged in two halves of three quads (request -> GroupNorm/SiLU in registers -> LDS
    // write into the idle buffer): 12 staging registers live at any time
    u32x4 rh[3];
    float4 g_mu, g_sc, g_be;
    unsigned st_hin = first.hin;                          // halo mask of the tile being STAGED

    auto hload = [&](const W2Tile& t, int chunk, int Q) -> u32x4 {
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned cs = (unsigned)(second ? C2 : C1);
        const unsigned soff = ((t.woff + 3u * (unsigned)Q) * cs + (unsigned)(second ? c0 - C1 : c0)) * 4u;
        const unsigned off = ((t.hin >> Q) & 1u) ? (s_pix * cs + (unsigned)col4 * 4u) * 4u : OOB;
        return second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, soff, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, soff, 0);
    };
    // Zero padding applies AFTER the activation.  Rows outside the image: the thread's scale and shift are zeroed for the
    // whole chunk (its requests return 0, (0 - mean) * 0 + 0 = 0 and SiLU(0) = 0) -- `hin` bit 1 = columns 3..5 of the
    // halo, always inside horizontally, i.e. "this thread's ROW is inside".  Columns outside the image only occur in the
    // first and the last round (halo columns 0 and 17): only those two quads carry a select.
    auto gparams = [&](int chunk, unsigned hin) {
        if (GN) {
            const int cg = chunk * KC + col4 * 4;
            const float rowf = (hin >> 1) & 1u ? 1.f : 0.f;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
            g_sc.x *= rowf; g_sc.y *= rowf; g_sc.z *= rowf; g_sc.w *= rowf;
            g_be.x *= rowf; g_be.y *= rowf; g_be.z *= rowf; g_be.w *= rowf;
        }
    };
    auto gloadH = [&](const W2Tile& t, int chunk, int h) {
#pragma unroll
        for (int q = 0; q < 3; ++q) rh[q] = hload(t, chunk, 3 * h + q);
        if (h == 0) {
            gparams(chunk, t.hin);
            st_hin = t.hin;
        }
    };
    auto xform1 = [&](int Q) {
        if (GN) rh[Q % 3] = gn_quad<GN>(rh[Q % 3], g_mu, g_sc, g_be, (Q == 0 || Q == 5) ? ((st_hin >> Q) & 1u) != 0 : true);
    };
    auto lstoreH = [&](int h) {
        float* Hb = Hs + s_st + 9 * h * LDS_ROW;
        if (s_on) {
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(Hb + 3 * q * LDS_ROW) = rh[q];
        }
    };

    const int lane = tid & 63;
    const int hq = wave & 3;                             // horizontal component of this wave; CH = wave >> 2 (template)
    const int li = lane & 31, kh = lane >> 5;
    const int ur = li >> 3, uc = li & 7;                 // this lane's unit: rows 4 ur.., columns 2 uc..
    // h0 = c0 - c2, h1 = c1 + c2, h2 = c2 - c1, h3 = c1 - c3: columns (ca, cb) and the sign of cb
    const int ca = hq == 0 ? 0 : hq == 2 ? 2 : 1, cb = hq == 2 ? 1 : hq == 3 ? 3 : 2;
    f32x2 sg2 = {hq == 1 ? 1.f : -1.f, hq == 1 ? 1.f : -1.f};
    asm volatile("" : "+v"(sg2));                        // a VGPR pair: the packed fma takes no scalar pair here
    // input rows start at halo row 4 ur (+1 for CH = 1); quad position (2 J + kh) ^ ((hy >> 2) & 1), hy = 4 ur + CH + r:
    // bit = (ur + ((CH + r) >> 2)) & 1 -- two address variants per column (rows without / with the carry)
    const int base = (4 * ur + CH) * W2_HROW + 2 * uc * LDS_ROW;
    const int k0 = 4 * (kh ^ (ur & 1)), k1 = 4 * (kh ^ (ur & 1) ^ 1);
    int aA0 = base + ca * LDS_ROW + k0, aA1 = base + ca * LDS_ROW + k1;
    int aB0 = base + cb * LDS_ROW + k0, aB1 = base + cb * LDS_ROW + k1;

    // weight fragments: 24 KB per (32-channel slice, h, chunk), [component 0..5][k-block][lane][4 floats]
    const int nchunks = Cin / KC;
    const unsigned wslice = (unsigned)((n0 >> 5) * 4 + hq) * (unsigned)nchunks;     // in 24 KB units; tile j adds 4 nchunks
    const unsigned wvo = (unsigned)lane * 16u + (unsigned)CH * 3u * 4096u;
    unsigned wvoc[3] = {wvo, wvo + 4096u, wvo + 8192u};
    asm volatile("" : "+v"(wvoc[0]), "+v"(wvoc[1]), "+v"(wvoc[2]));     // three registers, not an add per request
    const unsigned wj = 4u * (unsigned)nchunks * 24576u;                 // channel tile 1

    f32x16 acc[3][NJ];                                   // this wave's three vertical components x NJ channel tiles

    {   // first chunk of the block's first tile: all quads at once (the accumulators are not live yet)
        u32x4 t[H_LOADS];
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) t[q] = hload(first, 0, q);
        gparams(0, first.hin);
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            if (GN) t[q] = gn_quad<GN>(t[q], g_mu, g_sc, g_be, (q == 0 || q == 5) ? ((first.hin >> q) & 1u) != 0 : true);
            if (s_on) *reinterpret_cast<u32x4*>(Hs + s_lds + 3 * q * LDS_ROW) = t[q];
        }
    }
    __syncthreads();
    W2_TS(1)

#define W2_FENCE __builtin_amdgcn_sched_barrier(0);
    // (the four base registers INCLUDE the offset of the buffer being read and are flipped at the slot barrier: every
    // LDS read is base register + immediate; buffer-specific copies of the bases were being spilled)
#define W2_RD(BASE0, BASE1, R, J) \
    (*reinterpret_cast<const float4*>(Hs + ((CH + (R)) >= 4 ? (BASE1) : (BASE0)) + (R) * W2_HROW + (J) * 8))
    // weight fragment (component C, channel tile JT, k-block J) of the chunk whose per-tile scalar offsets are WB[0..1]:
    // voffset = one register per component, k-block in the 12-bit immediate, everything else in the scalar offset
#define W2_BLOAD(C, JT, J, WB, BF)                                                                                   \
    {                                                                                                                \
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, wvoc[C] + (J) * 1024, WB[JT], 0);               \
        BF[C][JT] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z),                    \
                                __uint_as_float(t.w));                                                               \
    }
#define W2_H2(Q, H) (*reinterpret_cast<f32x2*>(&(Q).x + 2 * (H)))
    // horizontal combination, in place: D[r] = ca + sg cb
#define W2_COMB(D, T, R, TR)                                                                                             \
    {                                                                                                                \
        W2_H2(D[R], 0) = __builtin_elementwise_fma(sg2, W2_H2(T[TR], 0), W2_H2(D[R], 0));                            \
        W2_H2(D[R], 1) = __builtin_elementwise_fma(sg2, W2_H2(T[TR], 1), W2_H2(D[R], 1));                            \
    }
    // vertical input transform, in place (as conv_f43.hip), per float2 half h: D[0..2] (CH 0) / D[4], D[1], D[2] (CH 1)
    // become the operands
#define W2_WXA_H(D, h)                                                                                               \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f};                                                             \
        const f32x2 r0 = W2_H2(D[0], h), r2 = W2_H2(D[2], h), r4 = W2_H2(D[4], h);                                   \
        const f32x2 v = __builtin_elementwise_fma(c4, r0, __builtin_elementwise_fma(cm5, r2, r4));                   \
        if (CH == 0) W2_H2(D[0], h) = v;                                                                             \
        else W2_H2(D[4], h) = v;                                                                                     \
    }
#define W2_WXB_H(D, h)                                                                                               \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm4 = {-4.f, -4.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};                        \
        const f32x2 r0 = W2_H2(D[0], h), r1 = W2_H2(D[1], h), r2 = W2_H2(D[2], h), r3 = W2_H2(D[3], h),              \
                    r4 = W2_H2(D[4], h);                                                                             \
        if (CH == 0) {                                                                                               \
            W2_H2(D[1], h) = __builtin_elementwise_fma(cm4, r1 + r2, r3 + r4);                                       \
            W2_H2(D[2], h) = __builtin_elementwise_fma(c4, r1 - r2, r4 - r3);                                        \
        } else {                                                                                                     \
            W2_H2(D[1], h) = __builtin_elementwise_fma(c2, r2 - r0, r3 - r1);                                        \
            W2_H2(D[2], h) = __builtin_elementwise_fma(cm2, r2 - r0, r3 - r1);                                       \
        }                                                                                                            \
    }
#define W2_WXA(D) { W2_WXA_H(D, 0) W2_WXA_H(D, 1) }
#define W2_WXB(D) { W2_WXB_H(D, 0) W2_WXB_H(D, 1) }
    // One k-block (8 channels per lane half: four MFMA k-steps) = 24 MFMAs in COMPONENT-MAJOR order: the eight MFMAs of
    // component c (k-steps x..w, the two channel tiles alternating -- never two MFMAs in a row on one accumulator) run
    // before component c + 1 starts.  The operands of a component are therefore dead after its eighth MFMA and the NEXT
    // k-block's weight fragments are requested straight into the same registers, a full k-block (16 MFMAs) before their
    // use: ONE set of 24 weight registers instead of two -- the two-set form sat at the 256-register cap and reloaded
    // spilled addresses through the in-order vector-memory queue.  Every other instruction rides in an MFMA gap: the ten
    // LDS reads of the next k-block and their horizontal combinations behind component 0, the vertical transform behind
    // component 1, one staged GroupNorm quad (SX) behind component 2.
#define W2_MF(V, c, K)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                   \
        acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[((c) == 0 && CH == 1) ? 4 : (c)].K, bF[c][j].K, acc[c][j], 0, 0, 0); \
    W2_FENCE
#define W2_BL(C, NJB, WB) _Pragma("unroll") for (int j = 0; j < NJ; ++j) W2_BLOAD(C, j, NJB, WB, bF)
    // twelve (component, k-step) pairs of NJ MFMAs, one filler group behind each
#define W2_PHASE(V, DN, NJB, WB, SX)                                                                                 \
    W2_MF(V, 0, x) DN[0] = W2_RD(aA0, aA1, 0, NJB); tN[0] = W2_RD(aB0, aB1, 0, NJB);                                 \
                   DN[1] = W2_RD(aA0, aA1, 1, NJB); tN[1] = W2_RD(aB0, aB1, 1, NJB); W2_FENCE                        \
    W2_MF(V, 0, y) DN[2] = W2_RD(aA0, aA1, 2, NJB); tN[2] = W2_RD(aB0, aB1, 2, NJB); W2_COMB(DN, tN, 0, 0) W2_FENCE  \
    W2_MF(V, 0, z) DN[3] = W2_RD(aA0, aA1, 3, NJB); tN[0] = W2_RD(aB0, aB1, 3, NJB); W2_COMB(DN, tN, 1, 1) W2_FENCE  \
    W2_MF(V, 0, w) DN[4] = W2_RD(aA0, aA1, 4, NJB); tN[1] = W2_RD(aB0, aB1, 4, NJB); W2_COMB(DN, tN, 2, 2) W2_FENCE  \
    W2_MF(V, 1, x) W2_BL(0, NJB, WB) W2_COMB(DN, tN, 3, 0) W2_FENCE                                                  \
    W2_MF(V, 1, y) W2_COMB(DN, tN, 4, 1) W2_WXA_H(DN, 0) W2_FENCE                                                    \
    W2_MF(V, 1, z) W2_WXA_H(DN, 1) W2_WXB_H(DN, 0) W2_FENCE                                                          \
    W2_MF(V, 1, w) W2_WXB_H(DN, 1) W2_FENCE                                                                          \
    W2_MF(V, 2, x) W2_BL(1, NJB, WB) W2_FENCE                                                                        \
    W2_MF(V, 2, y) SX W2_FENCE                                                                                       \
    W2_MF(V, 2, z)                                                                                                   \
    W2_MF(V, 2, w) W2_BL(2, NJB, WB) W2_FENCE

    float4 dA[5], dB[5], tN[3], bF[3][NJ];
    // operands of a tile's first k-block: requests (LDS rows + weights) and, later, combination + transform.  The block's
    // first tile issues both back to back; at a tile boundary the requests go out inside the output stage (before its
    // stores) and the rest follows it.
    auto start_issue = [&]() {
#pragma unroll
        for (int r = 0; r < 5; ++r) dA[r] = W2_RD(aA0, aA1, r, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) tN[r] = W2_RD(aB0, aB1, r, 0);
        const unsigned wb[2] = {wslice * 24576u, wslice * 24576u + wj};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j) W2_BLOAD(c, j, 0, wb, bF)
    };
    auto start_finish = [&]() {
#pragma unroll
        for (int r = 0; r < 3; ++r) W2_COMB(dA, tN, r, r)
#pragma unroll
        for (int r = 3; r < 5; ++r) tN[r - 3] = W2_RD(aB0, aB1, r, 0);
#pragma unroll
        for (int r = 3; r < 5; ++r) W2_COMB(dA, tN, r, r - 3)
        W2_WXA(dA) W2_WXB(dA)
    };
    // ---- The block's work is a stream of SLOTS (tile ti, chunk): slot s lives in halo buffer s & 1.  While slot s is
    // computed, slot s + 1 is staged (first half requested during slot s - 1's last phase, normalised and stored in
    // phase 0; second half requested in phase 0, stored in phase 2) and ONE barrier sits at the end of phase 2: from there
    // on buffer s & 1 is not read any more (phase 3 runs on operands read in phase 2) and buffer (s + 1) & 1 is complete,
    // so phase 3 already reads and transforms slot s + 1's first operands behind its own MFMAs -- no wave ever starts a
    // chunk with an empty matrix pipe.  The stream runs across tile boundaries; only the output stage interrupts it
    // (its exchange region lies behind buffer 0 = the next tile's first chunk: tpb > 1 needs an even chunk count).
    const int nslots = tpb * nchunks;
    int sg_t = 0, sg_c = 0;                              // the slot being staged (tile within the block, chunk)
    W2Tile stg = first;
    auto stage_advance = [&]() {
        if (++sg_c == nchunks) {
            sg_c = 0;
            ++sg_t;
            if (sg_t < tpb) stg = make_tile(mg * tpb + sg_t);
        }
        if (sg_t >= tpb) {                               // past the block's last slot: a harmless repeat of its last chunk
            sg_t = tpb;
            sg_c = nchunks - 1;
        }
    };
    stage_advance();                                     // slot 1
    gloadH(stg, sg_c, 0);
    start_issue();
    start_finish();
    int ti = 0, chunk = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    for (int slot = 0; slot < nslots; ++slot) {
        const bool last = chunk + 1 == nchunks;          // the tile's last chunk
        const unsigned wb[2] = {(wslice + (unsigned)chunk) * 24576u, (wslice + (unsigned)chunk) * 24576u + wj};
        const unsigned wbn0 = last ? wslice * 24576u : wb[0] + 24576u;       // slot + 1's chunk
        const unsigned wbn[2] = {wbn0, wbn0 + wj};
        // phase 0: the first half of slot + 1's halo is normalised (one quad per phase would leave the stores too late:
        // two ride here, the third opens phase 1) and stored, its second half requested
        W2_PHASE(dA, dB, 1, wb, { xform1(0); xform1(1); })
        // phase 1
        W2_PHASE(dB, dA, 2, wb, { xform1(2); lstoreH(0); gloadH(stg, sg_c, 1); })
        // phase 2: the second half, then THE barrier of the slot
        W2_PHASE(dA, dB, 3, wb, { xform1(3); xform1(4); })
        xform1(5);
        lstoreH(1);
        W2_TS(4 + 3 * slot)
        __syncthreads();                                 // slot + 1's halo is complete; nobody reads this slot's any more
        W2_TS(5 + 3 * slot)
        aA0 += dflip; aA1 += dflip; aB0 += dflip; aB1 += dflip;
        s_st -= dflip;
        dflip = -dflip;
        asm volatile("" : "+v"(aA0), "+v"(aA1), "+v"(aB0), "+v"(aB1), "+v"(s_st));   // (five registers, not ten)
        // phase 3: slot + 1's k-block 0 from the other buffer; slot + 2's first half is requested
        stage_advance();
        W2_PHASE(dB, dA, 0, wbn, { gloadH(stg, sg_c, 0); })
        W2_TS(6 + 3 * slot)
        if (last) {
            const bool more = ti + 1 < tpb;
            w2d_out<CH, NJ>(acc, smem + W2_HBUF, b, cur_y0, cur_x0, n0);
            W2_TS(2 + (ti & 1))
            if (!more) return;
            __syncthreads();                             // buffer 1 (under the exchange region) is written again in the next tile
            // (the operands phase 3 fetched for the next tile were not kept across the output stage: 36 registers.  Requesting
            // them inside the stage, ahead of its stores, and moving this barrier into the next tile's first phase both
            // measured equal: profiles/r05_w2d_probes.md)
            start_issue();
            start_finish();
            {
                const W2Tile nx = make_tile(mg * tpb + ti + 1);
                cur_y0 = nx.y0;
                cur_x0 = nx.x0;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
            ++ti;
            chunk = 0;
        } else {
            ++chunk;
        }
    }
#undef W2_FENCE
#undef W2_RD
#undef W2_BLOAD
#undef W2_H2
#undef W2_COMB
#undef W2_WXA
#undef W2_WXB
#undef W2_WXA_H
#undef W2_WXB_H
#undef W2_MF
#undef W2_BL
#undef W2_PHASE
}

// NJ = 2: 64 output channels per block (two 32-channel tiles per wave).  NJ = 1: 32 channels per block -- twice the blocks
// for launches that would leave CUs idle (the 32 x 32 level at batch 8: 128 blocks of 64 channels, 256 of 32): every staged
// halo element and every transformed operand feeds half the MFMAs, but all 256 CUs work with two waves per SIMD, where the
// 1-D kernel's 64-channel form runs one 4-wave block per CU.
template <int GN, int NJ>
__global__ __launch_bounds__(512, 2) void conv3x3_w2d_kernel(ConvArgs a, int tpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // waves 0..3: vertical half 0; waves 4..7: half 1.  Both bodies execute the same barriers.
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) conv3x3_w2d_body<GN, 1, NJ>(a, smem, tpb);
    else conv3x3_w2d_body<GN, 0, NJ>(a, smem, tpb);
}

// [Cout][9][Cin] -> fragment order [Cout/32][h 0..3][Cin/32][component 0..5][k-block j][lane][4]; stored vertical component
// order as conv_f43.hip: 0,1,2 = u0,u1,u2 (wave half 0), 3,4,5 = u5,u3,u4 (wave half 1)
__global__ __launch_bounds__(256) void w2d_weights_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                          float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (n, ci)
    if (idx >= (int64_t)Cout * Cin) return;
    const int ci = (int)(idx % Cin);
    const int64_t n = idx / Cin;
    float U[6][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const float g0 = w[(n * 9 + 0 + kx) * Cin + ci], g1 = w[(n * 9 + 3 + kx) * Cin + ci],
                    g2 = w[(n * 9 + 6 + kx) * Cin + ci];
        const float s02 = g0 + g2;
        U[0][kx] = 0.25f * g0;
        U[1][kx] = (s02 + g1) * (-1.f / 6.f);
        U[2][kx] = (s02 - g1) * (-1.f / 6.f);
        const float t = fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f)), h = g1 * (1.f / 12.f);
        U[3][kx] = g2;           // u5
        U[4][kx] = t + h;        // u3
        U[5][kx] = t - h;        // u4
    }
    const int nchunks = Cin >> 5;
    const int chunk = ci >> 5, j = (ci >> 3) & 3, kh = (ci >> 2) & 1, e = ci & 3;
    const int lane = kh * 32 + (int)(n & 31);
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        const float s02 = U[v][0] + U[v][2];
        const float wh[4] = {U[v][0], 0.5f * (s02 + U[v][1]), 0.5f * (s02 - U[v][1]), U[v][2]};
#pragma unroll
        for (int h = 0; h < 4; ++h)
            out[(((((n >> 5) * 4 + h) * nchunks + chunk) * 6 + v) * 4 + j) * 256 + lane * 4 + e] = wh[h];
    }
}

int launch_w2d_weights(const float* w_packed, int Cout, int Cin, float* out, hipStream_t s) {
    if ((Cout % 32) != 0 || (Cin % 32) != 0) {
        set_error("w2d_weights: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
        return ERR_SHAPE;
    }
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(w2d_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_packed, Cout, Cin, out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// tiles per block: as many (16, 8, 4, 2) as still leave FLOWSE_W2D_TPB_BLOCKS = 256 blocks -- ONE per CU, which is all that is
// resident at a time (A-B-A-B on one box: 256 -> 20.81 k frames/s, 512 -> 20.62 k, 1024 -> 20.38 k) -- so that the prologue is
// paid once per block and the staging pipeline runs across tile boundaries; needs an even chunk count and must divide the
// tiles of an image
static const int g_w2d_tpb_blocks = getenv("FLOWSE_W2D_TPB_BLOCKS") ? atoi(getenv("FLOWSE_W2D_TPB_BLOCKS")) : 256;
// channel tiles per block: 2 (64 channels) when that gives every CU a block, else 1 (32 channels: twice the blocks).  Measured
// at 256 / 384 blocks of 64 channels: 76 vs 91 us and 144 vs 172 us for the 64-channel form (profiles/r05_w2d_probes.md)
static const int g_w2d_nj2_blocks = getenv("FLOWSE_W2D_NJ2_BLOCKS") ? atoi(getenv("FLOWSE_W2D_NJ2_BLOCKS")) : 256;
int w2d_channel_tiles(int B, int H, int W, int Cout) {
    return ((int64_t)B * H * W / 256) * (Cout / 64) >= g_w2d_nj2_blocks ? 2 : 1;
}
int w2d_tiles_per_block(int B, int H, int W, int Cin, int Cout) {
    if (((Cin / KC) & 1) != 0) return 1;
    const int64_t blocks1 = ((int64_t)B * H * W / 256) * (Cout / (32 * w2d_channel_tiles(B, H, W, Cout)));
    // One block per CU is resident at a time, so the blocks run in rounds of 256: the largest t whose LAST round is still
    // (nearly) full -- 384 blocks would leave half the chip idle for a whole block (the ragged widths of config[3]: 19.6 k vs
    // 18.0 k frames/s on one rank's share) -- else the best-balanced t.
    int best = 1;
    double best_eff = 0.0;
    for (int t = 16; t >= 1; t >>= 1) {
        if (((int64_t)H * W / 256) % t != 0 || blocks1 / t < g_w2d_tpb_blocks) continue;
        const int64_t nb = blocks1 / t;
        const double eff = (double)nb / (256.0 * (double)((nb + 255) / 256));
        if (eff >= 0.94) return t;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = t; }
    }
    return best;
}

int launch_w2d(const ConvArgs& a, hipStream_t s) {
    // the launcher checks its own preconditions (launch_conv's ordering is not a contract)
    if (!a.wino2 || a.in_dt != DT_F32 || a.out_dt != DT_F32 || a.ksplit > 1 || a.partial || a.partial2 || a.sc1 ||
        !conv_w2d_shape_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        set_error("conv_w2d: fp32 in / out, whole K, two-dimensional Winograd weights (ConvArgs::wino2), 16 x 16-pixel tiles, "
                  "channel counts in multiples of 32 (Cout: 64)");
        return ERR_ARG;
    }
    if (a.stats && a.stats_nblk != a.H * a.W / 64) {
        set_error("conv_w2d: statistics come in blocks of 64 pixels (stats_nblk=%d, H W / 64 = %d)", a.stats_nblk, a.H * a.W / 64);
        return ERR_ARG;
    }
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int nj = w2d_channel_tiles(a.B, a.H, a.W, a.Cout);
    const int tpb = w2d_tiles_per_block(a.B, a.H, a.W, a.C1 + a.C2, a.Cout);
    if (tpb < 1 || ((int64_t)a.H * a.W / 256) % tpb != 0 || (tpb > 1 && (((a.C1 + a.C2) / KC) & 1))) {
        set_error("conv_w2d: internal: %d tiles per block do not divide the image's tiles / need an even chunk count", tpb);
        return ERR_STATE;
    }
    const int grid = (int)(M / 256 / tpb) * (a.Cout / (32 * nj));
    const size_t lds = (W2_HBUF + 8 * W2_XDEST) * sizeof(float);       // halo buffer 0 + the exchange region (> two halo buffers)
    const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
#define FLOWSE_LW2D(G, J)                                                                                    \
    {                                                                                                        \
        if (const int rc = allow_lds<&conv3x3_w2d_kernel<G, J>>(lds)) return rc;                             \
        hipLaunchKernelGGL((conv3x3_w2d_kernel<G, J>), dim3(grid), dim3(512), lds, s, a, tpb);                \
    }
    if (nj == 2) {
        if (gn == 2) FLOWSE_LW2D(2, 2) else if (gn == 1) FLOWSE_LW2D(1, 2) else FLOWSE_LW2D(0, 2)
    } else {
        if (gn == 2) FLOWSE_LW2D(2, 1) else if (gn == 1) FLOWSE_LW2D(1, 1) else FLOWSE_LW2D(0, 1)
    }
#undef FLOWSE_LW2D
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
