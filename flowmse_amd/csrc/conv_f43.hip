// Replaces (reference): ddpm_conv3x3 (flowmse/backbones/ncsnpp_utils/layers.py:118-124) inside ResnetBlockBigGANpp
// (layerspp.py:245-274) incl. the GroupNorm + SiLU in front of it (fused into the halo staging), the channel concat of
// ncsnpp.py:337 (two-source A operand), the per-sample time-embedding bias (layerspp.py:262-263), the (x + h)/sqrt(2) skip
// and the statistics of the NEXT GroupNorm (fused into the output stage).
#include "conv_common.h"

namespace flowse {

// ---------------------------------------------------------------------------------------------------
// F(4,3) Winograd 3x3 convolution in fp32 (the production kernel of the fp32 mode, ~75 % of its GPU time).
//
// Along the vertical axis FOUR output rows y..y+3 come from the six input rows d0..d5 = y-1..y+4 through six
// products per horizontal tap -- 6 multiplies where the direct form spends 12, i.e. HALF of the direct-convolution
// FLOPs on the matrix cores (interpolation points 0, +-1, +-2, inf):
//     v = B^T d :  v0 = 4 d0 - 5 d2 + d4          v1 = (d3 + d4) - 4 (d1 + d2)     v2 = (d4 - d3) + 4 (d1 - d2)
//                  v3 = (d4 - d2) + 2 (d3 - d1)    v4 = (d4 - d2) - 2 (d3 - d1)     v5 = 4 d1 - 5 d3 + d5
//     u = G g   :  u0 = g0/4   u1 = -(g0+g1+g2)/6   u2 = -(g0-g1+g2)/6   u3 = g0/24 + g1/12 + g2/6
//                  u4 = g0/24 - g1/12 + g2/6   u5 = g2
//     out = A^T m: o0 = m0+m1+m2+m3+m4   o1 = m1-m2+2(m3-m4)   o2 = m1+m2+4(m3+m4)   o3 = m1-m2+8(m3-m4)+m5
// The 8 x 16 pixel tile is 2 x 16 = 32 row QUADS = one 32-row MFMA tile, so all four waves work on the same quads:
// wave (wn, ch) owns output channels 32 wn .. +31 and the component half ch (0: m0..m2 from d0..d4, 1: m3..m5 from
// d1..d5) -- 3 x 16 accumulators per lane.  The two halves of A^T m are added in the epilogue's LDS tile.  fp32
// error of the 1-D F(4,3) form is ~3x the direct sum's (6e-7 vs 2e-7 rel-L2 on unit-variance data).
constexpr int F43_HROW = 18 * LDS_ROW + 8;

// ---- output stage of the 128-channel F(4,3) blocks (TN = 2): the two component halves of a channel group meet through
// ONE wide LDS exchange instead of a read-modify-write pass over a block-wide C tile.
//
// Wave (wn, CH) holds, for its 64 channels (two 32-channel MFMA tiles j = 0, 1) and the 32 row quads of the pixel tile,
// the Winograd components m0..m2 (CH 0) or m5, m3, m4 (CH 1).  out = A^T m needs both halves:
//     o0 = (m0 + m1 + m2) + (m3 + m4)        o1 = (m1 - m2) + 2 (m3 - m4)
//     o2 = (m1 + m2) + 4 (m3 + m4)           o3 = (m1 - m2) + 8 (m3 - m4) + m5
// Each wave KEEPS tile j = CH and GIVES tile j = 1 - CH to its partner (same wn, other CH; identical lane -> (channel,
// row quad) mapping) as three numbers per accumulator register -- CH 0: (m0 + m1 + m2, m1 - m2, m1 + m2), CH 1:
// (m3 + m4, m3 - m4, m5) -- written as 12 conflict-free ds_write_b128 per lane ([wave][12][lane][4]); one barrier; 12
// ds_read_b128 of the partner's region.  Afterwards every wave owns the FINISHED 32 channels x 128 pixels of one tile:
// it transposes them through the region it has just read (nobody else touches it again) in two passes of 4 image rows
// -- 32 ds_write_b32 + 8 ds_read_b128 per lane and pass, wave-private, no block barrier -- adds bias / per-sample bias /
// residual, scales, stores 16-byte quads (128 contiguous bytes per pixel) and leaves the GroupNorm partial statistics of
// its 32 channels over the whole tile (lane shuffles only: all 128 pixels of a channel live in ONE wave).
// LDS: 4 x 12 KB (overlays the halo buffers; the caller's last loop iteration ended with a barrier).
template <int CH>
__device__ __forceinline__ void f43_out_exchange(const ConvArgs& a, f32x16 (&acc)[3][2], float* smem, int b, int y0, int x0,
                                                 int n0, int tile) {
    const int tid = threadIdx.x;
    int lane = tid & 63;
    // Opaque to the optimiser: everything below that depends only on the lane (row-pass offsets, LDS addresses) would
    // otherwise be hoisted out of the caller's tile loop and kept alive across the main loop -- ~50 registers the 256-VGPR
    // kernel does not have; they were spilled (scratch stores that reached HBM: +25 MB written per launch, PMC WRITE_SIZE)
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    constexpr int JG = 1 - CH, JK = CH;                  // tile given away / tile kept
    float* Xmine = smem + wave * (12 * 256);             // [12][64 lanes][4]
    float* Xpart = smem + (wave ^ 2) * (12 * 256);
    const int W = a.W, Cout = a.Cout;
    const int ch0 = n0 + wn * 64 + JK * 32;              // first of this wave's 32 finished channels
    const int pl = lane >> 3, cq = lane & 7;             // row pass: pixel lane, channel quad
    const bool has_res = a.res != nullptr;
        const int64_t pix0 = ((int64_t)b * a.H + y0) * W + x0;
    const float* resb = a.res + pix0 * Cout + ch0 + cq * 4;
    float* outb = a.out + pix0 * Cout + ch0 + cq * 4;
    int roff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pp = i * 8 + pl;                       // pixel of a 4 x 16 pass, row-major
        roff[i] = ((pp >> 4) * W + (pp & 15)) * Cout;
    }
    // residual quads: the first pass's are requested now (in flight during the exchange), the second pass's as soon as
    // the accumulators are dead (in flight during the first pass) -- never more than the registers the loop state leaves
    float4 rres[2][8];
    if (has_res) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rres[0][i] = *reinterpret_cast<const float4*>(resb + roff[i]);
    }
    // ---- give
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 v0, v1, v2;
        float* e0 = &v0.x; float* e1 = &v1.x; float* e2 = &v2.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            if (CH == 0) {
                const float s12 = acc[1][JG][r] + acc[2][JG][r];
                e0[e] = acc[0][JG][r] + s12;
                e1[e] = acc[1][JG][r] - acc[2][JG][r];
                e2[e] = s12;
            } else {                                     // acc[0] = m5, acc[1] = m3, acc[2] = m4
                e0[e] = acc[1][JG][r] + acc[2][JG][r];
                e1[e] = acc[1][JG][r] - acc[2][JG][r];
                e2[e] = acc[0][JG][r];
            }
        }
        *reinterpret_cast<float4*>(Xmine + ((0 * 4 + g) * 64 + lane) * 4) = v0;
        *reinterpret_cast<float4*>(Xmine + ((1 * 4 + g) * 64 + lane) * 4) = v1;
        *reinterpret_cast<float4*>(Xmine + ((2 * 4 + g) * 64 + lane) * 4) = v2;
    }
    __syncthreads();
    // ---- take: o[k][r] = finished output row k of accumulator register r (tile JK)
    float o[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t0 = *reinterpret_cast<const float4*>(Xpart + ((0 * 4 + g) * 64 + lane) * 4);
        const float4 t1 = *reinterpret_cast<const float4*>(Xpart + ((1 * 4 + g) * 64 + lane) * 4);
        const float4 t2 = *reinterpret_cast<const float4*>(Xpart + ((2 * 4 + g) * 64 + lane) * 4);
        const float* q0 = &t0.x; const float* q1 = &t1.x; const float* q2 = &t2.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            if (CH == 0) {                               // own m0..m2; received s34, d34, m5
                const float s12 = acc[1][JK][r] + acc[2][JK][r], d12 = acc[1][JK][r] - acc[2][JK][r];
                o[0][r] = (acc[0][JK][r] + s12) + q0[e];
                o[1][r] = fmaf(2.f, q1[e], d12);
                o[2][r] = fmaf(4.f, q0[e], s12);
                o[3][r] = fmaf(8.f, q1[e], d12) + q2[e];
            } else {                                     // own m5, m3, m4; received m0+m1+m2, m1-m2, m1+m2
                const float s34 = acc[1][JK][r] + acc[2][JK][r], d34 = acc[1][JK][r] - acc[2][JK][r];
                o[0][r] = q0[e] + s34;
                o[1][r] = fmaf(2.f, d34, q1[e]);
                o[2][r] = fmaf(4.f, s34, q2[e]);
                o[3][r] = fmaf(8.f, d34, q1[e]) + acc[0][JK][r];
            }
        }
    }
    if (has_res) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rres[1][i] = *reinterpret_cast<const float4*>(resb + 4 * W * Cout + roff[i]);
    }
    // ---- transpose through the region just read (wave-private from here on), finish, store, statistics
    float* T = Xpart;                                    // [64 pixels][32 channels] per pass
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + ch0 + cq * 4);
    if (a.bias2) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + ch0 + cq * 4);
        bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
    }
    const float scale = a.scale;
    float4 piv = make_float4(0.f, 0.f, 0.f, 0.f), s1 = piv, s2 = piv;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {               // pass = row quad of the tile: image rows 4 pass .. 4 pass + 3
        if (pass == 1) __builtin_amdgcn_wave_barrier();  // LDS is in-order per wave: pass 0's reads precede these writes
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * pass + e;              // registers of this row quad; tile column (e & 3) + 8 (e >> 2) + 4 kh
                T[(k * 16 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + li] = o[k][r];
            }
        __builtin_amdgcn_wave_barrier();                 // in-order LDS: the tile is complete for this wave's reads
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 8 + pl;
            float4 v = *reinterpret_cast<const float4*>(T + pp * 32 + cq * 4);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (has_res) { v.x += rres[pass][i].x; v.y += rres[pass][i].y; v.z += rres[pass][i].z; v.w += rres[pass][i].w; }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            *reinterpret_cast<float4*>(outb + pass * 4 * W * Cout + roff[i]) = v;
            if (pass == 0 && i == 0) piv = v;
            const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
        }
    }
    if (!a.stats) return;
    // 16 values per lane and channel -> the 8 pixel lanes of a channel quad (equal-count Chan merges) -> 128 pixels
    float mean[4] = {piv.x + s1.x * (1.f / 16), piv.y + s1.y * (1.f / 16), piv.z + s1.z * (1.f / 16), piv.w + s1.w * (1.f / 16)};
    float m2[4] = {fmaxf(s2.x - s1.x * s1.x * (1.f / 16), 0.f), fmaxf(s2.y - s1.y * s1.y * (1.f / 16), 0.f),
                   fmaxf(s2.z - s1.z * s1.z * (1.f / 16), 0.f), fmaxf(s2.w - s1.w * s1.w * (1.f / 16), 0.f)};
    float cnt = 16.f;
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
        float* dst = a.stats + (((int64_t)b * a.stats_nblk + tile) * Cout + ch0 + cq * 4) * 2;
        *reinterpret_cast<float4*>(dst) = make_float4(mean[0], m2[0], mean[1], m2[1]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
    }
}

    // words per halo pixel row: 4 rows = 0 mod 64 banks (quad 1 vs quad 0)

// Per-tile state of the staging pipeline: which pixels of the 10 x 18 halo lie inside the image and the window
// descriptors of the two source tensors.  A block that owns several tiles (tpb > 1) keeps the state of the tile it
// computes and of the one it stages for.
struct F43Tile {
    int y0, x0, ty, tx;
    unsigned hin;                                        // bit q: this thread's halo quad q lies inside the image
    unsigned woff;                                       // pixel offset of the tile's window inside the sample's descriptor
};

template <int GN, int CH, bool SPLIT, int TN>
__device__ __forceinline__ void conv3x3_f43_body(const ConvArgs& a, float* smem, int tpb) {
    static_assert(TN == 1 || !SPLIT, "the sliced form keeps the 64-channel block");
    constexpr int BN = 64 * TN;                          // TN 32-channel tiles per wave, two channel groups (wn) per block
    constexpr int HROWS = 180;                           // 10 x 18 halo pixels
    constexpr int H_LOADS = 6;
    constexpr int HBUF = 10 * F43_HROW;                  // floats per halo buffer
    float* Hs = smem;                                    // [2][10][F43_HROW]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // a block owns `tpb` consecutive pixel tiles (in walk order) of ONE channel block
    const int mg = bid / n_ntiles;
    const int nt = bid - mg * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 3);
    const int n0 = nt * BN;
    const int col4 = tid & 7, row0 = tid >> 3;
    unsigned hpix[H_LOADS];                              // pixel offset of this thread's halo quads in the window
    int hlds[H_LOADS];                                   // their LDS word offset (-1: past the last halo pixel)
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        hpix[q] = (unsigned)(hy * W + hx);
        hlds[q] = hr < HROWS ? hy * F43_HROW + hx * LDS_ROW + col4 * 4 : -1;
    }
    // Tiles of an image are walked in vertical strips of 4 tiles (64 pixels), top to bottom: the rows a tile shares
    // with its vertical neighbour are re-read 4 tiles later instead of a full tile row later, which keeps that window
    // plus the streamed weights inside the 4 MB L2 of the XCD for 256-channel layers (2.3x -> ~1.1x HBM reads).
    // All tiles of a block lie in ONE sample (launch_f43: tiles per block divides the tiles of an image), so the two source
    // descriptors are per block -- base = the sample's pixel (-W - 1), i.e. the window origin of its first tile -- and a
    // tile only contributes the scalar offset of its window (no per-tile descriptor state in registers).
    const int bsmp = (mg * tpb) / tiles_img;
    const int b = bsmp;
    const int64_t sbase = (int64_t)bsmp * HW - W - 1;
    const int spix = HW + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + sbase * C1), 0, spix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + sbase * C2 : a.in1), 0, C2 ? spix * C2 * 4 : 0, 0x00020000);
    auto make_tile = [&](int mt) {
        F43Tile t;
        const int tt = mt - bsmp * tiles_img;
        if ((tiles_x & 3) == 0) {
            const int per_strip = 4 * (H >> 3);
            const int strip = tt / per_strip, w = tt - strip * per_strip;
            t.ty = w >> 2;
            t.tx = strip * 4 + (w & 3);
        } else {
            t.ty = tt / tiles_x;
            t.tx = tt - t.ty * tiles_x;
        }
        t.y0 = t.ty * 8;
        t.x0 = t.tx * 16;
        t.hin = 0;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            const int hy = hr / 18, hx = hr - hy * 18;
            const bool in = hr < HROWS && (unsigned)(t.y0 - 1 + hy) < (unsigned)H && (unsigned)(t.x0 - 1 + hx) < (unsigned)W;
            t.hin |= in ? (1u << q) : 0u;
        }
        t.woff = (unsigned)(t.y0 * W + t.x0);
        return t;
    };
    F43Tile cur = make_tile(mg * tpb);
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wino), 0, a.Cout * 18 * Cin * 4, 0x00020000);

    // The halo of the next chunk is staged in two halves of three quads (request -> GroupNorm/SiLU in registers ->
    // LDS write into the idle buffer), so that only 12 staging registers are live at any time
    u32x4 rh[3];
    float4 g_mu, g_sc, g_be;
    unsigned st_hin = cur.hin;                           // halo mask of the tile being STAGED (cur, or the block's next tile)

    auto hload = [&](const F43Tile& t, int chunk, int Q) -> u32x4 {
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned cs = (unsigned)(second ? C2 : C1);
        const unsigned soff = (t.woff * cs + (unsigned)(second ? c0 - C1 : c0)) * 4u;
        const unsigned off = ((t.hin >> Q) & 1u) ? (hpix[Q] * cs + (unsigned)col4 * 4u) * 4u : OOB;
        return second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, soff, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, soff, 0);
    };
    auto gparams = [&](const F43Tile& t, int chunk) {
        if (GN) {
            const int cg = chunk * KC + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
    };
    auto gloadH = [&](const F43Tile& t, int chunk, int h) {
#pragma unroll
        for (int q = 0; q < 3; ++q) rh[q] = hload(t, chunk, 3 * h + q);
        if (h == 0) {
            gparams(t, chunk);
            st_hin = t.hin;
        }
    };
    auto xform1 = [&](int Q) {
        if (GN) rh[Q % 3] = gn_quad<GN>(rh[Q % 3], g_mu, g_sc, g_be, (st_hin >> Q) & 1u);
    };
    auto lstoreH = [&](int buf, int h) {
        float* Hb = Hs + buf * HBUF;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (hlds[3 * h + q] >= 0) *reinterpret_cast<u32x4*>(Hb + hlds[3 * h + q]) = rh[q];
    };

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;                             // CH = wave >> 1 (template parameter)
    const int li = lane & 31, kh = lane >> 5;
    // this lane's quad: row quad li >> 4, column li & 15; its input rows start at halo row 4*quad (+1 for CH = 1)
    const int abase = (4 * (li >> 4) + CH) * F43_HROW + (li & 15) * LDS_ROW + kh * 4;
    // weight fragments: 24 KB per (32-channel slice, kx, chunk), [component 0..5][k-block][lane][4 floats]
    const int nchunks = Cin / KC;
    // split-K: gridDim.y slices of consecutive chunks; each slice leaves a raw partial tile (the epilogue's split form)
    const int per_slice = (nchunks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c_begin = (int)blockIdx.y * per_slice, c_end = min(nchunks, c_begin + per_slice);
    const unsigned wslice = (unsigned)((n0 >> 5) + wn * TN) * 3u * (unsigned)nchunks;    // in 24 KB units; tile j adds 3 nchunks
    const unsigned wvo = (unsigned)lane * 16u + (unsigned)CH * 3u * 4096u;

    f32x16 acc[3][TN];                                   // this wave's three Winograd components x TN channel tiles

    {   // first chunk of the block's first tile: all six quads at once (the accumulators are not live yet)
        u32x4 t[H_LOADS];
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) t[q] = hload(cur, c_begin, q);
        gparams(cur, c_begin);
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            if (GN) t[q] = gn_quad<GN>(t[q], g_mu, g_sc, g_be, (cur.hin >> q) & 1u);
            if (hlds[q] >= 0) *reinterpret_cast<u32x4*>(Hs + hlds[q]) = t[q];
        }
    }
    __syncthreads();

#define FLOWSE_FENCE __builtin_amdgcn_sched_barrier(0);
    // five halo rows of k-block (KX, J) from LDS; three weight components of k-block (KX, J) of chunk CHK from L2
#define FLOWSE_WLOADA(KX, J, D)                                                                                      \
    {                                                                                                                \
        const float* Ha = Hcur + abase + (KX) * LDS_ROW + (J) * 8;                                                   \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) D[r] = *reinterpret_cast<const float4*>(Ha + r * F43_HROW);    \
    }
#define FLOWSE_WLOADB(KX, J, CHK, BF)                                                                                \
    {                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                             \
            const unsigned so = (wslice + (unsigned)(3 * j + (KX)) * (unsigned)nchunks + (unsigned)(CHK)) * 24576u;  \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                          \
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, wvo + (c * 4 + (J)) * 1024, so, 0);     \
                BF[c][j] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z),            \
                                       __uint_as_float(t.w));                                                        \
            }                                                                                                        \
        }                                                                                                            \
    }
#define FLOWSE_F4(OP) { OP(x) OP(y) OP(z) OP(w) }
    // input transform, in place: D[0..2] become this wave's three operands.
    // CH 0 (rows d0..d4): v0 = 4 d0 - 5 d2 + d4, v1 = (d3 + d4) - 4 (d1 + d2), v2 = (d4 - d3) + 4 (d1 - d2)
    // CH 1 (rows d1..d5 as D[0..4]): v5 = 4 d1 - 5 d3 + d5 -> D[0];  v3 = (d4 - d2) + 2 (d3 - d1) -> D[1];
    //                                v4 = (d4 - d2) - 2 (d3 - d1) -> D[2]
    // (float2 halves: hipcc emits the packed v_pk_add / v_pk_fma forms, half the VALU instructions)
#define FLOWSE_H2(Q, H) (*reinterpret_cast<f32x2*>(&(Q).x + 2 * (H)))
#define FLOWSE_WXA(D)                                                                                                \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f};                                                             \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                              \
            const f32x2 r0 = FLOWSE_H2(D[0], h), r2 = FLOWSE_H2(D[2], h), r4 = FLOWSE_H2(D[4], h);                   \
            const f32x2 v = __builtin_elementwise_fma(c4, r0, __builtin_elementwise_fma(cm5, r2, r4));               \
            if (CH == 0) FLOWSE_H2(D[0], h) = v;   /* v0 = 4 d0 - 5 d2 + d4 -> D[0] */                                \
            else FLOWSE_H2(D[4], h) = v;           /* v5 = 4 d1 - 5 d3 + d5 -> D[4] */                                \
        }                                                                                                            \
    }
#define FLOWSE_WXB(D)                                                                                                \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm4 = {-4.f, -4.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};                        \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                              \
            const f32x2 r0 = FLOWSE_H2(D[0], h), r1 = FLOWSE_H2(D[1], h), r2 = FLOWSE_H2(D[2], h),                   \
                        r3 = FLOWSE_H2(D[3], h), r4 = FLOWSE_H2(D[4], h);                                            \
            if (CH == 0) { /* r1..r4 = d1..d4 */                                                                     \
                FLOWSE_H2(D[1], h) = __builtin_elementwise_fma(cm4, r1 + r2, r3 + r4);                               \
                FLOWSE_H2(D[2], h) = __builtin_elementwise_fma(c4, r1 - r2, r4 - r3);                                \
            } else {       /* r0..r3 = d1..d4 */                                                                     \
                FLOWSE_H2(D[1], h) = __builtin_elementwise_fma(c2, r2 - r0, r3 - r1);                                \
                FLOWSE_H2(D[2], h) = __builtin_elementwise_fma(cm2, r2 - r0, r3 - r1);                               \
            }                                                                                                        \
        }                                                                                                            \
    }
    // operands: CH 0 -> D[0], D[1], D[2] = v0, v1, v2;  CH 1 -> D[4], D[1], D[2] = v5, v3, v4
#define FLOWSE_WMMA3(V, BF, K)                                                                                       \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[(c == 0 && CH == 1) ? 4 : c].K, BF[c][j].K, acc[c][j], 0, 0, 0);
    // One k-block: request the next block's operands (halo rows from LDS, weights from L2), run this block's 12
    // MFMAs with the next block's transform (and one staged halo quad) fenced in between
    // HL >= 0: this phase also requests half HL of the next chunk's halo -- AFTER its own operand requests.  Vector
    // memory loads return (and vmcnt counts) in issue order, so every weight fragment requested after a halo load waits
    // for that load's HBM round trip.  Without the in-loop halo loads (probe build, results garbage) the kernel is 6-8 %
    // faster and the difference is exactly the waves' parked time (SQ_WAIT_ANY 1.19e8 -> 0.74e8 quad-cycles per launch,
    // profiles/r03_f43_probes.md).  Requesting the halo behind the phase's weights and normalising it three phases later
    // instead of one (this schedule) did NOT recover it (447 vs 447 us): the wait is not a fixed latency one can cover
    // with 3 000 cycles but the tail of the HBM round trips of the 8 load batches a block issues per chunk, which gates
    // the chunk barrier. 
    // The next k-block's operand requests ride in the gaps of this block's first two MFMA groups, ONE per MFMA (five
    // ds_read_b128 behind group x, 3 TN buffer loads behind group y) instead of as one clump ahead of group x: 432 -> 427 us on
    // the dominant instantiation (A-B on one box, three runs each).
#define FLOWSE_WM1(V, BF, K, c, j)                                                                                   \
    acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[((c) == 0 && CH == 1) ? 4 : (c)].K, BF[c][j].K, acc[c][j], 0, 0, 0);
#define FLOWSE_WPHASE(V, BF, NKX, NJ, NCHK, DN, BFN, XQ, HL)                                                         \
    {                                                                                                                \
        const float* Ha = Hcur + abase + (NKX) * LDS_ROW + (NJ) * 8;                                                 \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) _Pragma("unroll") for (int j = 0; j < TN; ++j) {               \
            FLOWSE_WM1(V, BF, x, c, j)                                                                               \
            FLOWSE_FENCE                                                                                             \
            _Pragma("unroll") for (int r = (c * TN + j) * 5 / (3 * TN); r < (c * TN + j + 1) * 5 / (3 * TN); ++r)    \
                DN[r] = *reinterpret_cast<const float4*>(Ha + r * F43_HROW);                                         \
            FLOWSE_FENCE                                                                                             \
        }                                                                                                            \
    }                                                                                                                \
    if (GN && (XQ) >= 0) xform1((XQ) < 0 ? 0 : (XQ));                                                                \
    FLOWSE_FENCE                                                                                                     \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) _Pragma("unroll") for (int j = 0; j < TN; ++j) {                   \
        FLOWSE_WM1(V, BF, y, c, j)                                                                                   \
        FLOWSE_FENCE                                                                                                 \
        {                                                                                                            \
            const unsigned so = (wslice + (unsigned)(3 * j + (NKX)) * (unsigned)nchunks + (unsigned)(NCHK)) * 24576u; \
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, wvo + (c * 4 + (NJ)) * 1024, so, 0);        \
            BFN[c][j] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z),               \
                                    __uint_as_float(t.w));                                                           \
        }                                                                                                            \
        FLOWSE_FENCE                                                                                                 \
    }                                                                                                                \
    if ((HL) >= 0) gloadH(stile, cnext, (HL) < 0 ? 0 : (HL));                                                        \
    FLOWSE_FENCE                                                                                                     \
    FLOWSE_WXA(DN) FLOWSE_FENCE FLOWSE_WMMA3(V, BF, z) FLOWSE_FENCE                                                  \
    FLOWSE_WXB(DN) FLOWSE_FENCE FLOWSE_WMMA3(V, BF, w) FLOWSE_FENCE

    float4 dA[5], dB[5], bA[3][TN], bB[3][TN];
    // first weight fragments of a tile: requested here for the block's first tile and again right after a tile's output
    // stage (not before it: 24 registers that would have to survive the stage)
    auto first_weights = [&]() { FLOWSE_WLOADB(0, 0, c_begin, bA) };
    first_weights();
    // ---- tiles of this block.  The staging pipeline runs ACROSS tile boundaries: during a tile's last chunk the halo of
    // the NEXT tile's first chunk (and its first weight fragments) are requested, normalised and written to the idle LDS
    // buffer exactly like any other "next chunk", so only the block's first tile pays a prologue (tpb > 1 needs an even
    // number of chunks: every tile then starts in buffer 0, and the output stage lives behind it, see launch_f43).
    for (int ti = 0; ti < tpb; ++ti) {
    const bool more = ti + 1 < tpb;
    const int y0 = cur.y0, x0 = cur.x0, ty = cur.ty, tx = cur.tx;
    const int m_tl = (b * H + y0) * W + x0;
    (void)ty; (void)tx;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float* Hcur = Hs + ((chunk - c_begin) & 1) * HBUF;
        const bool wrap = chunk + 1 >= c_end;            // the tile's last chunk stages for the block's next tile
        const int cnext = wrap ? (more ? c_begin : c_end - 1) : chunk + 1, nbuf = (chunk - c_begin + 1) & 1;
        // (the next tile's state is derived here, for the one chunk that needs it: nothing extra stays live in the loop)
        const F43Tile stile = (wrap && more) ? make_tile(mg * tpb + ti + 1) : cur;
        FLOWSE_WLOADA(0, 0, dA)
        FLOWSE_WXA(dA) FLOWSE_WXB(dA)
        FLOWSE_FENCE
        // twelve phases; the next chunk's halo: first half requested in phase 1, normalised in phases 4-6, second half
        // requested in phase 7, normalised in phases 10-12 (the idle buffer: nobody reads it during this chunk)
        FLOWSE_WPHASE(dA, bA, 0, 1, chunk, dB, bB, -1, 0)
        FLOWSE_WPHASE(dB, bB, 0, 2, chunk, dA, bA, -1, -1)
        FLOWSE_WPHASE(dA, bA, 0, 3, chunk, dB, bB, -1, -1)
        FLOWSE_WPHASE(dB, bB, 1, 0, chunk, dA, bA, 0, -1)
        FLOWSE_WPHASE(dA, bA, 1, 1, chunk, dB, bB, 1, -1)
        FLOWSE_WPHASE(dB, bB, 1, 2, chunk, dA, bA, 2, -1)
        lstoreH(nbuf, 0);
        FLOWSE_FENCE
        FLOWSE_WPHASE(dA, bA, 1, 3, chunk, dB, bB, -1, 1)
        FLOWSE_WPHASE(dB, bB, 2, 0, chunk, dA, bA, -1, -1)
        FLOWSE_WPHASE(dA, bA, 2, 1, chunk, dB, bB, -1, -1)
        FLOWSE_WPHASE(dB, bB, 2, 2, chunk, dA, bA, 3, -1)
        FLOWSE_WPHASE(dA, bA, 2, 3, chunk, dB, bB, 4, -1)
        if (!wrap) { FLOWSE_WLOADB(0, 0, cnext, bA) }     // first weights of the next chunk (a next TILE's: after the output stage)
        FLOWSE_FENCE
        FLOWSE_WMMA3(dB, bB, x) FLOWSE_FENCE
        if (GN) xform1(5);
        FLOWSE_FENCE
        FLOWSE_WMMA3(dB, bB, y) FLOWSE_WMMA3(dB, bB, z) FLOWSE_WMMA3(dB, bB, w)
        FLOWSE_FENCE
        lstoreH(nbuf, 1);
        __syncthreads();     // next chunk's halo is complete; everyone has left this chunk's
    }
#undef FLOWSE_WLOADA
#undef FLOWSE_WLOADB
#undef FLOWSE_F4
#undef FLOWSE_H2
#undef FLOWSE_WXA
#undef FLOWSE_WXB
#undef FLOWSE_WMMA3
#undef FLOWSE_WPHASE
#undef FLOWSE_FENCE

    // This wave's half of A^T m, laid out like four 32-pixel tiles of the <2,2,2,1> epilogue: accumulator register
    // r holds quad row (r&3) + 8*(r>>2) + 4*kh, i.e. row quad r >> 3; output row 4*quad + o is tile row pair
    // 2*quad + (o >> 1), second row of the pair when o is odd -> tile-row index i = 2*(r>>3) + (o>>1), r' = (r&7) + 8*(o&1).
    // half >= 0: the C tile holds 64 channels -- with TN = 1 the two channel groups (wn) side by side; with TN = 2 the
    // 64 channels of the group `half`, the other group's waves only keep the barrier.  half < 0 (TN = 2): the tile holds
    // all 128 channels and both groups scatter at once.
    auto scatter_half = [&](float* Cs, int CROW, int half) {
        const bool mine = TN == 1 || half < 0 || wn == half;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == CH && mine) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float* Cw = Cs + (TN == 1 ? wn * 32 : half < 0 ? wn * 64 + j * 32 : j * 32) + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o[4];
                        if (CH == 0) {
                            const float s12 = acc[1][j][r] + acc[2][j][r], d12 = acc[1][j][r] - acc[2][j][r];
                            o[0] = acc[0][j][r] + s12; o[1] = d12; o[2] = s12; o[3] = d12;
                        } else {       // acc[0] = m5, acc[1] = m3, acc[2] = m4
                            const float s34 = acc[1][j][r] + acc[2][j][r], d34 = acc[1][j][r] - acc[2][j][r];
                            o[0] = s34; o[1] = 2.f * d34; o[2] = 4.f * s34; o[3] = fmaf(8.f, d34, acc[0][j][r]);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // tile row of (quad r>>3, output row k, column bits of r)
                            const int row = (2 * (r >> 3) + (k >> 1)) * 32 + ((r & 7) + 8 * (k & 1) & 3) +
                                            8 * (((r & 7) + 8 * (k & 1)) >> 2) + 4 * kh;
                            if (pass == 0) Cw[row * CROW] = o[k];
                            else Cw[row * CROW] += o[k];
                        }
                    }
                }
            }
            if (pass == 0) __syncthreads();
        }
    };
    if constexpr (SPLIT) {                               // split slices: raw partial tiles through the shared epilogue
        conv_epilogue_with<2, 2, 2, 1>(a, smem, m_tl, n0, M, HW, (int)blockIdx.y, W,
                                       [&](float* Cs, int CROW) { scatter_half(Cs, CROW, 0); });
    } else if constexpr (TN == 2) {
        const int tile_ix = ty * tiles_x + tx;           // row-major index of this 8 x 16 tile in the sample's tile grid
        // the exchange region sits BEHIND halo buffer 0, which already holds the next tile's first chunk (tpb > 1)
        f43_out_exchange<CH>(a, acc, smem + HBUF, b, y0, x0, n0, tile_ix);
        if (more) {
            first_weights();
            __syncthreads();                             // buffer 1 (under the exchange region) is written again in the next tile
        }
    } else {
        // C tile of all 64 TN channels ([128][64 TN + 4] floats): with TN = 2 both channel groups scatter at once (two
        // waves per pass instead of one), then the output stage runs over the two 64-channel halves back to back
        constexpr int CROW = 64 * TN + 4;
        const int bsmp = m_tl / HW;
        const int rem = m_tl - bsmp * HW;
        const int tile = ((rem / W) >> 3) * (W >> 4) + ((rem % W) >> 4);
        scatter_half(smem, CROW, -1);
        __syncthreads();
        float* red = smem + 128 * CROW;
#pragma unroll 1
        for (int half = 0; half < TN; ++half) {
            if (half) __syncthreads();                   // the statistics scratch of the first half has been read
            tile128x64_out<float>(a, smem + half * 64, CROW, red, m_tl, W, n0 + half * 64, bsmp, tile);
        }
    }
    if (more) cur = make_tile(mg * tpb + ti + 1);
    }   // tiles of this block
}

// SPLIT: the launch is sliced over chunks (gridDim.y > 1) and every block leaves a raw partial tile
// TN = 1: 64 output channels per block, three blocks per CU.  TN = 2: 128 channels per block (each wave two 32-channel
// tiles), two blocks per CU: every transformed input fragment, every staged (GroupNorm + SiLU) halo element and every LDS
// read feeds twice the MFMAs -- the VALU work per MFMA, which is what holds the matrix pipe below 0.75 in the TN = 1 form
// (three waves of a SIMD issue ~2 VALU per 64-cycle MFMA), halves.
template <int GN, bool SPLIT = false, int TN = 1>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 2) void conv3x3_f43_kernel(ConvArgs a, int tpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (SPLIT || TN == 1) tpb = 1;                       // only the 128-channel whole-K form owns several tiles per block
    // waves 0,1: component half 0; waves 2,3: half 1.  Both bodies execute the same barriers.
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 7)) conv3x3_f43_body<GN, 1, SPLIT, TN>(a, smem, tpb);
    else conv3x3_f43_body<GN, 0, SPLIT, TN>(a, smem, tpb);
}

// [Cout][9][Cin] -> fragment order [Cout/32][kx][Cin/32][component 0..5][k-block j][lane][4]; stored component order:
// 0,1,2 = u0,u1,u2 (wave half 0), 3,4,5 = u5,u3,u4 (wave half 1: operands D[0] = v5, D[1] = v3, D[2] = v4)
__global__ __launch_bounds__(256) void f43_weights_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                          float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (n, kx, ci)
    if (idx >= (int64_t)Cout * 3 * Cin) return;
    const int ci = (int)(idx % Cin);
    const int kx = (int)((idx / Cin) % 3);
    const int64_t n = idx / ((int64_t)3 * Cin);
    const float g0 = w[(n * 9 + 0 + kx) * Cin + ci], g1 = w[(n * 9 + 3 + kx) * Cin + ci],
                g2 = w[(n * 9 + 6 + kx) * Cin + ci];
    const int nchunks = Cin >> 5;
    const int chunk = ci >> 5, j = (ci >> 3) & 3, kh = (ci >> 2) & 1, e = ci & 3;
    const int lane = kh * 32 + (int)(n & 31);
    float* o = out + ((((n >> 5) * 3 + kx) * nchunks + chunk) * 24 + j) * 256 + lane * 4 + e;   // component slot 0
    const float s02 = g0 + g2;
    o[0] = 0.25f * g0;
    o[1024] = (s02 + g1) * (-1.f / 6.f);
    o[2048] = (s02 - g1) * (-1.f / 6.f);
    const float t = fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f)), h = g1 * (1.f / 12.f);
    o[3072] = g2;            // u5
    o[4096] = t + h;         // u3
    o[5120] = t - h;         // u4
}

int launch_f43_weights(const float* w_packed, int Cout, int Cin, float* out, hipStream_t s) {
    if ((Cout % 32) != 0 || (Cin % 32) != 0) {
        set_error("f43_weights: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
        return ERR_SHAPE;
    }
    const int64_t n = (int64_t)Cout * 3 * Cin;
    hipLaunchKernelGGL(f43_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_packed, Cout, Cin, out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// 128-channel blocks (two per CU) when the layer allows it and at least one full round of 512 such blocks exists
// (measured: 512 beats 1024 and 256 at B = 1 and B = 8)
bool conv_f43_wide(int B, int H, int W, int Cout) {
    return (Cout % 128) == 0 && ((int64_t)B * H * W / 128) * (Cout / 128) >= 512;
}

int launch_f43(const ConvArgs& a, hipStream_t s) {
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int ks = a.ksplit > 1 ? a.ksplit : 1;       // slices of 32-channel chunks (gridDim.y), see f43_plan
    const bool wide = !a.partial && conv_f43_wide(a.B, a.H, a.W, a.Cout);
    // Tiles per block of the 128-channel form: as many (4, 2) as still leave two full rounds of 512 blocks (256 CUs x 2),
    // so that the prologue -- first halo from HBM, its GroupNorm, first weights, ~14 % of a one-tile block's life -- is
    // paid once per block instead of once per tile.  Needs an even chunk count (every tile starts in halo buffer 0).
    int tpb = 1;
    if (wide && (((a.C1 + a.C2) / KC) & 1) == 0) {
        const int64_t blocks1 = (M / 128) * (a.Cout / 128);
        for (int t = 4; t >= 2; t >>= 1)
            if (((int64_t)a.H * a.W / 128) % t == 0 && blocks1 / t >= 1024) { tpb = t; break; }
    }
    const int grid = (int)(M / 128 / tpb) * (a.Cout / (wide ? 128 : 64));
    const size_t lds_halo = 2 * 10 * F43_HROW * sizeof(float);         // two halo buffers; > the 64-channel C tile
    const size_t lds_c = (10 * F43_HROW + 4 * 12 * 256) * sizeof(float);   // halo buffer 0 + the exchange region of the output stage
    const size_t lds = wide && lds_c > lds_halo ? lds_c : lds_halo;
    const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
#define FLOWSE_LF43(G, SP, TNV)                                                                              \
    {                                                                                                        \
        if (const int rc = allow_lds<&conv3x3_f43_kernel<G, SP, TNV>>(lds)) return rc;                       \
        hipLaunchKernelGGL((conv3x3_f43_kernel<G, SP, TNV>), dim3(grid, ks), dim3(256), lds, s, a, tpb);          \
    }
    if (a.partial) {
        if (gn == 2) FLOWSE_LF43(2, true, 1) else if (gn == 1) FLOWSE_LF43(1, true, 1) else FLOWSE_LF43(0, true, 1)
    } else if (wide) {
        if (gn == 2) FLOWSE_LF43(2, false, 2) else if (gn == 1) FLOWSE_LF43(1, false, 2) else FLOWSE_LF43(0, false, 2)
    } else {
        if (gn == 2) FLOWSE_LF43(2, false, 1) else if (gn == 1) FLOWSE_LF43(1, false, 1) else FLOWSE_LF43(0, false, 1)
    }
#undef FLOWSE_LF43
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
