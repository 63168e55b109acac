// Dispatch and shape policies of the convolution entry point launch_conv (common.h).
//
//   out[m][n] = ( sum_{tap, ci} A[m][tap, ci] * Wp[n][tap][ci] + bias[n] + bias2[b(m)][n] + res[m][n] ) * scale
//   m = NHWC pixel (b, y, x), n = output channel, K = taps * Cin with the channel axis contiguous in both operands.
//
// Kernel families (one translation unit each):
//   conv_f43.hip      fp32 F(4,3) Winograd 3x3 (production kernel of the fp32 mode)
//   conv_w2d.hip      fp32 F(4,3) x F(2,3) two-dimensional Winograd 3x3 (large images, whole K)
//   conv_halo.hip     direct fp32 LDS-halo 3x3, 4-channel heads / input layers
//   conv_flat.hip     flat fp32 kernels: 1x1, small 3x3, split-K slices
//   conv_smallm.hip   fp32 convs on <= 2048 pixels: K split inside the block (no slab, no reduction launch)
//   conv_1x1.hip      fp32 1x1 on large images: every wave its own GEMM, operands straight into fragment registers
//   conv_reduce.hip   second pass of split-K launches (+ fused GroupNorm statistics / GroupNorm)
//   conv16.hip        16-bit operand / storage modes
// Replaces (reference): ddpm_conv3x3 / ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-124), NIN (:546-555).
#include "conv_common.h"

namespace flowse {

// Split-K policy: images so small that the 128x128 tiling yields < 256 blocks (one per CU) are sliced along K
// until ~512 blocks exist, keeping >= 4 K steps per slice; a deterministic second launch (conv_reduce.hip) sums the slices.
// (The in-launch reduction by the last-arriving slice was built in round 2 and measured slower -- 137.4 vs 122.7 ms per
// step at [8,1,256,256], 44.5 vs 33.4 ms at [1,1,256,256]: the last slice of a tile pulls ks x 64 KB through ONE CU's L1
// after its own K loop, while the two-pass kernel spreads the same bytes over every CU; removed in round 4.)
int conv_splitk_stats_group(int HW) { return sk_pixels_per_block(HW); }

int conv_fused_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps) {
    const int HW = H * W;
    if (Cout & 3) return 0;
    if (conv_smallm_ok(B, H, W, Cin, 0, Cout, taps)) return conv_smallm_stats_blocks(B, H, W);
    if (conv1x1_stream_ok(B, H, W, Cin, 0, Cout, taps)) return conv1x1_stream_stats_blocks(B, H, W);
    if (conv_ksplit(B, H, W, Cin, Cout, taps) != 1) {      // statistics come from the split-K reduction
        const int PB = sk_pixels_per_block(HW);
        return ((HW % PB) == 0 && Cout / 4 <= 256) ? HW / PB : 0;
    }
    return (HW % 128) == 0 ? HW / 128 : 0;
}

// test hooks, read once: FLOWSE_FORCE_GENERIC_CONV=1 routes every shape through the generic gather kernel,
// FLOWSE_NO_HALO_CONV=1 through the flat kernels, FLOWSE_NO_WINOGRAD=1 runs the direct (bitwise fmaf-chain) 3x3 kernel
static const bool g_force_generic = getenv("FLOWSE_FORCE_GENERIC_CONV") != nullptr;
static const bool g_no_halo = getenv("FLOWSE_NO_HALO_CONV") != nullptr;
static const bool g_no_wino = getenv("FLOWSE_NO_WINOGRAD") != nullptr;
static const bool g_no_wino_policy = g_no_wino || g_no_halo || g_force_generic;
// FLOWSE_W2D=0 keeps the 1-D F(4,3) kernel everywhere (A-B hook); default: the 2-D kernel where conv_supports_w2d says so
static const bool g_w2d = !(getenv("FLOWSE_W2D") && getenv("FLOWSE_W2D")[0] == '0');
bool conv_w2d_enabled() { return g_w2d && !g_no_wino_policy; }
// FLOWSE_NO_SMALLM=1: small images run the split-K flat / F(4,3) kernels + reduction launches of rounds 1-4 (A-B hook)
static const bool g_no_smallm = getenv("FLOWSE_NO_SMALLM") != nullptr;
static const int g_smallm_max = getenv("FLOWSE_SMALLM_MAX") ? atoi(getenv("FLOWSE_SMALLM_MAX")) : 2048;
bool conv_smallm_ok(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (g_no_smallm || g_force_generic || (taps != 1 && taps != 9) || (C1 % KC) || (C2 % KC) || (Cout % 32) || C1 <= 0) return false;
    const int64_t M = (int64_t)B * H * W;
    // (2048 pixels = 16 x 16 at batch 8: 4.96 vs 5.76 ms for the level against the sliced F(4,3) kernel + reduction launches,
    // A-B-A-B on one box; FLOWSE_SMALLM_MAX moves the limit)
    if (M > g_smallm_max || M < 1) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return (int64_t)(32 + 2 * W + 2) * cmax * 4 < (1LL << 31) && (int64_t)Cout * taps * (C1 + C2) * 4 < (1LL << 31);
}
bool conv_force_generic() { return g_force_generic; }

// Winograd plan for a 3x3 shape: 0 = not a Winograd shape (or too small even when sliced), 1 = the Winograd halo
// kernel runs over the whole K, >= 2 = it runs split over that many slices of 32-channel chunks (deterministic
// two-pass reduction as for the flat kernel).  The Winograd kernels tile N by 64, so 256 (pixel tile, channel
// block) pairs -- one per CU -- already beat slicing K; below that, K is cut until ~512 blocks exist while every
// slice keeps at least two chunks.  Only the F(4,3) kernel knows how to run a slice.
int f43_plan(int B, int H, int W, int Cin, int Cout, int taps) {
    if (g_no_wino_policy || taps != 9 || (H & 7) || (W & 15) || (Cin % KC) || (Cout % 64)) return 0;
    const int64_t blocks = ((int64_t)B * H * W / 128) * (Cout / 64);
    if (blocks >= 256) return 1;                      // (round 4 sweep: 128 / 512 / 1024 all lose at B = 1, 4, 8: profiles/r04_policy_sweep.md)
    const int nchunks = Cin / KC;
    if (blocks < 16 || nchunks < 4) return 0;
    int64_t ks = (512 + blocks - 1) / blocks;
    if (ks > nchunks / 2) ks = nchunks / 2;
    const int per = (int)((nchunks + ks - 1) / ks);
    ks = (nchunks + per - 1) / per;                   // every slice non-empty
    return ks >= 2 ? (int)ks : 0;
}

int conv_ksplit(int B, int H, int W, int Cin, int Cout, int taps) {
    if (conv_supports_head4(B, H, W, Cin, 0, Cout, taps)) return 1;     // 4-channel heads: dedicated kernel
    if (conv_smallm_ok(B, H, W, Cin, 0, Cout, taps)) return 1;          // K is split inside the block
    if (conv_w2d_enabled() && conv_supports_w2d(B, H, W, Cin, 0, Cout, taps)) return 1;   // whole K on the 2-D Winograd kernel
    const int64_t M = (int64_t)B * H * W;
    const int bn = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;       // N tile launch_conv picks for this width
    const int bm = conv_small_m(M, Cout) ? 32 : 128;              // M tile (single utterances at the 8x8 / 4x4 levels: 32 rows)
    const int64_t tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    const int steps = ((Cin + KC - 1) / KC) * taps;
    if (tiles >= 256 || steps < 8) return 1;          // measured: 256 beats 128 and 64 at B = 1..8
    const int wp = f43_plan(B, H, W, Cin, Cout, taps);
    if (wp >= 1) return wp;
    int64_t want = (512 + tiles - 1) / tiles;
    int64_t maxs = steps / 4;                         // (>= 4 K steps per slice: 2 and 1 lose, same sweep)
    int64_t ks = want < maxs ? want : maxs;
    if (ks < 1) ks = 1;
    // make every slice non-empty
    const int per = (int)((steps + ks - 1) / ks);
    ks = (steps + per - 1) / per;
    return (int)ks;
}

// Images with at most 64 pixels in the whole batch (one utterance at the 8x8 and 4x4 levels): a 128-row tile would spend
// 50-87 % of its MFMAs on padding rows and every K step costs 64 MFMAs per wave whatever M is.  They run 32-row tiles
// (4 waves side by side along N, 16 MFMAs per wave and step) and are sliced along K accordingly.
bool conv_small_m(int64_t M, int Cout) { return M <= 64 && Cout > 64; }

bool conv_splitk_is_wino(int B, int H, int W, int Cin, int Cout, int taps) {
    return conv_supports_wino(B, H, W, Cin, 0, Cout, taps) && f43_plan(B, H, W, Cin, Cout, taps) >= 2;
}

bool conv_supports_fused_gn(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (conv_supports_head4(B, H, W, C1, C2, Cout, taps)) return true;
    if (g_no_halo || g_force_generic) return false;
    if (conv_smallm_ok(B, H, W, C1, C2, Cout, taps)) return false;       // takes a materialised (normalised) input
    if (taps != 9 || (H & 7) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout & 3)) return false;
    const int ks = conv_ksplit(B, H, W, C1 + C2, Cout, taps);
    if (ks != 1 && ks != f43_plan(B, H, W, C1 + C2, Cout, taps)) return false;    // only the Winograd kernel runs split
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return (int64_t)(9 * W + 18) * cmax * 4 < (1LL << 31) && (int64_t)Cout * 9 * (C1 + C2) * 4 < (1LL << 31);
}


bool conv_supports_wino(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    return !g_no_wino && (Cout % 64) == 0 && conv_supports_fused_gn(B, H, W, C1, C2, Cout, taps) &&
           f43_plan(B, H, W, C1 + C2, Cout, taps) >= 1 && (int64_t)Cout * 18 * (C1 + C2) * 4 < (1LL << 31) &&
           // the F(4,3) kernel addresses a whole sample through one buffer descriptor per source tensor
           ((int64_t)H * W + 2 * W + 2) * (C1 > C2 ? C1 : C2) * 4 < (1LL << 31);
}

// The two-dimensional form: whole-K launches on images that give every CU at least two blocks of 16 x 16 pixels x 64
// channels, or one of 32 channels (below that the 1-D kernel's K slices fill the chip better).
static const int g_w2d_min_blocks = getenv("FLOWSE_W2D_MIN_BLOCKS") ? atoi(getenv("FLOWSE_W2D_MIN_BLOCKS")) : 128;
bool conv_w2d_shape_ok(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (g_no_wino_policy || taps != 9 || (H & 15) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout % 64) || B < 1) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return (int64_t)Cout * 24 * (C1 + C2) * 4 < (1LL << 31) && ((int64_t)H * W + 2 * W + 2) * cmax * 4 < (1LL << 31);
}
bool conv_supports_w2d(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    // at least one block per CU: of 64 channels, or -- 128..255 such blocks -- of 32 channels (conv3x3_w2d_kernel<GN, 1>)
    const int64_t b64 = ((int64_t)B * H * W / 256) * (Cout / 64);
    return conv_w2d_shape_ok(B, H, W, C1, C2, Cout, taps) && b64 >= g_w2d_min_blocks;
}

int launch_conv(const ConvArgs& a, hipStream_t s, bool with_reduce) {
    if ((a.C1 & 3) || (a.C2 & 3) || (a.Cout & 3) || (a.bias2 && (a.bias2_stride & 3)) || (a.taps != 1 && a.taps != 9) ||
        a.C1 <= 0 || (a.in2 == nullptr && a.C2 != 0)) {
        set_error("conv: unsupported channel counts C1=%d C2=%d taps=%d", a.C1, a.C2, a.taps);
        return ERR_SHAPE;
    }
    if ((int64_t)a.B * a.H * a.W >= (1LL << 31) / 4) {
        set_error("conv: too many pixels for 32-bit pixel indices");
        return ERR_SHAPE;
    }
    if (a.sc1 || a.sc2 || a.wfrag_sc) {               // folded 1x1 shortcut: the producer / consumer kernel only
        if (a.in_dt == DT_F32 || a.ksplit > 1 || a.out_dt != a.in_dt || !a.wfrag ||
            !conv16_uses_pc(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
            set_error("conv: a folded shortcut needs a launch of conv3x3_pc16_kernel (16-bit storage, 16 x 16-pixel tiles)");
            return ERR_ARG;
        }
        return launch_pc16(a, s);
    }
    if (a.in_dt != DT_F32) {                          // activations stored as bf16 / half
        if (a.ksplit <= 1 && !a.partial && !a.bias2 && !a.stats && a.out_dt == DT_F32 &&
            conv_supports_head4(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
            return launch_head4(a, s);
        if (a.ksplit <= 1 && a.out_dt == a.in_dt && conv16_uses_halo(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
            if (a.wfrag && conv16_uses_pc(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) return launch_pc16(a, s);
            return launch_halo16_any(a, s);
        }
        if (a.wfrag && a.ksplit <= 1 && !a.partial && !a.gn.mean && (a.out_dt == a.in_dt || a.out_dt == DT_F32) &&
            conv16_smallm_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
            return launch_smallm16b(a, s);
        if (a.ksplit > 1 && !a.partial) {
            set_error("conv: split-K needs a partial buffer");
            return ERR_ARG;
        }
        const int rc = launch_flat16(a, s);
        if (rc != OK || a.ksplit <= 1 || !with_reduce) return rc;
        return launch_splitk_reduce(a, s);
    }
    if (a.ksplit <= 1 && !a.partial && !a.bias2 && !a.stats && conv_supports_head4(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
        return launch_head4(a, s);
    if (a.wsm && a.ksplit <= 1 && !a.partial && !a.gn.mean && a.out_dt == DT_F32 &&
        conv_smallm_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
        return launch_smallm(a, s);
    if (a.wsm && a.ksplit <= 1 && !a.partial && !a.gn.mean && a.out_dt == DT_F32 &&
        conv1x1_stream_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
        return launch_1x1_stream(a, s);
    if (a.ksplit <= 1 && conv_supports_fused_gn(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        if (a.wq && (a.Cout % 128) == 0) {
            if (a.terms == 3 && !a.wq_f16) return launch_halo_bf16x3(a, s);
            if (a.terms == 1) return launch_halo16_any(a, s);
            set_error("conv: 16-bit path needs terms = 1 (bf16 / f16) or 3 (bf16 only)");
            return ERR_ARG;
        }
        if (a.wino2 && conv_w2d_shape_ok(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
            if (a.stats && a.stats_nblk != a.H * a.W / 64) {
                set_error("conv: the 2-D Winograd kernel writes its statistics in blocks of 64 pixels (stats_nblk=%d)", a.stats_nblk);
                return ERR_ARG;
            }
            return launch_w2d(a, s);
        }
        if (a.wino && conv_supports_wino(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) return launch_f43(a, s);
        return launch_halo_fp32(a, s);
    }
    if (a.ksplit > 1 && !a.partial) {
        set_error("conv: split-K needs a partial buffer");
        return ERR_ARG;
    }
    if (a.ksplit > 1 && a.wino && conv_supports_wino(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps) &&
        a.ksplit == f43_plan(a.B, a.H, a.W, a.C1 + a.C2, a.Cout, a.taps)) {
        const int rc = launch_f43(a, s);              // gridDim.y = ksplit slices of chunks, raw partial tiles
        if (rc != OK || !with_reduce) return rc;
        return launch_splitk_reduce(a, s);
    }
    if (a.gn.mean) {
        set_error("conv: fused GroupNorm input requested for a shape the halo kernel does not cover");
        return ERR_ARG;
    }
    const int rc = launch_flat_fp32(a, s);
    if (rc != OK || a.ksplit <= 1 || !with_reduce) return rc;
    return launch_splitk_reduce(a, s);
}

}  // namespace flowse
