// 3x3 / 1x1 convolution as an implicit GEMM on the CDNA4 matrix cores.
//
//   out[m][n] = ( sum_{tap, ci} A[m][tap, ci] * Wp[n][tap][ci] + bias[n] + bias2[b(m)][n] + res[m][n] ) * scale
//
//   m = NHWC pixel (b, y, x), n = output channel, K = taps * Cin with the channel axis contiguous in both
//   operands.  All arithmetic is fp32 (v_mfma_f32_32x32x2_f32; no TF32 on gfx950).  The direct kernels are bitwise
//   an fmaf chain; the production 3x3 kernel uses the F(4,3) Winograd form (half the multiplies, fp32 rounding
//   ~3x the direct sum's).  Optional 16-bit operand modes (bf16x3 / bf16 / fp16) exist for the direct LDS-halo
//   3x3 kernel only.
//
// Replaces (reference): ddpm_conv3x3 / ddpm_conv1x1 (flowmse/backbones/ncsnpp_utils/layers.py:100-124) and
// NIN (layers.py:546-555) as used by ResnetBlockBigGANpp (layerspp.py:245-274), AttnBlockpp (:75-91), the
// progressive-output heads (ncsnpp.py:345-366) incl. the channel concat of ncsnpp.py:337 (two-source A operand),
// the per-sample time-embedding bias of layerspp.py:262-263 (bias2), the (x + h)/sqrt(2) skip (res, scale), the
// GroupNorm + SiLU in front of every ResBlock conv (layerspp.py:246,265; fused into the halo staging) and the
// statistics of the NEXT GroupNorm (fused into the epilogue).
//
// Kernels in this file (dispatch: launch_conv / launch_conv_cin4):
//   conv3x3_f43_kernel         3x3, H % 8 == 0, W % 16 == 0, Cin % 32 == 0, Cout % 64 == 0, >= 256 blocks: F(4,3)
//                              Winograd along the vertical axis; 8x16-pixel tile x 128 channels (2 blocks per CU, layers
//                              with >= 512 such blocks) or x 64 channels (3 blocks per CU); the input halo of a
//                              32-channel chunk double-buffered in LDS (optional fused GroupNorm+SiLU on the way in),
//                              transformed weights fetched from L2 in MFMA fragment order.  The production kernel
//                              (~75 % of GPU time).  conv3x3_wino_kernel: the F(2,3) form.
//   conv3x3_halo_kernel        the direct form of the same tiling (per-tap weight tile through LDS): narrow heads
//                              (Cout = 4), FLOWSE_NO_WINOGRAD=1, and the base of
//   conv3x3_halo_bf16_kernel   the bf16x3 operand mode and small 16-bit launches; conv3x3_halo16_kernel: the 16-bit
//                              storage modes (v_mfma_f32_32x32x16_*, 8x16 or 16x16 pixel tile x 128 channels, output
//                              straight from the accumulators); conv_flat16_kernel: their flat 1x1 / small 3x3 form.
//   conv_mfma_fast_kernel      flat pixel tiling, A gathered per tap through a window buffer descriptor: 1x1 convs
//                              and 3x3 on small images; split-K (gridDim.y) + splitk_reduce[_stats] for tiny images.
//   conv_mfma_kernel           generic fallback (any channel count multiple of 4, chunks straddling the concat).
//   conv3x3_cin4_mfma_kernel   the 4 -> 128 input layer on the matrix cores (A operand = a neighbour pixel's float4).
//   conv_cin4_kernel           direct VALU conv for 4 input channels (Combine 1x1, small / odd input layers).
// Common to the LDS-staged kernels: block = 4 waves; LDS rows of 36 floats (conflict-free ds_read_b128); each lane
// reads 4 consecutive k of its row once and feeds 4 successive MFMAs (lanes 0-31 carry k = 8j+e, lanes 32-63 carry
// k = 8j+4+e, for A and B alike); epilogue through LDS (bias, temb bias, residual, scale, GroupNorm partials).
// Environment switches (read once): FLOWSE_NO_WINOGRAD, FLOWSE_WINOGRAD=f23, FLOWSE_NO_HALO_CONV,
// FLOWSE_FORCE_GENERIC_CONV, FLOWSE_NO_CIN4_MFMA -- test / A-B hooks, never needed for correctness.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <string>

#include "common.h"

namespace flowse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;          // channels per K step
constexpr int LDS_ROW = 36;     // floats per LDS tile row (KC + 4 pad)
bool conv_small_m(int64_t M, int Cout);
bool conv_splitk_in_launch();

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, each with its own
// L2); remapping so that every XCD walks a CONTIGUOUS range of tiles keeps the rows shared by vertically
// adjacent pixel tiles (the 3x3 halo) and the N tiles of one pixel tile in one L2.  Bijective for any grid size.
// Placement is a speed matter only -- nothing depends on it for correctness.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device setting: remember on which devices of this process the
// kernel has been configured (one process normally drives one GPU, but nothing here relies on that).
template <auto Kernel>
static int allow_lds(size_t bytes) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    FLOWSE_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        FLOWSE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes));
        done.fetch_or(bit, std::memory_order_release);
    }
    return OK;
}

// ---- shared epilogue.  The accumulators go through LDS (C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) so that bias / per-sample bias / residual are read and the result
// is written as coalesced float4 rows of the NHWC output.  Precondition: all waves are past their last LDS read.
// OT = storage type of `res` and `out` (float, or bf16 / half in the 16-bit modes); the statistics are taken over the
// values as rounded to OT, i.e. over what the consumer of `out` will read.
template <int WM, int WN, int TM, int TN, class OT = float, class Scatter>
__device__ __forceinline__ void conv_epilogue_with(const ConvArgs& a, float* smem, int m0, int n0, int M, int HW,
                                                   int split, int rowW, Scatter scatter) {
    const OT* resp = reinterpret_cast<const OT*>(a.res);
    OT* outp = reinterpret_cast<OT*>(a.out);
    // rowW == 0: tile row rr is flat pixel m0 + rr; rowW > 0: the tile is 8 x 16 pixels of an image with row
    // pitch rowW, m0 = its top-left pixel
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = 64 * WM * WN;
    const int tid = threadIdx.x;
    constexpr int CROW = BN + 4;
    static_assert(BM * CROW <= 2 * (BM + BN) * LDS_ROW, "C tile must fit in the staging buffers");
    float* Cs = smem;                          // [BM][CROW]; safe: the last loop iteration ended with a barrier
    constexpr int C4 = BN / 4;                 // float4 per tile row
    constexpr int RPP = NT / C4;               // rows per pass
    constexpr int NR = BM / RPP;               // rows per thread
    const int ec4 = tid % C4, er0 = tid / C4;
    const int n = n0 + ec4 * 4;
    const bool ncol = n < a.Cout;              // Cout % 4 == 0: a quad is entirely inside or outside
    const bool has_b2 = a.bias2 != nullptr && !a.partial, has_res = a.res != nullptr && !a.partial;
    // Residual / per-sample bias quads of this thread's rows are requested FIRST (unconditional loads on clamped
    // addresses), so they are in flight while the accumulators make their trip through LDS.
    float4 rres[NR], rb2[NR];
    if (has_res && ncol) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int rr = er0 + k * RPP;
            const int m = rowW ? m0 + (rr >> 4) * rowW + (rr & 15) : m0 + rr;
            rres[k] = St<OT>::ld4(resp + (m < M ? (int64_t)m * a.Cout : 0) + n);
        }
    }
    if (has_b2 && ncol) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int rr = er0 + k * RPP;
            const int m = rowW ? m0 + (rr >> 4) * rowW + (rr & 15) : m0 + rr;
            rb2[k] = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)(m < M ? m / HW : 0) * a.bias2_stride + n);
        }
    }
    scatter(Cs, CROW);                         // accumulators -> Cs[tile row][channel]
    __syncthreads();
    if (a.partial) {                           // split-K slice: raw partial sums
        if (ncol) {
            float* dst = a.partial + (int64_t)split * M * a.Cout;
            for (int rr = er0; rr < BM; rr += RPP) {
                const int m = rowW ? m0 + (rr >> 4) * rowW + (rr & 15) : m0 + rr;
                if (m >= M) break;
                *reinterpret_cast<float4*>(dst + (int64_t)m * a.Cout + n) =
                    *reinterpret_cast<const float4*>(Cs + rr * CROW + ec4 * 4);
            }
        }
        if (!a.sk_ticket) return;              // two-pass form: splitk_reduce sums the slices and runs the epilogue
        // ---- in-launch reduction.  Release the tile, take a ticket; the block that takes the last one of this tile
        // acquires, sums the slices 0 .. ks-1 in that order (its own included, re-read: the sum is independent of which
        // slice arrived last), applies bias / per-sample bias / residual / scale, stores, and leaves the GroupNorm
        // partial statistics of what it stored -- exactly the layouts the two-pass reduction kernels write.
        if constexpr (BM != 128) return;       // (the in-launch form is written for 128-row tiles; launch_conv never pairs it with others)
        __threadfence();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);            // the C tile is free again
        if (tid == 0) {
            const unsigned ks = gridDim.y;
            const unsigned t = atomicAdd(a.sk_ticket + blockIdx.x, 1u);
            if (t == ks - 1) a.sk_ticket[blockIdx.x] = 0;   // ready for the next launch on the stream
            *flag = (t == ks - 1) ? 1 : 0;
        }
        __syncthreads();
        const int last = *flag;
        if (!last) return;
        __threadfence();
        __syncthreads();                                     // everyone has read the flag: Cs may be overwritten
        const int ks = (int)gridDim.y;
        const int64_t slice = (int64_t)M * a.Cout;
        if (ncol) {
            float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll 2
            for (int k = 0; k < NR; ++k) {
                const int rr = er0 + k * RPP;
                const int m = rowW ? m0 + (rr >> 4) * rowW + (rr & 15) : m0 + rr;
                if (m >= M) break;
                const int64_t i4 = (int64_t)m * a.Cout + n;
                float4 v = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
                for (int s = 1; s < ks; ++s) {               // independent loads: many in flight
                    const float4 t = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                if (a.bias2) {
                    const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)(m / HW) * a.bias2_stride + n);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                if (a.res) {
                    const float4 t = St<OT>::ld4(resp + i4);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
                St<OT>::st4(outp + i4, v);
                if (a.stats) *reinterpret_cast<float4*>(Cs + rr * CROW + ec4 * 4) = St<OT>::rnd4(v);
            }
        }
        if (!a.stats) return;
        __syncthreads();
        // statistics: a thread = (channel, every `NT / BN`-th group of GR tile rows); two passes over the LDS tile (exact
        // mean, then M2).  Flat tiles: groups of sk_group flat pixels (never straddling a sample: sk_group | HW);
        // 8 x 16 tiles: the whole tile is one block of the sample's tile grid.
        const int GR = rowW ? BM : a.sk_group;
        const int c = tid % BN, g0 = tid / BN;
        if (n0 + c < a.Cout) {
            for (int g = g0; g * GR < BM; g += NT / BN) {
                const int r0 = g * GR;
                const int mf = rowW ? m0 : m0 + r0;          // first pixel of the group
                if (mf >= M) break;
                float sum = 0.f;
                for (int r = 0; r < GR; ++r) sum += Cs[(r0 + r) * CROW + c];
                const float mean = sum / (float)GR;
                float m2 = 0.f;
                for (int r = 0; r < GR; ++r) {
                    const float d = Cs[(r0 + r) * CROW + c] - mean;
                    m2 = fmaf(d, d, m2);
                }
                const int bsmp = mf / HW;
                const int rem = mf - bsmp * HW;
                const int blk = rowW ? ((rem / rowW) >> 3) * (rowW >> 4) + ((rem % rowW) >> 4) : rem / GR;
                float* dst = a.stats + (((int64_t)bsmp * a.stats_nblk + blk) * a.Cout + n0 + c) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
        }
        return;
    }
    if (ncol) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + n);
        Stat4 st;
        st.init();
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int rr = er0 + k * RPP;
            const int m = rowW ? m0 + (rr >> 4) * rowW + (rr & 15) : m0 + rr;
            if (m >= M) continue;
            float4 v = *reinterpret_cast<const float4*>(Cs + rr * CROW + ec4 * 4);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (has_b2) { v.x += rb2[k].x; v.y += rb2[k].y; v.z += rb2[k].z; v.w += rb2[k].w; }
            if (has_res) { v.x += rres[k].x; v.y += rres[k].y; v.z += rres[k].z; v.w += rres[k].w; }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            St<OT>::st4(outp + (int64_t)m * a.Cout + n, v);
            st.add(St<OT>::rnd4(v));
        }
        if (a.stats) {
            // Fused GroupNorm statistics of the tile just written (launch guarantees: tile inside one sample, all
            // BM rows valid).  Threads tid and tid+32 of a wave own the same channel quad when C4 == 32; in general
            // threads with equal ec4 are reduced through the free tail of the LDS block.
            float* red = smem + BM * CROW;                       // [NT / C4][C4][8] floats
            st.finish(red + (er0 * C4 + ec4) * 8);                  // every thread covers BM / RPP rows
        }
    }
    if (a.stats) {
        __syncthreads();
        if (n < a.Cout && er0 == 0) {
            const float* red = smem + BM * CROW;
            float acc8[8], nacc = (float)(BM / RPP);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = red[ec4 * 8 + j];
            for (int r = 1; r < RPP; ++r) chan_merge4(nacc, acc8, (float)(BM / RPP), red + (r * C4 + ec4) * 8);
            const int bsmp = m0 / HW;
            int tile = (m0 - bsmp * HW) / BM;
            if (rowW) {                       // 8x16 tiles, row-major over the image
                const int rem = m0 - bsmp * HW;
                tile = ((rem / rowW) >> 3) * (rowW >> 4) + ((rem % rowW) >> 4);
            }
            float* dst = a.stats + (((int64_t)bsmp * a.stats_nblk + tile) * a.Cout + n) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dst[2 * j] = acc8[j];
                dst[2 * j + 1] = acc8[4 + j];
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, class OT = float>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0,
                                              int M, int HW, int split, int rowW = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, kh = lane >> 5;
    conv_epilogue_with<WM, WN, TM, TN, OT>(a, smem, m0, n0, M, HW, split, rowW, [&](float* Cs, int CROW) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    Cs[row * CROW + (wn * TN + jn) * 32 + li] = acc[i][jn][r];
                }
    });
}

// ---- output stage for one [128 tile rows = 8 x 16 pixels][64 channels] fp32 tile sitting in LDS (row pitch CROW floats).
// Thread = (row lane, 16-byte channel group): bias + per-sample bias + residual + scale, ONE rounding to OT, 16-byte
// loads / stores (4 floats or 8 x 16 bit per lane: half the memory instructions of an 8-byte form for the 16-bit types),
// and the GroupNorm partial statistics (mean, M2 per channel over the 128 pixels) of exactly what was stored.  The
// statistics are pivoted per thread and merged with Chan's formula for EQUAL counts -- lane shuffles inside a wave, one
// LDS hop across the four waves -- so there is no division and no serial merge loop.  `red`: 4 x 64 x 2 floats of LDS
// scratch.  All 256 threads must call; Cs must be complete (barrier before) and may be overwritten after the call's
// last barrier.
template <class OT>
__device__ __forceinline__ void tile128x64_out(const ConvArgs& a, const float* Cs, int CROW, float* red, int m_tl, int W,
                                               int n_base, int bsmp, int tile) {
    constexpr int CPT = Vec16<OT>::N, TPR = 64 / CPT, RPP = 256 / TPR, NRW = 128 / RPP;
    const int tid = threadIdx.x;
    const int ec = tid % TPR, er0 = tid / TPR;
    const int n = n_base + ec * CPT;
    const OT* resb = reinterpret_cast<const OT*>(a.res) + (int64_t)m_tl * a.Cout + n;
    OT* outb = reinterpret_cast<OT*>(a.out) + (int64_t)m_tl * a.Cout + n;
    const bool has_res = a.res != nullptr;
    float bq[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) bq[j] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) bq[j] = a.bias[n + j];
    }
    if (a.bias2) {
        const float* b2 = a.bias2 + (int64_t)bsmp * a.bias2_stride + n;
#pragma unroll
        for (int j = 0; j < CPT; ++j) bq[j] += b2[j];
    }
    float piv[CPT], s1[CPT], s2[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) { piv[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    constexpr int GRP = NRW < 4 ? NRW : 4;               // rows per batch: their residuals are requested together
#pragma unroll
    for (int g0 = 0; g0 < NRW; g0 += GRP) {
        float rres[GRP][CPT];
        int roff[GRP];
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
            const int rr = er0 + (g0 + k) * RPP;
            roff[k] = ((rr >> 4) * W + (rr & 15)) * a.Cout;
            if (has_res) ld16<OT>(resb + roff[k], rres[k]);
        }
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
            const int rr = er0 + (g0 + k) * RPP;
            float v[CPT];
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(Cs + rr * CROW + ec * CPT + q * 4);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                float t = v[j] + bq[j];
                if (has_res) t += rres[k][j];
                v[j] = t * a.scale;
            }
            st16_round<OT>(outb + roff[k], v);
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                if (g0 + k == 0) piv[j] = v[j];
                const float d = v[j] - piv[j];
                s1[j] += d;
                s2[j] = fmaf(d, d, s2[j]);
            }
        }
    }
    if (!a.stats) return;
    // per-thread (mean, M2) over NRW rows, then equal-count merges: lanes with equal `ec` inside the wave, then waves
    float cnt = (float)NRW;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float mean = piv[j] + s1[j] * (1.f / NRW);
        const float m2 = fmaxf(s2[j] - s1[j] * s1[j] * (1.f / NRW), 0.f);
        piv[j] = mean;
        s2[j] = m2;
    }
#pragma unroll
    for (int off = TPR; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const float mo = __shfl_xor(piv[j], off), qo = __shfl_xor(s2[j], off);
            const float d = mo - piv[j];
            s2[j] = s2[j] + qo + d * d * (0.5f * cnt);
            piv[j] = 0.5f * (piv[j] + mo);
        }
        cnt *= 2.f;
    }
    const int wave = tid >> 6;
    if ((tid & 63) < TPR) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            red[(wave * 64 + ec * CPT + j) * 2] = piv[j];
            red[(wave * 64 + ec * CPT + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    if (tid < 64) {                                       // one thread per channel: ((w0 + w1) + (w2 + w3)), 32 pixels each
        float m01, q01, m23, q23;
        {
            const float ma = red[tid * 2], qa = red[tid * 2 + 1], mb = red[(64 + tid) * 2], qb = red[(64 + tid) * 2 + 1];
            const float d = mb - ma;
            m01 = 0.5f * (ma + mb);
            q01 = qa + qb + d * d * 16.f;
        }
        {
            const float ma = red[(128 + tid) * 2], qa = red[(128 + tid) * 2 + 1], mb = red[(192 + tid) * 2],
                        qb = red[(192 + tid) * 2 + 1];
            const float d = mb - ma;
            m23 = 0.5f * (ma + mb);
            q23 = qa + qb + d * d * 16.f;
        }
        const float d = m23 - m01;
        float* dst = a.stats + (((int64_t)bsmp * a.stats_nblk + tile) * a.Cout + n_base + tid) * 2;
        dst[0] = 0.5f * (m01 + m23);
        dst[1] = q01 + q23 + d * d * 32.f;
    }
}

// ---- output stage straight from the accumulators of a [MT x 128 pixels][128 channels] block tile held as 4 waves x
// MT x 2 x 2 tiles of the 32x32 MFMA, 16-bit storage.  In the C/D layout a lane owns ONE output channel (col = lane & 31)
// at 16 pixels of every tile, so the per-channel work -- bias, the GroupNorm partial statistics -- is in-lane arithmetic,
// and no trip through LDS is needed to reach the NHWC rows: neighbouring lanes (channels c, c + 1) swap one value per
// pixel pair through DPP, after which the even lane holds both channels at pixel r and the odd lane both at pixel r + 1.
// Each then loads / stores ONE dword (two 16-bit channels); the 16 lane pairs of a half-wave cover 64 contiguous bytes
// of an output pixel.  Per thread and tile: 8 loads (residual), 8 stores, ~20 VALU per pixel pair -- against two
// block-wide passes through a 35 KB LDS tile, four barriers and half the waves idle in the staged form.
// Statistics: pivoted (mean, M2) over the lane's 16 values per channel, equal-count merges lane pair -> half-waves ->
// the two waves that share the channels (through `red`, [2][MT][128][2] floats of LDS).  All 256 threads must call;
// every wave must be past its last read of the LDS that `red` overlays.
template <class OT, int MT>
__device__ __forceinline__ void halo16_out_direct(const ConvArgs& a, f32x16 (&acc)[MT][2][2], float* red, int m_tl, int W,
                                                  int n0, int bsmp, int tile0, int tiles_x) {
    static_assert(sizeof(OT) == 2, "16-bit storage only");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kh = lane >> 5;
    const bool odd = li & 1;
    const int Cout = a.Cout;
    const bool has_res = a.res != nullptr;
    const unsigned* resw = reinterpret_cast<const unsigned*>(reinterpret_cast<const OT*>(a.res) + (int64_t)m_tl * Cout);
    unsigned* outw = reinterpret_cast<unsigned*>(reinterpret_cast<OT*>(a.out) + (int64_t)m_tl * Cout);
    // dword offset of this lane's channel pair at tile pixel 0, plus its pixel inside the 4-pixel group (odd + 4 kh)
    const int lane_off = ((odd ? 1 : 0) + 4 * kh) * (Cout >> 1) + ((n0 + wn * 64 + (li & ~1)) >> 1);
    float bq[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wn * 64 + j * 32 + (li & ~1);
        bq[j][0] = a.bias ? a.bias[c] : 0.f;
        bq[j][1] = a.bias ? a.bias[c + 1] : 0.f;
        if (a.bias2) {
            const float* b2 = a.bias2 + (int64_t)bsmp * a.bias2_stride + c;
            bq[j][0] += b2[0];
            bq[j][1] += b2[1];
        }
    }
    const float scale = a.scale;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        float piv[2][2], s1[2][2], s2[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) { piv[j][e] = 0.f; s1[j][e] = 0.f; s2[j][e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // tile row of pixel pair p: rho = 2 (p & 1) + 8 ((p >> 1) & 1) + 16 (p >> 2) [+ odd + 4 kh]: image row p >> 2
            const int row_off = ((t * 8 + 2 * (wm * 2 + i)) * W) * (Cout >> 1) + lane_off;
            const int row_step = W * (Cout >> 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned rres[8];
                int off[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    off[p] = row_off + (p >> 2) * row_step + (2 * (p & 1) + 8 * ((p >> 1) & 1)) * (Cout >> 1) + j * 16;
                    if (has_res) rres[p] = resw[off[p]];
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const float lo = acc[t][i][j][2 * p], hi = acc[t][i][j][2 * p + 1];
                    const float send = odd ? lo : hi;
                    const float recv = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
                    float v0 = (odd ? recv : lo) + bq[j][0];          // channel pair (c, c + 1) at this lane's pixel
                    float v1 = (odd ? hi : recv) + bq[j][1];
                    if (has_res) {
                        float r0, r1;
                        St<OT>::unpack2(rres[p], r0, r1);
                        v0 += r0;
                        v1 += r1;
                    }
                    v0 *= scale;
                    v1 *= scale;
                    const unsigned w = St<OT>::pack2(v0, v1);
                    outw[off[p]] = w;
                    St<OT>::unpack2(w, v0, v1);                        // statistics of what was stored
                    if (i == 0 && p == 0) { piv[j][0] = v0; piv[j][1] = v1; }
                    const float d0 = v0 - piv[j][0], d1 = v1 - piv[j][1];
                    s1[j][0] += d0; s2[j][0] = fmaf(d0, d0, s2[j][0]);
                    s1[j][1] += d1; s2[j][1] = fmaf(d1, d1, s2[j][1]);
                }
            }
        }
        if (!a.stats) continue;
        // 16 values per lane and channel -> lane pair (32) -> half-waves (64 = this wave's pixels of the sub-tile)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float mean = piv[j][e] + s1[j][e] * (1.f / 16);
                float m2 = fmaxf(s2[j][e] - s1[j][e] * s1[j][e] * (1.f / 16), 0.f);
                {
                    const float mo = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mean), 0xB1, 0xF, 0xF, false));
                    const float qo = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m2), 0xB1, 0xF, 0xF, false));
                    const float d = mo - mean;
                    m2 = m2 + qo + d * d * 8.f;
                    mean = 0.5f * (mean + mo);
                }
                {
                    const float mo = __shfl_xor(mean, 32), qo = __shfl_xor(m2, 32);
                    const float d = mo - mean;
                    m2 = m2 + qo + d * d * 16.f;
                    mean = 0.5f * (mean + mo);
                }
                if (kh == 0 && !odd) {
                    float* dst = red + (((wm * MT + t) * 128) + wn * 64 + j * 32 + li + e) * 2;
                    dst[0] = mean;
                    dst[1] = m2;
                }
            }
    }
    if (!a.stats) return;
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float ma = red[((0 * MT + t) * 128 + tid) * 2], qa = red[((0 * MT + t) * 128 + tid) * 2 + 1];
            const float mb = red[((1 * MT + t) * 128 + tid) * 2], qb = red[((1 * MT + t) * 128 + tid) * 2 + 1];
            const float d = mb - ma;
            float* dst = a.stats + (((int64_t)bsmp * a.stats_nblk + tile0 + t * tiles_x) * Cout + n0 + tid) * 2;
            dst[0] = 0.5f * (ma + mb);
            dst[1] = qa + qb + d * d * 32.f;
        }
    }
}

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

// ---------------------------------------------------------------------------------------------------
// Fast path for the production shapes: C1 % 32 == 0 and C2 % 32 == 0, so every K step (tap, 32-channel chunk)
// lies in ONE source tensor and is full.  Everything that varies per step is wave-uniform and lives in SGPRs
// (buffer descriptor + soffset); per lane only a 9-bit tap-validity mask and one byte offset per gathered row
// are kept.  Out-of-image taps (conv zero padding), rows past M and channels past Cout are redirected to an
// out-of-range buffer offset, for which the hardware returns 0 -- no masking VALU, no branches, ~10 VALU per
// step next to 64 MFMAs.  The block addresses its inputs through a window descriptor based at pixel
// m0 - W - 1, so offsets stay small whatever the tensor size.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

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
#ifdef FLOWSE_PROBE_FLAT_NOA     /* measurement probes (results garbage): no activation / no weight loads in the flat kernel */
            ra[R][q] = u32x4{vo, soff_a, (unsigned)s, 0x3f800000u};
#else
            ra[R][q] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, vo, soff_a, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, vo, soff_a, 0);
#endif
        }
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q)
#ifdef FLOWSE_PROBE_FLAT_NOB
            rb[R][q] = u32x4{bvo[q], soff_b, (unsigned)s, 0x3f800000u};
#else
            rb[R][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo[q], soff_b, 0);
#endif
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
#ifdef FLOWSE_PROBE_FLAT_NOEPI   /* measurement probe: no output stage */
    {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) a.out[m0] = t;
        return;
    }
#endif
    conv_epilogue<WM, WN, TM, TN, OT>(a, acc, smem, m0, n0, M, HW, split);
}

// One channel quad of an activation tensor through a buffer descriptor, widened to fp32 (bit patterns).  `voff` / `soff`
// are BYTE offsets (element index * sizeof(ST)); out-of-range offsets return zeros for every type.
template <class ST>
__device__ __forceinline__ u32x4 buf_ld_quad(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (std::is_same<ST, float>::value) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    } else {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        u32x4 o;
        if constexpr (std::is_same<ST, bf16_t>::value) {
            o.x = t.x << 16; o.y = t.x & 0xffff0000u; o.z = t.y << 16; o.w = t.y & 0xffff0000u;
        } else {
            // (element-wise bit casts: a bit cast of the dword to a 2 x half vector was miscompiled by hipcc 7.2 here)
            const _Float16 e0 = __builtin_bit_cast(_Float16, (unsigned short)(t.x & 0xffffu));
            const _Float16 e1 = __builtin_bit_cast(_Float16, (unsigned short)(t.x >> 16));
            const _Float16 e2 = __builtin_bit_cast(_Float16, (unsigned short)(t.y & 0xffffu));
            const _Float16 e3 = __builtin_bit_cast(_Float16, (unsigned short)(t.y >> 16));
            o.x = __float_as_uint((float)e0); o.y = __float_as_uint((float)e1);
            o.z = __float_as_uint((float)e2); o.w = __float_as_uint((float)e3);
        }
        return o;
    }
}

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
// x * sigmoid(x) with v_exp_f32 / v_rcp_f32 (about 2 ulp)
__device__ __forceinline__ float fast_silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// GroupNorm affine (+ SiLU) of one staged channel quad, written on float2 halves so that hipcc emits the packed
// fp32 VALU forms (v_pk_add / v_pk_fma / v_pk_mul); only v_exp_f32 / v_rcp_f32 stay scalar.  `keep` = 0 zeroes the
// quad (halo pixel outside the image: zero padding applies AFTER the activation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef FLOWSE_NOPK
template <int GN>
__device__ __forceinline__ u32x4 gn_quad(u32x4 raw, float4 mu, float4 sc, float4 be, bool keep) {
    float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
    const float m[4] = {mu.x, mu.y, mu.z, mu.w}, s[4] = {sc.x, sc.y, sc.z, sc.w}, b[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = fmaf(v[e] - m[e], s[e], b[e]);
        if (GN == 2) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v[e] * -1.44269504088896341f));
    }
    u32x4 o;
    o.x = keep ? __float_as_uint(v[0]) : 0u; o.y = keep ? __float_as_uint(v[1]) : 0u;
    o.z = keep ? __float_as_uint(v[2]) : 0u; o.w = keep ? __float_as_uint(v[3]) : 0u;
    return o;
}
#else
template <int GN>
__device__ __forceinline__ u32x4 gn_quad(u32x4 raw, float4 mu, float4 sc, float4 be, bool keep) {
    f32x2 lo = {__uint_as_float(raw.x), __uint_as_float(raw.y)}, hi = {__uint_as_float(raw.z), __uint_as_float(raw.w)};
    const f32x2 mlo = {mu.x, mu.y}, mhi = {mu.z, mu.w}, slo = {sc.x, sc.y}, shi = {sc.z, sc.w};
    const f32x2 blo = {be.x, be.y}, bhi = {be.z, be.w};
    lo = __builtin_elementwise_fma(lo - mlo, slo, blo);
    hi = __builtin_elementwise_fma(hi - mhi, shi, bhi);
    if (GN == 2) {
        const f32x2 nl2e = {-1.44269504088896341f, -1.44269504088896341f}, one = {1.f, 1.f};
        f32x2 el = lo * nl2e, eh = hi * nl2e;
        el.x = __builtin_amdgcn_exp2f(el.x); el.y = __builtin_amdgcn_exp2f(el.y);
        eh.x = __builtin_amdgcn_exp2f(eh.x); eh.y = __builtin_amdgcn_exp2f(eh.y);
        el += one; eh += one;
        el.x = __builtin_amdgcn_rcpf(el.x); el.y = __builtin_amdgcn_rcpf(el.y);
        eh.x = __builtin_amdgcn_rcpf(eh.x); eh.y = __builtin_amdgcn_rcpf(eh.y);
        lo *= el; hi *= eh;
    }
    u32x4 o;
    o.x = keep ? __float_as_uint(lo.x) : 0u; o.y = keep ? __float_as_uint(lo.y) : 0u;
    o.z = keep ? __float_as_uint(hi.x) : 0u; o.w = keep ? __float_as_uint(hi.y) : 0u;
    return o;
}

#endif

#ifndef FLOWSE_HTAP
#define FLOWSE_HTAP 1
#endif
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

// test hook: FLOWSE_FORCE_GENERIC_CONV=1 routes every shape through the generic gather kernel
static const bool g_force_generic = getenv("FLOWSE_FORCE_GENERIC_CONV") != nullptr;

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
                      (int64_t)a.Cout * a.taps * (a.C1 + a.C2) * 4 < (1LL << 31) && !g_force_generic;
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

// out = (sum_s partial[s] + bias + bias2 + res) * scale, float4 streams; grid (ceil(HW * Cout/4 / 256), B)
template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a) {
    const unsigned Q = a.Cout >> 2;
    const unsigned HW = a.H * a.W;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= HW * Q) return;
    const int b = blockIdx.y;
    const int n = (idx % Q) * 4;
    const int64_t i4 = ((int64_t)b * HW * Q + idx) * 4;            // float offset of this quad
    const int64_t slice = (int64_t)a.B * HW * a.Cout;
    float4 v = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
    for (int s = 1; s < a.ksplit; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
#pragma unroll 4
    for (int s = 0; s < a.ksplit2; ++s) {                 // the shortcut conv's slices (ConvArgs::partial2)
        const float4 t = *reinterpret_cast<const float4*>(a.partial2 + s * slice + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias_x) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias_x + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias2) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.res) {
        const float4 t = St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + i4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
    St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, v);
}

// Same reduction, organised like gn_stats (grid (HW / PB, B); a thread owns one channel quad and strides over the
// block's PB pixels) so that it can also emit the GroupNorm partial sums of the tensor it writes:
// stats[((b * nblk + blk) * Cout + c) * 2 + {0,1}].
// pixels per block of the split-K reduction: small images want many blocks (parallelism), larger ones few
// partials (every partial is later read by gn_finalize)
static inline int sk_pixels_per_block(int HW) { return HW <= 8 ? HW : HW <= 1024 ? 8 : HW <= 4096 ? 32 : 64; }

template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(ConvArgs a, int PB) {
    __shared__ float red[256 * 8];
    const int Q = a.Cout >> 2, PR = 256 / Q;
    const int HW = a.H * a.W;
    const int tid = threadIdx.x;
    const int pr = tid / Q, cq = tid - pr * Q;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int n = cq * 4;
    const int64_t slice = (int64_t)a.B * HW * a.Cout;
    Stat4 st;
    st.init();
    if (pr < PR) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + n);
        if (a.bias2) {
            const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
            bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
        }
        if (a.bias_x) {
            const float4 t = *reinterpret_cast<const float4*>(a.bias_x + n);
            bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
        }
        for (int p = blk * PB + pr; p < (blk + 1) * PB; p += PR) {
            const int64_t i4 = ((int64_t)b * HW + p) * a.Cout + n;
            float4 v = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
            for (int s = 1; s < a.ksplit; ++s) {           // independent loads: keep many in flight
                const float4 t = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
#pragma unroll 4
            for (int s = 0; s < a.ksplit2; ++s) {          // the shortcut conv's slices (ConvArgs::partial2)
                const float4 t = *reinterpret_cast<const float4*>(a.partial2 + s * slice + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (a.res) {
                const float4 t = St<OT>::ld4(reinterpret_cast<const OT*>(a.res) + i4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale;
            St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, v);
            st.add(St<OT>::rnd4(v));
        }
    }
    float* mine = red + tid * 8;
    st.finish(mine);
    __syncthreads();
    if (pr == 0) {
        float acc8[8], nacc = (float)st.n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = mine[j];
        for (int r = 1; r < PR; ++r) {
            const int cnt = r < PB ? (PB - r + PR - 1) / PR : 0;         // pixels lane r visited
            chan_merge4(nacc, acc8, (float)cnt, red + (r * Q + cq) * 8);
        }
        float* dst = a.stats + (((int64_t)b * a.stats_nblk + blk) * a.Cout + n) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[2 * j] = acc8[j];
            dst[2 * j + 1] = acc8[4 + j];
        }
    }
}

// Reduction fused with the consuming GroupNorm (see launch_splitk_reduce_gn in common.h).  grid (G, B), 256 threads;
// a thread owns up to RG_MAXQ channel quads of the group's H*W x cpg elements in REGISTERS (no second pass over the
// slices, no LDS tile), the group's sum / sum of squares are reduced in fp64 (exact products of fp32 values; the
// subtraction mean^2 loses log2(mean^2 / var) of 53 bits -- see gn_group_stats in norm.hip).
constexpr int RG_MAXQ = 32;
bool conv_reduce_gn_ok(int B, int HW, int Cout) {
    static const bool off = getenv("FLOWSE_NO_REDUCE_GN") != nullptr;
    const int G = Cout / 4 < 32 ? Cout / 4 : 32;
    if (off || G <= 0 || (Cout % G) != 0) return false;
    // one block per (group, sample): with fewer than ~128 blocks the launch is slower than the two it replaces -- measured
    // at [1,1,256,256] (32 blocks, each pulling up to 512 KB of slices through one CU): 8.27 k vs 8.67 k frames/s
    if ((int64_t)B * G < 128) return false;
    const int cpg = Cout / G;
    return (cpg & 3) == 0 && (int64_t)HW * (cpg / 4) <= 256 * RG_MAXQ;
}

template <class OT>
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(ConvArgs a, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               int apply, float* __restrict__ gn_mean,
                                                               float* __restrict__ gn_scale) {
    __shared__ double wsum[8];
    const int G = gridDim.x, g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int HW = a.H * a.W, Cout = a.Cout, cpg = Cout / G, qpg = cpg >> 2;
    const int items = HW * qpg;                          // channel quads of this (sample, group)
    const int64_t slice = (int64_t)a.B * HW * Cout;
    float4 v[RG_MAXQ];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < RG_MAXQ; ++k) {
        const int it = tid + k * 256;
        if (it < items) {
            const int p = it / qpg, q = it - p * qpg;
            const int n = g * cpg + q * 4;
            const int64_t i4 = ((int64_t)b * HW + p) * Cout + n;
            float4 t = *reinterpret_cast<const float4*>(a.partial + i4);
#pragma unroll 8
            for (int s = 1; s < a.ksplit; ++s) {         // independent loads: keep many in flight
                const float4 u = *reinterpret_cast<const float4*>(a.partial + s * slice + i4);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (a.bias) {
                const float4 u = *reinterpret_cast<const float4*>(a.bias + n);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (a.bias2) {
                const float4 u = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + n);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            t.x *= a.scale; t.y *= a.scale; t.z *= a.scale; t.w *= a.scale;
            if (!apply) {                                // the consumer reads this tensor: statistics of what is stored
                St<OT>::st4(reinterpret_cast<OT*>(a.out) + i4, t);
                t = St<OT>::rnd4(t);
            }
            v[k] = t;
            s1 += ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
            s2 += ((double)t.x * t.x + (double)t.y * t.y) + ((double)t.z * t.z + (double)t.w * t.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((tid & 63) == 0) {
        wsum[tid >> 6] = s1;
        wsum[4 + (tid >> 6)] = s2;
    }
    __syncthreads();
    s1 = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    s2 = (wsum[4] + wsum[5]) + (wsum[6] + wsum[7]);
    const double N = (double)HW * cpg;
    const double mu = s1 / N;
    const double var = fmax(s2 / N - mu * mu, 0.0);
    const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (!apply) {
        if (tid < cpg) {
            const int c = g * cpg + tid;
            gn_mean[(int64_t)b * Cout + c] = mean;
            gn_scale[(int64_t)b * Cout + c] = rstd * gamma[c];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < RG_MAXQ; ++k) {
        const int it = tid + k * 256;
        if (it < items) {
            const int p = it / qpg, q = it - p * qpg;
            const int n = g * cpg + q * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + n), be = *reinterpret_cast<const float4*>(beta + n);
            float4 t = v[k];
            t.x = fmaf(t.x - mean, rstd * ga.x, be.x);
            t.y = fmaf(t.y - mean, rstd * ga.y, be.y);
            t.z = fmaf(t.z - mean, rstd * ga.z, be.z);
            t.w = fmaf(t.w - mean, rstd * ga.w, be.w);
            if (silu) {
                t.x = t.x / (1.f + expf(-t.x)); t.y = t.y / (1.f + expf(-t.y));
                t.z = t.z / (1.f + expf(-t.z)); t.w = t.w / (1.f + expf(-t.w));
            }
            St<OT>::st4(reinterpret_cast<OT*>(a.out) + ((int64_t)b * HW + p) * Cout + n, t);
        }
    }
}

int launch_splitk_reduce_gn(const ConvArgs& a, const float* gamma, const float* beta, float eps, int silu, int apply,
                            float* gn_mean, float* gn_scale, hipStream_t s) {
    const int HW = a.H * a.W;
    if (!a.partial || a.ksplit < 1 || a.res || !conv_reduce_gn_ok(a.B, HW, a.Cout) || !gamma || (apply && !beta) ||
        (!apply && (!gn_mean || !gn_scale))) {
        set_error("splitk_reduce_gn: unsupported arguments (HW=%d Cout=%d ksplit=%d)", HW, a.Cout, a.ksplit);
        return ERR_ARG;
    }
    const int G = a.Cout / 4 < 32 ? a.Cout / 4 : 32;
    FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_gn_kernel<OT>, dim3(G, a.B), dim3(256), 0, s, a, gamma,
                                                      beta, eps, silu, apply, gn_mean, gn_scale));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// Split-K policy: images so small that the 128x128 tiling yields < 256 blocks (one per CU) are sliced along K
// until ~512 blocks exist, keeping >= 4 K steps per slice.
// Measured on MI355X (round 2): the in-launch form removes 75 (B = 8) / 111 (B = 1) launches per network evaluation
// and is SLOWER -- 137.4 vs 122.7 ms per step at [8,1,256,256], 44.5 vs 33.4 ms at [1,1,256,256]: the last slice of a
// tile pulls ks x 64 KB through ONE CU's L1 (>= 8 us at 64 B/clk) after its own K loop, while the two-pass kernel
// spreads the same bytes over every CU in ~4 us.  So the two-pass form stays the default; FLOWSE_SPLITK_IN_LAUNCH=1
// selects the in-launch form (same results to the last bit of the summation order, tests/test_gpu_model.py).
bool conv_splitk_in_launch() {
    static const bool on = getenv("FLOWSE_SPLITK_IN_LAUNCH") != nullptr;
    return on;
}
int conv_splitk_stats_group(int HW) { return sk_pixels_per_block(HW); }

int conv_fused_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps) {
    const int HW = H * W;
    if (Cout & 3) return 0;
    if (conv_ksplit(B, H, W, Cin, Cout, taps) != 1) {      // statistics come from the split-K reduction
        if (conv_splitk_in_launch() && conv_splitk_is_wino(B, H, W, Cin, Cout, taps)) return HW / 128;   // per 8 x 16 tile
        const int PB = sk_pixels_per_block(HW);
        return ((HW % PB) == 0 && Cout / 4 <= 256) ? HW / PB : 0;
    }
    return (HW % 128) == 0 ? HW / 128 : 0;
}

static const bool g_no_wino_policy = getenv("FLOWSE_NO_WINOGRAD") != nullptr || getenv("FLOWSE_NO_HALO_CONV") != nullptr ||
                                     getenv("FLOWSE_FORCE_GENERIC_CONV") != nullptr;

// Winograd plan for a 3x3 shape: 0 = not a Winograd shape (or too small even when sliced), 1 = the Winograd halo
// kernel runs over the whole K, >= 2 = it runs split over that many slices of 32-channel chunks (deterministic
// two-pass reduction as for the flat kernel).  The Winograd kernels tile N by 64, so 256 (pixel tile, channel
// block) pairs -- one per CU -- already beat slicing K; below that, K is cut until ~512 blocks exist while every
// slice keeps at least two chunks.  Only the F(4,3) kernel knows how to run a slice.
static int wino_plan(int B, int H, int W, int Cin, int Cout, int taps) {
    if (g_no_wino_policy || taps != 9 || (H & 7) || (W & 15) || (Cin % KC) || (Cout % 64)) return 0;
    const int64_t blocks = ((int64_t)B * H * W / 128) * (Cout / 64);
    if (blocks >= 256) return 1;
    const int nchunks = Cin / KC;
    if (!conv_wino_default_f43() || blocks < 16 || nchunks < 4) return 0;
    int64_t ks = (512 + blocks - 1) / blocks;
    if (ks > nchunks / 2) ks = nchunks / 2;
    const int per = (int)((nchunks + ks - 1) / ks);
    ks = (nchunks + per - 1) / per;                   // every slice non-empty
    return ks >= 2 ? (int)ks : 0;
}

int conv_ksplit(int B, int H, int W, int Cin, int Cout, int taps) {
    if (conv_supports_head4(B, H, W, Cin, 0, Cout, taps)) return 1;     // 4-channel heads: dedicated kernel
    const int64_t M = (int64_t)B * H * W;
    const int bn = Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128;       // N tile launch_conv picks for this width
    const int bm = conv_small_m(M, Cout) ? 32 : 128;              // M tile (single utterances at the 8x8 / 4x4 levels: 32 rows)
    const int64_t tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
    const int steps = ((Cin + KC - 1) / KC) * taps;
    if (tiles >= 256 || steps < 8) return 1;          // measured: 256 beats 128 and 64 at B = 1..8
    const int wp = wino_plan(B, H, W, Cin, Cout, taps);
    if (wp >= 1) return wp;
    int64_t want = (512 + tiles - 1) / tiles;
    int64_t maxs = steps / 4;
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
bool conv_small_m(int64_t M, int Cout) { return M <= 64 && Cout > 64 && !conv_splitk_in_launch(); }   // (that form: 128-row tiles only)

bool conv_splitk_is_wino(int B, int H, int W, int Cin, int Cout, int taps) {
    return conv_wino_default_f43() && conv_supports_wino(B, H, W, Cin, 0, Cout, taps) && wino_plan(B, H, W, Cin, Cout, taps) >= 2;
}

static const bool g_no_halo = getenv("FLOWSE_NO_HALO_CONV") != nullptr;

bool conv_supports_fused_gn(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (conv_supports_head4(B, H, W, C1, C2, Cout, taps)) return true;
    if (g_no_halo || g_force_generic) return false;
    if (taps != 9 || (H & 7) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout & 3)) return false;
    const int ks = conv_ksplit(B, H, W, C1 + C2, Cout, taps);
    if (ks != 1 && ks != wino_plan(B, H, W, C1 + C2, Cout, taps)) return false;    // only the Winograd kernel runs split
    const int64_t cmax = C1 > C2 ? C1 : C2;
    return (int64_t)(9 * W + 18) * cmax * 4 < (1LL << 31) && (int64_t)Cout * 9 * (C1 + C2) * 4 < (1LL << 31);
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

// ---------------------------------------------------------------------------------------------------
// F(2,3) Winograd variant of the LDS-halo 3x3 kernel (fp32, the production kernel when Cout % 64 == 0).
//
// The 3x3 filter is factored along its VERTICAL axis: a pair of vertically adjacent output pixels (y, y+1) needs
// the four input rows d0..d3 = y-1..y+2 and, per horizontal tap kx, the four products
//     m0 = (d0 - d2) g0,  m1 = (d1 + d2) (g0+g1+g2)/2,  m2 = (d2 - d1) (g0-g1+g2)/2,  m3 = (d1 - d3) g2
//     out(y) = m0 + m1 + m2,   out(y+1) = m1 - m2 - m3                      (g_k = w[ky = k][kx])
// i.e. 4 multiplies where the direct form spends 6: the matrix cores execute 2/3 of the direct-convolution FLOPs.
// The block still owns an 8 x 16 pixel tile = 64 row pairs (4 x 16) and BN = 64 output channels; the GEMM rows are
// the PAIRS, each wave accumulates the four components of a 32-pair x 32-channel tile (4 x 16 accumulators) and
// combines them once, in registers, before the shared epilogue.
//   A side: the same LDS halo tile the direct kernel stages (GroupNorm + SiLU fused the same way), double-buffered
//           per 32-channel chunk (ONE barrier per chunk); the input transform is four VALU subtractions per float4
//           fragment, done one k-block ahead between the MFMAs.
//   B side: the pre-transformed weights are stored in MFMA FRAGMENT ORDER (launch_wino_weights:
//           [Cout/32][kx][Cin/32][component][k-block][lane][4]), so each wave fetches its B operand with one fully
//           coalesced 1 KB buffer load per (component, k-block), straight from L2 into registers, one k-block ahead:
//           no LDS staging, no LDS writes and no barrier for the weights.
template <int GN>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(ConvArgs a) {
    constexpr int BN = 64, NT = 256;
    constexpr int HROWS = 180;                           // 10 x 18 halo pixels
    constexpr int H_LOADS = 6;
    constexpr int HBUF = HROWS * LDS_ROW;                // floats per halo buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                    // [2][HROWS][LDS_ROW]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 3);
    const int b = mt / tiles_img, tt = mt - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * 8, x0 = tx * 16, n0 = nt * BN;
    const int m_tl = (b * H + y0) * W + x0;

    const int col4 = tid & 7, row0 = tid >> 3;
    unsigned hvo1[H_LOADS], hvo2[H_LOADS];
    unsigned hin = 0;
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HROWS && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        hvo1[q] = in ? (unsigned)((hy * W + hx) * C1 + col4 * 4) * 4u : OOB;
        hvo2[q] = in ? (unsigned)((hy * W + hx) * C2 + col4 * 4) * 4u : OOB;
        hin |= in ? (1u << q) : 0u;
    }
    const int64_t wbase = (int64_t)m_tl - W - 1;
    const int wpix = 9 * W + 18;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + wbase * C1), 0, wpix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + wbase * C2 : a.in1), 0, C2 ? wpix * C2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wino), 0, a.Cout * 12 * Cin * 4, 0x00020000);

    u32x4 rh[H_LOADS];
    float4 g_mu, g_sc, g_be;

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
    auto xform1 = [&](int q) {
        if (!GN) return;
        rh[q] = gn_quad<GN>(rh[q], g_mu, g_sc, g_be, (hin >> q) & 1u);
    };
    auto lstoreH = [&](int buf) {
        float* Hb = Hs + buf * HBUF;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            if (hr < HROWS) *reinterpret_cast<u32x4*>(Hb + hr * LDS_ROW + col4 * 4) = rh[q];
        }
    };

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    // this lane's pair: tile row pair 2*wm + (li >> 4), column li & 15; its four input rows start at halo row 2*pair
    const int abase = ((2 * (2 * wm + (li >> 4))) * 18 + (li & 15)) * LDS_ROW + kh * 4;
    // weight fragments: 16 KB per (32-channel slice, kx, chunk), [component][k-block][lane][4 floats]
    const int nchunks = Cin / KC;
    const unsigned wslice = (unsigned)((n0 >> 5) + wn) * 3u * (unsigned)nchunks;     // in 16 KB units
    const unsigned wvo = (unsigned)lane * 16u;

    f32x16 acc[4];                                       // the four Winograd components
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    gloadH(0);
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) xform1(q);
    lstoreH(0);
    __syncthreads();

    // Operand pipeline: the operands of k-block j+1 (four halo rows from LDS, four weight components from L2) are
    // requested before the MFMAs of block j; the halo rows are turned into the Winograd operands d0-d2, d1+d2,
    // d2-d1, d1-d3 (in place) by VALU blocks fenced between those MFMAs, so neither the load latency nor the
    // transform sits on the matrix pipe's critical path.  (Unfenced, hipcc sinks every transform to just before the
    // MFMA that consumes it and the reads right in front of that.)
#define FLOWSE_FENCE __builtin_amdgcn_sched_barrier(0);
#define FLOWSE_WLOAD(KX, J, D, BF)                                                                                   \
    {                                                                                                                \
        const float* Ha = Hcur + abase + (KX) * LDS_ROW + (J) * 8;                                                   \
        D[0] = *reinterpret_cast<const float4*>(Ha);                                                                 \
        D[1] = *reinterpret_cast<const float4*>(Ha + 18 * LDS_ROW);                                                  \
        D[2] = *reinterpret_cast<const float4*>(Ha + 36 * LDS_ROW);                                                  \
        D[3] = *reinterpret_cast<const float4*>(Ha + 54 * LDS_ROW);                                                  \
        const unsigned so = (wslice + (unsigned)(KX) * (unsigned)nchunks + (unsigned)chunk) * 16384u;                \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                              \
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, wvo + (c * 4 + (J)) * 1024, so, 0);         \
            BF[c] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z),                   \
                                __uint_as_float(t.w));                                                               \
        }                                                                                                            \
    }
#define FLOWSE_WX03(D)                                                                                               \
    {                                                                                                                \
        D[0].x -= D[2].x; D[0].y -= D[2].y; D[0].z -= D[2].z; D[0].w -= D[2].w;                                      \
        D[3].x = D[1].x - D[3].x; D[3].y = D[1].y - D[3].y; D[3].z = D[1].z - D[3].z; D[3].w = D[1].w - D[3].w;      \
    }
#define FLOWSE_WX12(D)                                                                                               \
    {                                                                                                                \
        const float4 t1 = D[1];                                                                                      \
        D[1].x += D[2].x; D[1].y += D[2].y; D[1].z += D[2].z; D[1].w += D[2].w;                                      \
        D[2].x -= t1.x; D[2].y -= t1.y; D[2].z -= t1.z; D[2].w -= t1.w;                                              \
    }
#define FLOWSE_WMMA4(V, BF, K)                                                                                       \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                    \
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[c].K, BF[c].K, acc[c], 0, 0, 0);
    // One k-block: request the NEXT block's operands (NKX, NJ -> DN, BN_), run this block's 16 MFMAs with the next
    // block's transform and one staged halo quad (XQ >= 0) fenced in between
#define FLOWSE_WPHASE(V, BF, NKX, NJ, DN, BN_, XQ)                                                                   \
    FLOWSE_WLOAD(NKX, NJ, DN, BN_) FLOWSE_FENCE                                                                      \
    FLOWSE_WMMA4(V, BF, x) FLOWSE_FENCE                                                                              \
    if (GN && (XQ) >= 0) xform1((XQ) < 0 ? 0 : (XQ));                                                                \
    FLOWSE_FENCE FLOWSE_WMMA4(V, BF, y) FLOWSE_FENCE                                                                 \
    FLOWSE_WX03(DN) FLOWSE_FENCE FLOWSE_WMMA4(V, BF, z) FLOWSE_FENCE                                                 \
    FLOWSE_WX12(DN) FLOWSE_FENCE FLOWSE_WMMA4(V, BF, w) FLOWSE_FENCE

    float4 dA[4], dB[4], bA[4], bB[4];
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const float* Hcur = Hs + (chunk & 1) * HBUF;
        gloadH(min(chunk + 1, nchunks - 1));             // next chunk's halo: normalised at kx = 1, stored at kx = 2
        FLOWSE_WLOAD(0, 0, dA, bA)
        FLOWSE_WX03(dA) FLOWSE_WX12(dA)
        FLOWSE_FENCE
        FLOWSE_WPHASE(dA, bA, 0, 1, dB, bB, -1)
        FLOWSE_WPHASE(dB, bB, 0, 2, dA, bA, -1)
        FLOWSE_WPHASE(dA, bA, 0, 3, dB, bB, -1)
        FLOWSE_WPHASE(dB, bB, 1, 0, dA, bA, -1)
        FLOWSE_WPHASE(dA, bA, 1, 1, dB, bB, 0)
        FLOWSE_WPHASE(dB, bB, 1, 2, dA, bA, 1)
        FLOWSE_WPHASE(dA, bA, 1, 3, dB, bB, 2)
        FLOWSE_WPHASE(dB, bB, 2, 0, dA, bA, 3)
        FLOWSE_WPHASE(dA, bA, 2, 1, dB, bB, 4)
        FLOWSE_WPHASE(dB, bB, 2, 2, dA, bA, 5)
        lstoreH((chunk + 1) & 1);                        // the other buffer: nobody reads it during this chunk
        FLOWSE_FENCE
        FLOWSE_WPHASE(dA, bA, 2, 3, dB, bB, -1)
        FLOWSE_WMMA4(dB, bB, x) FLOWSE_WMMA4(dB, bB, y) FLOWSE_WMMA4(dB, bB, z) FLOWSE_WMMA4(dB, bB, w)
        FLOWSE_FENCE
        __syncthreads();     // next chunk's halo is complete; everyone has left this chunk's
    }
#undef FLOWSE_WLOAD
#undef FLOWSE_WX03
#undef FLOWSE_WX12
#undef FLOWSE_WMMA4
#undef FLOWSE_WPHASE
#undef FLOWSE_FENCE

    // output transform in registers; accumulator register r of a 32x32 tile holds pair row (r&3) + 8*(r>>2) + 4*kh:
    // registers 0-7 belong to the wave's first row pair, 8-15 to the second, which is exactly the split the direct
    // kernel's <2,2,2,1> epilogue expects between its two 32-pixel tiles (two image rows x 16 columns each)
    f32x16 outp[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * i + q;
            outp[i][0][q] = acc[0][r] + acc[1][r] + acc[2][r];
            outp[i][0][q + 8] = acc[1][r] - acc[2][r] - acc[3][r];
        }
    conv_epilogue<2, 2, 2, 1>(a, outp, smem, m_tl, n0, M, HW, 0, W);
}

// [Cout][9][Cin] -> MFMA fragment order [Cout/32][kx][Cin/32][component][k-block j][lane = kh*32 + li][4]:
// element (n = 32 t + li, ci = 32 chunk + 8 j + 4 kh + e)
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, int Cout, int Cin,
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
    float* o = out + ((((n >> 5) * 3 + kx) * nchunks + chunk) * 16 + j) * 256 + lane * 4 + e;   // component 0
    o[0] = g0;
    o[1024] = 0.5f * ((g0 + g2) + g1);
    o[2048] = 0.5f * ((g0 + g2) - g1);
    o[3072] = g2;
}

int launch_wino_weights(const float* w_packed, int Cout, int Cin, float* out, hipStream_t s) {
    if ((Cout % 32) != 0 || (Cin % 32) != 0) {
        set_error("wino_weights: Cout=%d Cin=%d must be multiples of 32", Cout, Cin);
        return ERR_SHAPE;
    }
    const int64_t n = (int64_t)Cout * 3 * Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w_packed, Cout, Cin, out);
    FLOWSE_LAUNCH_CHECK();
    return OK;
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
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};

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
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        __syncthreads();                                 // everyone has left this chunk's tiles
        lstore();
        __syncthreads();
    }
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
    static const bool off = getenv("FLOWSE_NO_HEAD4") != nullptr;      // test / A-B hook
    return !off && !g_force_generic && taps == 9 && Cout == 4 && C2 == 0 && (C1 % KC) == 0 && !(H & 15) && !(W & 15) &&
           (int64_t)B * (H >> 4) * (W >> 4) >= 64 && (int64_t)(17 * W + 18) * C1 * 4 < (1LL << 31);
}

static int launch_head4(const ConvArgs& a, hipStream_t s) {
    const int grid = a.B * (a.H >> 4) * (a.W >> 4);
    const size_t lds = (size_t)(18 * 18 + 36) * LDS_ROW * sizeof(float);
    if (a.out_dt != DT_F32) {
        set_error("head4: the 4-channel output is fp32");
        return ERR_ARG;
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

#ifdef FLOWSE_TS
// Measurement build only (-DFLOWSE_TS): per-block phase timestamps of the F(4,3) and 16-bit halo kernels (s_memtime) + HW ids.
__device__ unsigned long long g_ts[8192 * 10];
extern "C" int flowse_debug_ts(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ts), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#define FLOWSE_TS_MARK(k) if (ts_on) ts[k] = __builtin_amdgcn_s_memtime();
// phase marks inside taps 2..7 of chunk 1: ts[k] accumulates the time since the previous mark
#define FLOWSE_TS_TAP(k)                                                  \
    if (ts_on && chunk == 1 && tap >= 2 && tap <= 7) {                    \
        const unsigned long long now = __builtin_amdgcn_s_memtime();      \
        if (k >= 2) ts[k] += now - ts_last;                               \
        ts_last = now;                                                    \
    }
#else
#define FLOWSE_TS_MARK(k)
#define FLOWSE_TS_TAP(k)
#endif

// ---------------------------------------------------------------------------------------------------
// F(4,3) Winograd variant (same tile, same halo staging, same fragment-order weight stream as the F(2,3) kernel).
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
#ifndef FLOWSE_F43_EXCHANGE
#define FLOWSE_F43_EXCHANGE 1            /* 0: the staged C-tile output stage of round 2 (A-B builds) */
#endif
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
#ifdef FLOWSE_TS
    unsigned long long ts[10];
    const bool ts_on = a.H == 256 && a.C1 + a.C2 == 128 && GN == 2 && !SPLIT;
    for (int k = 0; k < 10; ++k) ts[k] = 0;
#endif
    FLOWSE_TS_MARK(0)

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // a block owns `tpb` consecutive pixel tiles (in walk order) of ONE channel block
    int mg = bid / n_ntiles;
    const int nt = bid - mg * n_ntiles;
    if (a.reverse) mg = (int)gridDim.x / n_ntiles - 1 - mg;
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
    const int wpix = 9 * W + 18;
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
#ifdef FLOWSE_PROBE_BLOCKED_HALO   /* measurement probe (results garbage): address the halo AS IF activations were stored */
        /* channel-chunk-major [C/32][H][W][32] -- a halo row of a chunk is then 18 x 128 contiguous bytes instead of 18     */
        /* pieces of 128 bytes at a pitch of C x 4 bytes; same number of loads, same bytes                                    */
        const unsigned cs = 32u;
        const unsigned soff = (t.woff * 32u + (unsigned)((second ? c0 - C1 : c0) >> 5) * (unsigned)HW * 32u) * 4u;
#else
        const unsigned cs = (unsigned)(second ? C2 : C1);
        const unsigned soff = (t.woff * cs + (unsigned)(second ? c0 - C1 : c0)) * 4u;
#endif
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
#ifdef FLOWSE_PROBE_NOHALO      /* measurement probe: no halo loads inside the loop (results are garbage, timing what-if) */
#pragma unroll
        for (int q = 0; q < 3; ++q) rh[q] = u32x4{(unsigned)chunk, (unsigned)h, (unsigned)q, 0x3f800000u};
#else
#pragma unroll
        for (int q = 0; q < 3; ++q) rh[q] = hload(t, chunk, 3 * h + q);
#endif
        if (h == 0) {
            gparams(t, chunk);
            st_hin = t.hin;
        }
    };
    auto xform1 = [&](int Q) {
#ifndef FLOWSE_PROBE_NOGN
        if (GN) rh[Q % 3] = gn_quad<GN>(rh[Q % 3], g_mu, g_sc, g_be, (st_hin >> Q) & 1u);
#endif
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
    FLOWSE_TS_MARK(1)

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
#if defined(FLOWSE_PROBE_NOXFORM)      /* measurement probe: no input transform (results are garbage, timing what-if) */
#define FLOWSE_WXA(D)
#define FLOWSE_WXB(D)
#elif defined(FLOWSE_NOPK)
    // scalar fp32 forms: packed fp32 VALU (v_pk_fma_f32 / v_pk_add_f32) beside MFMAs costs more than the two scalar
    // instructions it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
#define FLOWSE_E4(Q, I) (reinterpret_cast<float*>(&(Q))[I])
#define FLOWSE_WXA(D)                                                                                                \
    {                                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
            const float r0 = FLOWSE_E4(D[0], e), r2 = FLOWSE_E4(D[2], e), r4 = FLOWSE_E4(D[4], e);                   \
            const float v = fmaf(4.f, r0, fmaf(-5.f, r2, r4));                                                       \
            if (CH == 0) FLOWSE_E4(D[0], e) = v;                                                                     \
            else FLOWSE_E4(D[4], e) = v;                                                                             \
        }                                                                                                            \
    }
#define FLOWSE_WXB(D)                                                                                                \
    {                                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
            const float r0 = FLOWSE_E4(D[0], e), r1 = FLOWSE_E4(D[1], e), r2 = FLOWSE_E4(D[2], e),                   \
                        r3 = FLOWSE_E4(D[3], e), r4 = FLOWSE_E4(D[4], e);                                            \
            if (CH == 0) {                                                                                           \
                FLOWSE_E4(D[1], e) = fmaf(-4.f, r1 + r2, r3 + r4);                                                   \
                FLOWSE_E4(D[2], e) = fmaf(4.f, r1 - r2, r4 - r3);                                                    \
            } else {                                                                                                 \
                FLOWSE_E4(D[1], e) = fmaf(2.f, r2 - r0, r3 - r1);                                                    \
                FLOWSE_E4(D[2], e) = fmaf(-2.f, r2 - r0, r3 - r1);                                                   \
            }                                                                                                        \
        }                                                                                                            \
    }
#else
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
#endif
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
    // the chunk barrier.  FLOWSE_F43_HALO_EARLY selects round 2's schedule (A-B builds).
    // ONE wait for all of this phase's weight fragments (requested a phase ago; the only younger loads are the ones just
    // issued: 3 TN fragments, + 6 halo / GroupNorm-parameter loads in a halo phase) instead of one s_waitcnt per fragment
#ifdef FLOWSE_F43_WAIT1
#define FLOWSE_WAIT_B(HL) __builtin_amdgcn_s_waitcnt(0x0F70 | (((HL) >= 0 ? 3 * TN + 6 : 3 * TN) & 15));
#else
#define FLOWSE_WAIT_B(HL)
#endif
#define FLOWSE_WPHASE(V, BF, NKX, NJ, NCHK, DN, BFN, XQ, HL)                                                         \
    FLOWSE_WLOADA(NKX, NJ, DN) FLOWSE_WLOADB(NKX, NJ, NCHK, BFN)                                                     \
    if ((HL) >= 0) gloadH(stile, cnext, (HL) < 0 ? 0 : (HL));                                                        \
    FLOWSE_FENCE                                                                                                     \
    FLOWSE_WAIT_B(HL)                                                                                                \
    FLOWSE_WMMA3(V, BF, x) FLOWSE_FENCE                                                                              \
    if (GN && (XQ) >= 0) xform1((XQ) < 0 ? 0 : (XQ));                                                                \
    FLOWSE_FENCE FLOWSE_WMMA3(V, BF, y) FLOWSE_FENCE                                                                 \
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
#ifdef FLOWSE_F43_HALO_EARLY   /* round-2 schedule (A-B builds): halo requested right before the next phase's weights, normalised from the next phase on */
        gloadH(stile, cnext, 0);
        FLOWSE_WPHASE(dA, bA, 0, 1, chunk, dB, bB, -1, -1)
        FLOWSE_WPHASE(dB, bB, 0, 2, chunk, dA, bA, 0, -1)
        FLOWSE_WPHASE(dA, bA, 0, 3, chunk, dB, bB, 1, -1)
        FLOWSE_WPHASE(dB, bB, 1, 0, chunk, dA, bA, 2, -1)
        lstoreH(nbuf, 0);
        gloadH(stile, cnext, 1);
        FLOWSE_FENCE
        FLOWSE_WPHASE(dA, bA, 1, 1, chunk, dB, bB, -1, -1)
        FLOWSE_WPHASE(dB, bB, 1, 2, chunk, dA, bA, 3, -1)
        FLOWSE_WPHASE(dA, bA, 1, 3, chunk, dB, bB, 4, -1)
        FLOWSE_WPHASE(dB, bB, 2, 0, chunk, dA, bA, 5, -1)
        lstoreH(nbuf, 1);
        FLOWSE_FENCE
        FLOWSE_WPHASE(dA, bA, 2, 1, chunk, dB, bB, -1, -1)
        FLOWSE_WPHASE(dB, bB, 2, 2, chunk, dA, bA, -1, -1)
        FLOWSE_WPHASE(dA, bA, 2, 3, chunk, dB, bB, -1, -1)
        if (!wrap) { FLOWSE_WLOADB(0, 0, cnext, bA) }
        FLOWSE_FENCE                                      // first weights of the next chunk
        FLOWSE_WMMA3(dB, bB, x) FLOWSE_WMMA3(dB, bB, y) FLOWSE_WMMA3(dB, bB, z) FLOWSE_WMMA3(dB, bB, w)
        FLOWSE_FENCE
#else
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
#endif
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
#undef FLOWSE_WAIT_B
#undef FLOWSE_FENCE
#ifdef FLOWSE_E4
#undef FLOWSE_E4
#endif
    FLOWSE_TS_MARK(6)

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
#ifdef FLOWSE_PROBE_NOEPI       /* measurement probe: no output stage (keeps the accumulators alive; results are garbage) */
    {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[c][j][r];
        if (t == 12345.678f) a.out[m_tl] = t;
        return;
    }
#endif
    if constexpr (SPLIT) {                               // split slices: raw partial tiles through the shared epilogue
        conv_epilogue_with<2, 2, 2, 1>(a, smem, m_tl, n0, M, HW, (int)blockIdx.y, W,
                                       [&](float* Cs, int CROW) { scatter_half(Cs, CROW, 0); });
    } else if constexpr (TN == 2 && FLOWSE_F43_EXCHANGE) {
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
#ifdef FLOWSE_TS
        if (ts_on) ts[2] = __builtin_amdgcn_s_memtime();
#endif
        float* red = smem + 128 * CROW;
#pragma unroll 1
        for (int half = 0; half < TN; ++half) {
            if (half) __syncthreads();                   // the statistics scratch of the first half has been read
            tile128x64_out<float>(a, smem + half * 64, CROW, red, m_tl, W, n0 + half * 64, bsmp, tile);
#ifdef FLOWSE_TS
            if (ts_on && half == 0) ts[3] = ts[4] = __builtin_amdgcn_s_memtime();
#endif
        }
    }
#ifdef FLOWSE_TS
    if (ts_on) {
        __syncthreads();
        if (CH == 0) {
            ts[7] = __builtin_amdgcn_s_memtime();
            ts[8] = __builtin_amdgcn_s_getreg(63492);          // HW_ID
            ts[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
            ts[5] = blockIdx.x;
            if (tid == 0 && bid < 8192)
                for (int k = 0; k < 10; ++k) g_ts[bid * 10 + k] = ts[k];
        }
    }
#endif
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
// (measured: 512 beats 1024 and 256 at B = 1 and B = 8); FLOWSE_F43_BN64=1 keeps the 64-channel form everywhere (A-B hook)
bool conv_f43_forced_bn64() {
    static const bool bn64 = getenv("FLOWSE_F43_BN64") != nullptr;
    return bn64;
}
static bool f43_single_tile() {      // FLOWSE_F43_TPB1=1: one tile per block everywhere (A-B hook)
    static const bool one = getenv("FLOWSE_F43_TPB1") != nullptr;
    return one;
}
bool conv_f43_wide(int B, int H, int W, int Cout) {
    return !conv_f43_forced_bn64() && (Cout % 128) == 0 && ((int64_t)B * H * W / 128) * (Cout / 128) >= 512;
}


// ---------------------------------------------------------------------------------------------------
// F(4,3), producer / consumer form (FLOWSE_F43_PC=1; 128-output-channel layers with >= 8 tiles per CU, i.e. the 256 x 256
// level).  profiles/r03_f43_probes.md: the waves that wait for weight fragments must not issue the halo loads (vector
// loads return in issue order: -6...8 % without them), and the output stage is worth 11 %.  Here ONE persistent block
// per CU has eight waves: four MFMA waves (wn, CH as in conv3x3_f43_kernel) that only read operands (halo from LDS,
// weights from L2) and issue MFMAs, and four LOADER waves that stage every halo (request, GroupNorm + SiLU, LDS write)
// one chunk ahead and run the whole output stage from an LDS dump of the accumulators while the MFMA waves are already
// in the next tile.  One register allocation per kernel (256) is why this is one block per CU and not two.
//   step g (one 32-channel chunk of one tile; all eight waves meet at ONE barrier per step):
//     MFMA waves   twelve k-block phases on halo buffer g & 1; after a tile's last chunk: accumulators -> D (24
//                  ds_write_b128 per lane), accumulators = 0
//     loader waves request halo g + 1; output stage of the tile dumped at the end of step g - 1 (first half) / g - 2
//                  (second half); GroupNorm of halo g + 1 -> buffer (g + 1) & 1
// LDS: 2 halo buffers (52 KB) + D (4 waves x 24 KB): 148 KB.  The loaders' transposition scratch is the part of D they
// have just read.
constexpr int F43PC_D_FLOATS = 4 * 3 * 2 * 1024;              // [mfma wave][component][channel tile][4 reg quads][64 lanes][4]

template <int CH>
__device__ __forceinline__ void f43pc_mfma_waves(const ConvArgs& a, float* smem, int steps, int nchunks) {
    constexpr int TN = 2;
    constexpr int HBUF = 10 * F43_HROW;
    float* Hs = smem;
    float* D = smem + 2 * HBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
    const int wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int Cin = a.C1 + a.C2;
    const int abase = (4 * (li >> 4) + CH) * F43_HROW + (li & 15) * LDS_ROW + kh * 4;
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wino), 0, a.Cout * 18 * Cin * 4, 0x00020000);
    const unsigned wslice = (unsigned)(wn * TN) * 3u * (unsigned)nchunks;     // n0 = 0: one 128-channel block per layer
    const unsigned wvo = (unsigned)lane * 16u + (unsigned)CH * 3u * 4096u;
    f32x16 acc[3][TN];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
#define PC_FENCE __builtin_amdgcn_sched_barrier(0);
#define PC_LOADA(KX, J, DD)                                                                                          \
    {                                                                                                                \
        const float* Ha = Hcur + abase + (KX) * LDS_ROW + (J) * 8;                                                   \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) DD[r] = *reinterpret_cast<const float4*>(Ha + r * F43_HROW);   \
    }
#define PC_LOADB(KX, J, CHK, BF)                                                                                     \
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
#define PC_H2(Q, H) (*reinterpret_cast<f32x2*>(&(Q).x + 2 * (H)))
#define PC_XA(DD)                                                                                                    \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f};                                                             \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                              \
            const f32x2 r0 = PC_H2(DD[0], h), r2 = PC_H2(DD[2], h), r4 = PC_H2(DD[4], h);                            \
            const f32x2 v = __builtin_elementwise_fma(c4, r0, __builtin_elementwise_fma(cm5, r2, r4));               \
            if (CH == 0) PC_H2(DD[0], h) = v;                                                                        \
            else PC_H2(DD[4], h) = v;                                                                                \
        }                                                                                                            \
    }
#define PC_XB(DD)                                                                                                    \
    {                                                                                                                \
        const f32x2 c4 = {4.f, 4.f}, cm4 = {-4.f, -4.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};                        \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                              \
            const f32x2 r0 = PC_H2(DD[0], h), r1 = PC_H2(DD[1], h), r2 = PC_H2(DD[2], h), r3 = PC_H2(DD[3], h),      \
                        r4 = PC_H2(DD[4], h);                                                                        \
            if (CH == 0) {                                                                                           \
                PC_H2(DD[1], h) = __builtin_elementwise_fma(cm4, r1 + r2, r3 + r4);                                  \
                PC_H2(DD[2], h) = __builtin_elementwise_fma(c4, r1 - r2, r4 - r3);                                   \
            } else {                                                                                                 \
                PC_H2(DD[1], h) = __builtin_elementwise_fma(c2, r2 - r0, r3 - r1);                                   \
                PC_H2(DD[2], h) = __builtin_elementwise_fma(cm2, r2 - r0, r3 - r1);                                  \
            }                                                                                                        \
        }                                                                                                            \
    }
#define PC_MMA(V, BF, K)                                                                                             \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) _Pragma("unroll") for (int j = 0; j < TN; ++j)                     \
        acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[(c == 0 && CH == 1) ? 4 : c].K, BF[c][j].K, acc[c][j], 0, 0, 0);
#define PC_PHASE(V, BF, NKX, NJ, NCHK, DN, BFN)                                                                      \
    PC_LOADA(NKX, NJ, DN) PC_LOADB(NKX, NJ, NCHK, BFN) PC_FENCE                                                      \
    PC_MMA(V, BF, x) PC_FENCE PC_MMA(V, BF, y) PC_FENCE                                                              \
    PC_XA(DN) PC_FENCE PC_MMA(V, BF, z) PC_FENCE                                                                     \
    PC_XB(DN) PC_FENCE PC_MMA(V, BF, w) PC_FENCE
    float4 dA[5], dB[5], bA[3][TN], bB[3][TN];
    PC_LOADB(0, 0, 0, bA)
    __syncthreads();                                             // halo of step 0 is staged
    int chunk = 0;
    for (int g = 0; g < steps; ++g) {
        const float* Hcur = Hs + (g & 1) * HBUF;
        const int cnext = chunk + 1 < nchunks ? chunk + 1 : 0;
        PC_LOADA(0, 0, dA)
        PC_XA(dA) PC_XB(dA)
        PC_FENCE
        PC_PHASE(dA, bA, 0, 1, chunk, dB, bB)
        PC_PHASE(dB, bB, 0, 2, chunk, dA, bA)
        PC_PHASE(dA, bA, 0, 3, chunk, dB, bB)
        PC_PHASE(dB, bB, 1, 0, chunk, dA, bA)
        PC_PHASE(dA, bA, 1, 1, chunk, dB, bB)
        PC_PHASE(dB, bB, 1, 2, chunk, dA, bA)
        PC_PHASE(dA, bA, 1, 3, chunk, dB, bB)
        PC_PHASE(dB, bB, 2, 0, chunk, dA, bA)
        PC_PHASE(dA, bA, 2, 1, chunk, dB, bB)
        PC_PHASE(dB, bB, 2, 2, chunk, dA, bA)
        PC_PHASE(dA, bA, 2, 3, chunk, dB, bB)
        PC_LOADB(0, 0, cnext, bA) PC_FENCE
        PC_MMA(dB, bB, x) PC_MMA(dB, bB, y) PC_MMA(dB, bB, z) PC_MMA(dB, bB, w)
        PC_FENCE
        if (cnext == 0) {                                        // tile finished: hand the accumulators to the loader waves
            float* Dw = D + wave * (3 * 2 * 1024);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<float4*>(Dw + ((c * 2 + j) * 4 + q) * 256 + lane * 4) =
                            make_float4(acc[c][j][4 * q], acc[c][j][4 * q + 1], acc[c][j][4 * q + 2], acc[c][j][4 * q + 3]);
                        acc[c][j][4 * q] = acc[c][j][4 * q + 1] = acc[c][j][4 * q + 2] = acc[c][j][4 * q + 3] = 0.f;
                    }
        }
        chunk = cnext;
        __syncthreads();
    }
    __syncthreads();                                             // the loaders' two drain steps
    __syncthreads();
#undef PC_FENCE
#undef PC_LOADA
#undef PC_LOADB
#undef PC_H2
#undef PC_XA
#undef PC_XB
#undef PC_MMA
#undef PC_PHASE
}

template <int GN>
__device__ __forceinline__ void f43pc_loader_waves(const ConvArgs& a, float* smem, int tile0, int ntile, int nchunks) {
    constexpr int HBUF = 10 * F43_HROW, HROWS = 180, H_LOADS = 6;
    float* Hs = smem;
    float* D = smem + 2 * HBUF;
    const int tl = threadIdx.x - 256;                            // 0..255 over the four loader waves
    const int lane = tl & 63;
    const int lw = __builtin_amdgcn_readfirstlane(tl >> 6);      // 0..3: output stage of channel group wn = lw & 1, tile j = lw >> 1
    const int H = a.H, W = a.W, HW = H * W;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2, Cout = a.Cout;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 3);
    const int col4 = tl & 7, row0 = tl >> 3;
    unsigned hpix[H_LOADS];
    int hlds[H_LOADS];
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        hpix[q] = (unsigned)(hy * W + hx);
        hlds[q] = hr < HROWS ? hy * F43_HROW + hx * LDS_ROW + col4 * 4 : -1;
    }
    const int b = tile0 / tiles_img;                             // all tiles of a block lie in one sample (launch_f43pc)
    const int64_t sbase = (int64_t)b * HW - W - 1;
    const int spix = HW + 2 * W + 2;
    const __amdgpu_buffer_rsrc_t rsrc1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in1 + sbase * C1), 0, spix * C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(C2 ? a.in2 + sbase * C2 : a.in1), 0, C2 ? spix * C2 * 4 : 0, 0x00020000);
    auto tile_xy = [&](int mt, int& ty, int& tx) {
        const int tt = mt - b * tiles_img;
        if ((tiles_x & 3) == 0) {
            const int per_strip = 4 * (H >> 3);
            const int strip = tt / per_strip, w = tt - strip * per_strip;
            ty = w >> 2;
            tx = strip * 4 + (w & 3);
        } else {
            ty = tt / tiles_x;
            tx = tt - ty * tiles_x;
        }
    };
    u32x4 rh[H_LOADS];
    unsigned hin = 0;
    // request the halo of (tile index ti, chunk): registers only
    auto request = [&](int ti, int chunk) {
        int ty, tx;
        tile_xy(tile0 + ti, ty, tx);
        const int y0 = ty * 8, x0 = tx * 16;
        hin = 0;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const int hr = row0 + 32 * q;
            const int hy = hr / 18, hx = hr - hy * 18;
            const bool in = hr < HROWS && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
            hin |= in ? (1u << q) : 0u;
        }
        const unsigned woff = (unsigned)(y0 * W + x0);
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;
        const unsigned cs = (unsigned)(second ? C2 : C1);
        const unsigned soff = (woff * cs + (unsigned)(second ? c0 - C1 : c0)) * 4u;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const unsigned off = ((hin >> q) & 1u) ? (hpix[q] * cs + (unsigned)col4 * 4u) * 4u : OOB;
            rh[q] = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, soff, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off, soff, 0);
        }
    };
    // GroupNorm + SiLU of the requested halo, into halo buffer `buf`
    auto deliver = [&](int chunk, int buf) {
        float4 g_mu = make_float4(0.f, 0.f, 0.f, 0.f), g_sc = g_mu, g_be = g_mu;
        if (GN) {
            const int cg = chunk * KC + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
        float* Hb = Hs + buf * HBUF;
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            u32x4 v = rh[q];
            if (GN) v = gn_quad<GN>(v, g_mu, g_sc, g_be, (hin >> q) & 1u);
            if (hlds[q] >= 0) *reinterpret_cast<u32x4*>(Hb + hlds[q]) = v;
        }
    };
    // ---- output stage state (lane mapping of the MFMA C/D layout, as dumped)
    const int li = lane & 31, kh = lane >> 5;
    const int wn = lw & 1, jt = lw >> 1;
    const int pl = lane >> 3, cq = lane & 7;
    const int ch0 = wn * 64 + jt * 32;
    float o[4][16];
    float4 piv = make_float4(0.f, 0.f, 0.f, 0.f), s1 = piv, s2 = piv;
    int roff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pp = i * 8 + pl;
        roff[i] = ((pp >> 4) * W + (pp & 15)) * Cout;
    }
    const bool has_res = a.res != nullptr;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + ch0 + cq * 4);
    if (a.bias2) {
        const float4 t = *reinterpret_cast<const float4*>(a.bias2 + (int64_t)b * a.bias2_stride + ch0 + cq * 4);
        bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w;
    }
    const float scale = a.scale;
    // part 0: read the six components of this wave's 32 channels, form the four output rows, first row quad of the tile;
    // part 1: second row quad, statistics
    auto out_part = [&](int ti, int part) {
        int ty, tx;
        tile_xy(tile0 + ti, ty, tx);
        const int64_t pix0 = ((int64_t)b * H + ty * 8 + 4 * part) * W + tx * 16;
        const float* resb = a.res + pix0 * Cout + ch0 + cq * 4;
        float* outb = a.out + pix0 * Cout + ch0 + cq * 4;
        float4 rres[8];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rres[i] = *reinterpret_cast<const float4*>(resb + roff[i]);
        }
        const float* D0 = D + (0 * 2 + wn) * (3 * 2 * 1024);     // MFMA wave (wn, CH 0): m0, m1, m2
        const float* D1 = D + (1 * 2 + wn) * (3 * 2 * 1024);     // MFMA wave (wn, CH 1): m5, m3, m4
        if (part == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 m[6];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    m[c] = *reinterpret_cast<const float4*>(D0 + ((c * 2 + jt) * 4 + q) * 256 + lane * 4);
                    m[3 + c] = *reinterpret_cast<const float4*>(D1 + ((c * 2 + jt) * 4 + q) * 256 + lane * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m0 = (&m[0].x)[e], m1 = (&m[1].x)[e], m2 = (&m[2].x)[e];
                    const float m5 = (&m[3].x)[e], m3 = (&m[4].x)[e], m4 = (&m[5].x)[e];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    const int r = 4 * q + e;
                    o[0][r] = (m0 + s12) + s34;
                    o[1][r] = fmaf(2.f, d34, d12);
                    o[2][r] = fmaf(4.f, s34, s12);
                    o[3][r] = fmaf(8.f, d34, d12) + m5;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // transposition scratch [64 px][32 ch] = two 4 KB blocks of D that only this wave reads and has read: components
        // 0 and 1 of (wn, CH 0, tile jt) hold pixels 0-31 and 32-63
        float* Tlo = const_cast<float*>(D0) + (0 * 2 + jt) * 1024;
        float* Thi = const_cast<float*>(D0) + (1 * 2 + jt) * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * part + e;
                (k < 2 ? Tlo : Thi)[((k & 1) * 16 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + li] = o[k][r];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 8 + pl;
            float4 v = *reinterpret_cast<const float4*>((i < 4 ? Tlo : Thi) + (pp & 31) * 32 + cq * 4);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
            if (has_res) { v.x += rres[i].x; v.y += rres[i].y; v.z += rres[i].z; v.w += rres[i].w; }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            *reinterpret_cast<float4*>(outb + roff[i]) = v;
            if (part == 0 && i == 0) { piv = v; s1 = s2 = make_float4(0.f, 0.f, 0.f, 0.f); }
            const float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
        }
        if (part == 0 || !a.stats) return;
        float mean[4] = {piv.x + s1.x * (1.f / 16), piv.y + s1.y * (1.f / 16), piv.z + s1.z * (1.f / 16), piv.w + s1.w * (1.f / 16)};
        float m2[4] = {fmaxf(s2.x - s1.x * s1.x * (1.f / 16), 0.f), fmaxf(s2.y - s1.y * s1.y * (1.f / 16), 0.f),
                       fmaxf(s2.z - s1.z * s1.z * (1.f / 16), 0.f), fmaxf(s2.w - s1.w * s1.w * (1.f / 16), 0.f)};
        float cnt = 16.f;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mo = __shfl_xor(mean[e], off), qo = __shfl_xor(m2[e], off);
                const float d = mo - mean[e];
                m2[e] = m2[e] + qo + d * d * (0.5f * cnt);
                mean[e] = 0.5f * (mean[e] + mo);
            }
            cnt *= 2.f;
        }
        if (pl == 0) {
            const int tile_ix = ty * tiles_x + tx;
            float* dst = a.stats + (((int64_t)b * a.stats_nblk + tile_ix) * Cout + ch0 + cq * 4) * 2;
            *reinterpret_cast<float4*>(dst) = make_float4(mean[0], m2[0], mean[1], m2[1]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
        }
    };
    const int steps = ntile * nchunks;
    request(0, 0);
    deliver(0, 0);
    __syncthreads();
    int chunk = 0, ti = 0;
    for (int g = 0; g < steps; ++g) {
        // what the MFMA waves compute now: (ti, chunk); stage (ti', chunk') = the next step's
        int cn = chunk + 1, tn = ti;
        if (cn == nchunks) { cn = 0; ++tn; }
        const bool more = g + 1 < steps;
        if (more) request(tn, cn);
        if (ti > 0 && chunk == 0) out_part(ti - 1, 0);
        if (ti > 0 && chunk == 1) out_part(ti - 1, 1);
        if (more) deliver(cn, (g + 1) & 1);
        chunk = cn;
        ti = tn;
        __syncthreads();
    }
    out_part(ntile - 1, 0);
    __syncthreads();
    out_part(ntile - 1, 1);
    __syncthreads();
}

template <int GN>
__global__ __launch_bounds__(512, 2) void conv3x3_f43pc_kernel(ConvArgs a, int tpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nchunks = (a.C1 + a.C2) / KC;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4) f43pc_loader_waves<GN>(a, smem, bid * tpb, tpb, nchunks);
    else if (wave >= 2) f43pc_mfma_waves<1>(a, smem, tpb * nchunks, nchunks);
    else f43pc_mfma_waves<0>(a, smem, tpb * nchunks, nchunks);
}

// FLOWSE_F43_PC=1 and: one 128-channel block per layer, tiles divisible over 256 persistent blocks with >= 8 tiles each,
// all of a block's tiles in one sample, >= 4 chunks
static bool f43pc_ok(const ConvArgs& a, int* tpb) {
    static const bool on = getenv("FLOWSE_F43_PC") != nullptr;
    if (!on || a.partial || a.Cout != 128) return false;
    const int64_t tiles = (int64_t)a.B * a.H * a.W / 128, tiles_img = (int64_t)a.H * a.W / 128;
    const int nch = (a.C1 + a.C2) / KC;
    if (tiles % 256 != 0 || nch < 4) return false;
    const int64_t t = tiles / 256;
    if (t < 8 || tiles_img % t != 0) return false;
    *tpb = (int)t;
    return true;
}

static int launch_f43pc(const ConvArgs& a, int tpb, hipStream_t s) {
    const size_t lds = (2 * 10 * F43_HROW + F43PC_D_FLOATS) * sizeof(float);
    const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
    const int grid = (int)((int64_t)a.B * a.H * a.W / 128 / tpb);
#define FLOWSE_LPC(G)                                                                               \
    {                                                                                               \
        if (const int rc = allow_lds<&conv3x3_f43pc_kernel<G>>(lds)) return rc;                     \
        hipLaunchKernelGGL((conv3x3_f43pc_kernel<G>), dim3(grid), dim3(512), lds, s, a, tpb);       \
    }
    if (gn == 2) FLOWSE_LPC(2) else if (gn == 1) FLOWSE_LPC(1) else FLOWSE_LPC(0)
#undef FLOWSE_LPC
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

static int launch_f43(const ConvArgs& a, hipStream_t s) {
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int ks = a.ksplit > 1 ? a.ksplit : 1;       // slices of 32-channel chunks (gridDim.y), see wino_plan
    const bool wide = !a.partial && conv_f43_wide(a.B, a.H, a.W, a.Cout);
    {
        int pc_tpb = 0;
        if (wide && f43pc_ok(a, &pc_tpb)) return launch_f43pc(a, pc_tpb, s);
    }
    // Tiles per block of the 128-channel form: as many (4, 2) as still leave two full rounds of 512 blocks (256 CUs x 2),
    // so that the prologue -- first halo from HBM, its GroupNorm, first weights, ~14 % of a one-tile block's life -- is
    // paid once per block instead of once per tile.  Needs an even chunk count (every tile starts in halo buffer 0).
    int tpb = 1;
    if (wide && FLOWSE_F43_EXCHANGE && (((a.C1 + a.C2) / KC) & 1) == 0 && !f43_single_tile()) {
        const int64_t blocks1 = (M / 128) * (a.Cout / 128);
        for (int t = 4; t >= 2; t >>= 1)
            if (((int64_t)a.H * a.W / 128) % t == 0 && blocks1 / t >= 1024) { tpb = t; break; }
    }
    const int grid = (int)(M / 128 / tpb) * (a.Cout / (wide ? 128 : 64));
    const size_t lds_halo = 2 * 10 * F43_HROW * sizeof(float);         // two halo buffers; > the 64-channel C tile
#if FLOWSE_F43_EXCHANGE
    const size_t lds_c = (10 * F43_HROW + 4 * 12 * 256) * sizeof(float);   // halo buffer 0 + the exchange region of the output stage
#else
    const size_t lds_c = ((size_t)128 * (64 * 2 + 4) + 4 * 64 * 2) * sizeof(float);   // the 128-channel C tile + statistics scratch
#endif
    const size_t lds = wide && lds_c > lds_halo ? lds_c : lds_halo;
    const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
    static const int snake = getenv("FLOWSE_F43_SNAKE") ? atoi(getenv("FLOWSE_F43_SNAKE")) : 0;   // A-B hook, see ConvArgs::reverse
    static int flip = 0;                                 // (launch order is the plan's order: deterministic per process)
    ConvArgs ar = a;
    if (snake == 1 && !a.partial) ar.reverse = (flip++) & 1;
    else if (snake == 2 && !a.partial) ar.reverse = 1;
#define FLOWSE_LF43(G, SP, TNV)                                                                              \
    {                                                                                                        \
        if (const int rc = allow_lds<&conv3x3_f43_kernel<G, SP, TNV>>(lds)) return rc;                       \
        hipLaunchKernelGGL((conv3x3_f43_kernel<G, SP, TNV>), dim3(grid, ks), dim3(256), lds, s, ar, tpb);         \
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

static const bool g_no_wino = getenv("FLOWSE_NO_WINOGRAD") != nullptr;
bool conv_wino_default_f43() {
    static const bool f23 = getenv("FLOWSE_WINOGRAD") && std::string(getenv("FLOWSE_WINOGRAD")) == "f23";
    return !f23;
}

bool conv_supports_wino(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    return !g_no_wino && (Cout % 64) == 0 && conv_supports_fused_gn(B, H, W, C1, C2, Cout, taps) &&
           wino_plan(B, H, W, C1 + C2, Cout, taps) >= 1 && (int64_t)Cout * 18 * (C1 + C2) * 4 < (1LL << 31) &&
           // the F(4,3) kernel addresses a whole sample through one buffer descriptor per source tensor
           ((int64_t)H * W + 2 * W + 2) * (C1 > C2 ? C1 : C2) * 4 < (1LL << 31);
}

static int launch_wino(const ConvArgs& a, hipStream_t s) {
    const int64_t M = (int64_t)a.B * a.H * a.W;
    const int grid = (int)(M / 128) * (a.Cout / 64);
    const size_t lds = 2 * 180 * LDS_ROW * sizeof(float);              // two halo buffers; > the <2,2,2,1> epilogue's 43 KB
    if (const int rc = allow_lds<&conv3x3_wino_kernel<0>>(lds)) return rc;
    if (const int rc = allow_lds<&conv3x3_wino_kernel<1>>(lds)) return rc;
    if (const int rc = allow_lds<&conv3x3_wino_kernel<2>>(lds)) return rc;
    if (a.gn.mean && a.gn_silu)
        hipLaunchKernelGGL(conv3x3_wino_kernel<2>, dim3(grid), dim3(256), lds, s, a);
    else if (a.gn.mean)
        hipLaunchKernelGGL(conv3x3_wino_kernel<1>, dim3(grid), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(conv3x3_wino_kernel<0>, dim3(grid), dim3(256), lds, s, a);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

// raw (un-widened) channel quad: 4 dwords for float, 2 for the 16-bit types
template <class ST> struct RawQuad { typedef u32x4 type; };
template <> struct RawQuad<bf16_t> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <> struct RawQuad<f16_t> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <class ST>
__device__ __forceinline__ typename RawQuad<ST>::type buf_ld_raw(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (std::is_same<ST, float>::value) return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
template <class ST>
__device__ __forceinline__ u32x4 widen_quad(typename RawQuad<ST>::type t) {
    if constexpr (std::is_same<ST, float>::value) {
        return t;
    } else if constexpr (std::is_same<ST, bf16_t>::value) {
        u32x4 o;
        o.x = t.x << 16; o.y = t.x & 0xffff0000u; o.z = t.y << 16; o.w = t.y & 0xffff0000u;
        return o;
    } else {
        const _Float16 e0 = __builtin_bit_cast(_Float16, (unsigned short)(t.x & 0xffffu));
        const _Float16 e1 = __builtin_bit_cast(_Float16, (unsigned short)(t.x >> 16));
        const _Float16 e2 = __builtin_bit_cast(_Float16, (unsigned short)(t.y & 0xffffu));
        const _Float16 e3 = __builtin_bit_cast(_Float16, (unsigned short)(t.y >> 16));
        u32x4 o;
        o.x = __float_as_uint((float)e0); o.y = __float_as_uint((float)e1);
        o.z = __float_as_uint((float)e2); o.w = __float_as_uint((float)e3);
        return o;
    }
}


// ---------------------------------------------------------------------------------------------------
// Single-plane (bf16 / half) LDS-halo 3x3 kernel, built for THREE blocks per CU.
//
// Same tile (8 x 16 pixels x 128 output channels, 4 waves x 2 x 2 tiles of v_mfma_f32_32x32x16) and the same halo staging
// with fused GroupNorm + SiLU as conv3x3_halo_bf16_kernel.  A 16-bit MFMA phase is 16x shorter than an fp32 one, so with
// K = 9 x 128 .. 9 x 512 the kernel is a chain of short latencies -- LDS fragment reads, one barrier per tap, the
// epilogue (which for K = 1152 costs about as many issue cycles as the whole main loop) -- and what hides them is
// occupancy: measured (rocprofv3 SQ counters) the two-blocks-per-CU form kept the matrix pipe 28 % busy with the waves
// parked on s_waitcnt / s_barrier 36 % of the time, no matter whether one or three taps were staged per barrier.  So:
//   * LDS <= 43 KB: one tap's weight tile (10 KB) double-buffered + the halo (15 KB); the epilogue handles the 128
//     output channels as two halves of 64 through a 35 KB C tile;
//   * <= 168 VGPRs: one weight register set, requested one tap ahead (pinned in front of the MFMAs: hipcc otherwise
//     sinks the loads behind them and exposes the full L2 latency every tap).
// IT / OT: storage types of the inputs and of res / out (float, or the 16-bit type of the operands).
// MT = 8 x 16 pixel sub-tiles per block, stacked vertically: 1 (three blocks per CU) or 2 (a 16 x 16 pixel tile, two
// blocks per CU, 16-bit inputs only).  With MT = 2 every staged weight tile feeds twice the MFMAs: half the per-block
// weight stream from L2, half the barriers and B-fragment reads per MFMA, a smaller halo overhead (324 / 256 vs 180 / 128
// pixels read per pixel computed).
template <bool GN, bool F16, class IT, class OT, int MT = 1>
__global__ __launch_bounds__(256, MT == 1 ? 3 : 2) void conv3x3_halo16_kernel(ConvArgs a) {
    constexpr int BN = 128, ROWB = 80, TROWS = 8 * MT, HROWS = (TROWS + 2) * 18, H_LOADS = (HROWS * 8 + 255) / 256;
    constexpr int HPITCH = 18 * ROWB + 96;                 // halo image row: 1536 B = 0 mod 256, so the two image rows a
                                                           // wave's 32 lanes touch use the same bank pattern (conflict-free)
    constexpr int BTILE = BN * ROWB;                       // one tap's weight tile
    constexpr unsigned ES = sizeof(IT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FLOWSE_TS
    unsigned long long ts[10], ts_last = 0;
    const bool ts_on = a.H == 256 && a.C1 + a.C2 == 128 && GN;
    for (int k = 0; k < 10; ++k) ts[k] = 0;
#endif
    FLOWSE_TS_MARK(0)
    char* Hs = reinterpret_cast<char*>(smem);              // [TROWS + 2][HPITCH]
    char* Bs = Hs + (TROWS + 2) * HPITCH;                           // [2 buffers][BN][ROWB]

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    const int M = a.B * HW;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int nchunks = Cin / KC;
    const int n_ntiles = a.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_ntiles, nt = bid - mt * n_ntiles;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H / TROWS);
    const int b = mt / tiles_img, tt = mt - b * tiles_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = ty * TROWS, x0 = tx * 16, n0 = nt * BN;
    const int m_tl = (b * H + y0) * W + x0;

    const int col4 = tid & 7, row0 = tid >> 3;             // halo staging: 8 channel quads x 32 rows per pass
    // Window row / column of this thread's halo quads, two quads per register (hy | hx << 8 in 16 bits each): the pixel
    // offset (hy W + hx) and the LDS offset are re-derived where used, once per chunk -- registers matter more here.
    unsigned hyx[(H_LOADS + 1) / 2];
    unsigned hin = 0;                                      // bit q: quad q lies inside the image
    unsigned hval = 0;                                     // bit q: quad q is a halo pixel at all (hr < HROWS)
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) {
        const int hr = row0 + 32 * q;
        const int hy = hr / 18, hx = hr - hy * 18;
        const bool in = hr < HROWS && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
        const unsigned pk = (unsigned)hy | ((unsigned)hx << 8);
        if (q & 1) hyx[q >> 1] |= pk << 16; else hyx[q >> 1] = pk;
        hin |= in ? (1u << q) : 0u;
        hval |= hr < HROWS ? (1u << q) : 0u;
    }
    // `fence_hyx()` makes the packed coordinates opaque at the point of use: without it hipcc hoists every derived
    // per-quad offset out of the chunk loop and, out of registers, parks them in scratch (reloaded under vmcnt(0)).
    auto fence_hyx = [&]() {
#pragma unroll
        for (int k = 0; k < (H_LOADS + 1) / 2; ++k) asm volatile("" : "+v"(hyx[k]));
    };
    auto h_y = [&](int q) { return (hyx[q >> 1] >> ((q & 1) * 16)) & 0xffu; };
    auto h_x = [&](int q) { return (hyx[q >> 1] >> ((q & 1) * 16 + 8)) & 0xffu; };
    const int bcol = tid & 3, brow0 = tid >> 2;            // weight staging: 4 x 16-byte columns, rows brow0 + 64 q
    const unsigned bvo0 = (unsigned)((n0 + brow0) * 9 * nchunks * 64 + bcol * 16), bvo_step = (unsigned)(64 * 9 * nchunks * 64);
    const int64_t wbase = (int64_t)m_tl - W - 1;
    const int wpix = (TROWS + 1) * W + 18;
    const IT* in1p = reinterpret_cast<const IT*>(a.in1);
    const IT* in2p = reinterpret_cast<const IT*>(a.in2);
    const IT* win1 = in1p + wbase * C1;                   // window origins of the two inputs (block-uniform)
    const IT* win2 = C2 ? in2p + wbase * C2 : in1p;
    const __amdgpu_buffer_rsrc_t rsrcw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wq), 0, a.Cout * 9 * nchunks * 64, 0x00020000);

    typename RawQuad<IT>::type rh[H_LOADS];                // raw until transformed; afterwards .xy = the quad as 4 x 16 bit
    u32x4 rb[2];
    float4 g_mu, g_sc, g_be;
    auto gloadH = [&](int chunk) {
        const int c0 = chunk * KC;
        const bool second = c0 >= C1;                      // block-uniform: the descriptor is built from scalars, per chunk
        const unsigned soff = (unsigned)(second ? c0 - C1 : c0) * ES;
        const unsigned cs = (unsigned)(second ? C2 : C1);
        // the window origin goes through readfirstlane: under SGPR pressure hipcc otherwise keeps it in VGPRs and wraps
        // every load in a waterfall loop
        const uint64_t wsel = reinterpret_cast<uint64_t>(second ? win2 : win1);
        const uint64_t wuni = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wsel) |
                              ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wsel >> 32)) << 32);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<IT*>(wuni), 0, __builtin_amdgcn_readfirstlane(wpix * (int)cs * (int)ES), 0x00020000);
        unsigned hin_l = hin;
        asm volatile("" : "+v"(hin_l));                    // opaque, like the coordinates: no hoisted per-quad masks
        fence_hyx();
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q) {
            const unsigned pix = h_y(q) * (unsigned)W + h_x(q);
            const unsigned off = ((hin_l >> q) & 1u) ? (pix * cs + (unsigned)col4 * 4u) * ES : OOB;
            rh[q] = buf_ld_raw<IT>(rsrc, off, soff);
        }
        if (GN) {
            const int cg = c0 + col4 * 4;
            g_mu = *reinterpret_cast<const float4*>(a.gn.mean + (int64_t)b * Cin + cg);
            g_sc = *reinterpret_cast<const float4*>(a.gn.scale + (int64_t)b * Cin + cg);
            g_be = *reinterpret_cast<const float4*>(a.gn.beta + cg);
        }
    };
    // GroupNorm + SiLU of quad Q in fp32 (packed forms, v_exp / v_rcp), then one rounding to the operand type:
    // rh[Q].xy = the quad as 4 x 16 bit.  Out-of-image pixels are zero AFTER the activation.
    auto xform1 = [&](int Q) {
        u32x4 t = widen_quad<IT>(rh[Q]);
        if (GN) t = a.gn_silu ? gn_quad<2>(t, g_mu, g_sc, g_be, (hin >> Q) & 1u) : gn_quad<1>(t, g_mu, g_sc, g_be, (hin >> Q) & 1u);
        const float v[4] = {__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
        unsigned short h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (F16) {
                const _Float16 c = (_Float16)v[e];
                h[e] = __builtin_bit_cast(unsigned short, c);
            } else {
                const __bf16 c = (__bf16)v[e];
                h[e] = __builtin_bit_cast(unsigned short, c);
            }
        }
        rh[Q].x = (unsigned)h[0] | ((unsigned)h[1] << 16);
        rh[Q].y = (unsigned)h[2] | ((unsigned)h[3] << 16);
    };
    auto lstoreH = [&]() {
        fence_hyx();
#pragma unroll
        for (int q = 0; q < H_LOADS; ++q)
            if ((hval >> q) & 1u)
                *reinterpret_cast<uint2*>(Hs + h_y(q) * HPITCH + h_x(q) * ROWB + col4 * 8) = make_uint2(rh[q].x, rh[q].y);
    };
    const int S_all = nchunks * 9;
    auto gloadB = [&](int s) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const unsigned soff_b = (unsigned)((tap * nchunks + chunk) * 64);
#pragma unroll
        for (int q = 0; q < 2; ++q) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcw, bvo0 + q * bvo_step, soff_b, 0);
    };
    auto lstoreB = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            *reinterpret_cast<u32x4*>(Bs + buf * BTILE + (brow0 + 64 * q) * ROWB + bcol * 16) = rb[q];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    int abase[2];                                          // sub-tile 0; sub-tile t adds 8 t image rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = 2 * (wm * 2 + i) + (li >> 4), px = li & 15;
        abase[i] = (py + 1) * HPITCH + (px + 1) * ROWB + kh * 16;
    }
    f32x16 acc[MT][2][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    gloadH(0);
    gloadB(0);
#pragma unroll
    for (int q = 0; q < H_LOADS; ++q) xform1(q);
    lstoreH();
    lstoreB(0);
    __syncthreads();
    FLOWSE_TS_MARK(1)

    // One tap.  TAP is a literal: the nine taps of a chunk are straight-line code, no load sits under a branch.  The
    // next tap's weights are requested first (pinned in front of the MFMAs), the next chunk's halo at tap 1; its six
    // quads are normalised one per tap behind the MFMAs of taps 2..7 and written after tap 8's barrier.
#define FLOWSE_TAP16(TAP)                                                                                            \
    {                                                                                                                \
        constexpr int tap = TAP;                                                                                     \
        const int s = chunk * 9 + tap;                                                                               \
        const int buf = s & 1;                                                                                       \
        FLOWSE_TS_TAP(0)                                                                                             \
        gloadB(min(s + 1, S_all - 1));                                                                               \
        if (tap == 1) gloadH(min(chunk + 1, nchunks - 1));                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        constexpr int tapoff = (tap / 3 - 1) * HPITCH + (tap % 3 - 1) * ROWB;                                        \
        const char* Bb = Bs + buf * BTILE + (wn * 64 + li) * ROWB + kh * 16;                                         \
        _Pragma("unroll") for (int mh = 0; mh < 2; ++mh) {                                                           \
            bf16x8 af[MT][2], bf[2];                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
                _Pragma("unroll") for (int t = 0; t < MT; ++t)                                                       \
                    af[t][i] = *reinterpret_cast<const bf16x8*>(Hs + abase[i] + t * 8 * HPITCH + tapoff + mh * 32);  \
                bf[i] = *reinterpret_cast<const bf16x8*>(Bb + i * 32 * ROWB + mh * 32);                              \
            }                                                                                                        \
            _Pragma("unroll") for (int t = 0; t < MT; ++t) _Pragma("unroll") for (int i = 0; i < 2; ++i)             \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
                if (F16)                                                                                             \
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[t][i]),       \
                                                                         __builtin_bit_cast(f16x8, bf[j]), acc[t][i][j], 0, 0, 0); \
                else                                                                                                 \
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][i], bf[j], acc[t][i][j], 0, 0, 0);  \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        FLOWSE_TS_TAP(2)                                                                                             \
        /* GroupNorm of the next chunk's quads, in the shadow of the MFMAs just issued: taps 2..7 take them all */   \
        if (tap >= 2 && tap <= 7) {                                                                                  \
            constexpr int QPT = (H_LOADS + 5) / 6;                                                                   \
            _Pragma("unroll") for (int qq = 0; qq < QPT; ++qq)                                                       \
                if ((tap - 2) * QPT + qq < H_LOADS) xform1((tap - 2) * QPT + qq);                                    \
        }                                                                                                            \
        FLOWSE_TS_TAP(3)                                                                                             \
        lstoreB(buf ^ 1);                                /* at the very last tap: a spare tile into the idle buffer */ \
        FLOWSE_TS_TAP(4)                                                                                             \
        __syncthreads();                                                                                             \
        FLOWSE_TS_TAP(5)                                                                                             \
        if (tap == 8) {                                  /* everyone is done with this chunk's halo */              \
            lstoreH();                                                                                               \
            __syncthreads();                                                                                         \
        }                                                                                                            \
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        FLOWSE_TAP16(0) FLOWSE_TAP16(1) FLOWSE_TAP16(2) FLOWSE_TAP16(3) FLOWSE_TAP16(4)
        FLOWSE_TAP16(5) FLOWSE_TAP16(6) FLOWSE_TAP16(7) FLOWSE_TAP16(8)
    }
#undef FLOWSE_TAP16
    FLOWSE_TS_MARK(6)

    if constexpr (sizeof(OT) == 2) {
        // ---- 16-bit storage: output straight from the accumulators (halo16_out_direct)
        const int bsmp = m_tl / HW;
        const int rem = m_tl - bsmp * HW;
        const int tile0 = ((rem / W) >> 3) * (W >> 4) + ((rem % W) >> 4);                  // 8 x 16 statistics tiles, row-major
        halo16_out_direct<OT, MT>(a, acc, smem, m_tl, W, n0, bsmp, tile0, W >> 4);
    } else {
    // ---- epilogue in two halves of 64 output channels (C tile [128][68] floats = 35 KB instead of 68 KB).  Half h is
    // held by the waves with wn == h; then all 256 threads run the shared output stage on it.
        constexpr int CROW = 68;
        float* Cs = smem;
        float* red = smem + 128 * CROW;
        const int bsmp = m_tl / HW;
        const int rem = m_tl - bsmp * HW;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int tile = (((rem / W) >> 3) + t) * (W >> 4) + ((rem % W) >> 4);     // 8 x 16 statistics tiles, row-major
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                __syncthreads();                           // previous users of the tile (main loop / earlier pass) are done
                if (wn == half) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                                Cs[row * CROW + jn * 32 + li] = acc[t][i][jn][r];
                            }
                }
                __syncthreads();
                tile128x64_out<OT>(a, Cs, CROW, red, m_tl + t * 8 * W, W, n0 + half * 64, bsmp, tile);
            }
        }
    }
#ifdef FLOWSE_TS
    if (ts_on) {
        __syncthreads();
        ts[7] = __builtin_amdgcn_s_memtime();
        ts[8] = __builtin_amdgcn_s_getreg(63492);          // HW_ID
        ts[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
        if (tid == 0 && bid < 8192)
            for (int k = 0; k < 10; ++k) g_ts[bid * 10 + k] = ts[k];
    }
#endif
}

template <bool F16>
static int launch_halo16(const ConvArgs& a, hipStream_t s) {
    using T16 = typename std::conditional<F16, f16_t, bf16_t>::type;
    const int64_t M = (int64_t)a.B * a.H * a.W;
    if (a.in_dt != a.out_dt || (a.in_dt != DT_F32 && a.in_dt != St<T16>::dt) || a.terms != 1 || a.partial) {
        set_error("halo16: input / output storage must agree and match the operand type; no split-K form");
        return ERR_ARG;
    }
    // 16 x 16 pixel tiles (two sub-tiles per block) when the input is 16-bit, H allows it and >= 512 blocks remain
    static const bool no_mt2 = getenv("FLOWSE_HALO16_MT1") != nullptr;                 // A-B hook
    const bool mt2 = !no_mt2 && a.in_dt != DT_F32 && (a.H & 15) == 0 && (M / 256) * (a.Cout / 128) >= 512;
    const int grid = (int)(M / (mt2 ? 256 : 128)) * (a.Cout / 128);
    const size_t lds_stage = (size_t)(mt2 ? 18 : 10) * (18 * 80 + 96) + (size_t)2 * 128 * 80;
    const size_t lds_epi = ((size_t)128 * 68 + 4 * 64 * 2) * sizeof(float);
    size_t lds = (a.in_dt == DT_F32 && lds_epi > lds_stage) ? lds_epi : lds_stage;      // the staged epilogue is the fp32-storage one
#ifdef FLOWSE_TS
    static const size_t lds_min = getenv("FLOWSE_HALO16_LDS") ? (size_t)atoi(getenv("FLOWSE_HALO16_LDS")) : 0;   // occupancy probe
    if (lds < lds_min) lds = lds_min;
#endif
#define FLOWSE_LH16(GNF, IT, OT, MTV)                                                                             \
    {                                                                                                             \
        if (const int rc = allow_lds<&conv3x3_halo16_kernel<GNF, F16, IT, OT, MTV>>(lds)) return rc;              \
        hipLaunchKernelGGL((conv3x3_halo16_kernel<GNF, F16, IT, OT, MTV>), dim3(grid), dim3(256), lds, s, a);     \
    }
    if (a.in_dt == DT_F32) {
        if (a.gn.mean) FLOWSE_LH16(true, float, float, 1) else FLOWSE_LH16(false, float, float, 1)
    } else if (mt2) {
        if (a.gn.mean) FLOWSE_LH16(true, T16, T16, 2) else FLOWSE_LH16(false, T16, T16, 2)
    } else {
        if (a.gn.mean) FLOWSE_LH16(true, T16, T16, 1) else FLOWSE_LH16(false, T16, T16, 1)
    }
#undef FLOWSE_LH16
    FLOWSE_LAUNCH_CHECK();
    return OK;
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
    static const bool off = getenv("FLOWSE_NO_HALO16") != nullptr;         // test / A-B hook: everything on the flat kernel
    if (off || taps != 9 || (H & 7) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout % 128)) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    if ((int64_t)(9 * W + 18) * cmax * 4 >= (1LL << 31) || (int64_t)Cout * 9 * (C1 + C2) * 4 >= (1LL << 31)) return false;
    return ((int64_t)B * H * W / 128) * (Cout / 128) >= 128;
}

int conv16_ksplit(int B, int H, int W, int Cin, int Cout, int taps) {
    const int64_t M = (int64_t)B * H * W;
    const int64_t tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
    const int stages = ((Cin / KC) * taps + 1) / 2;
    if (tiles >= 256 || stages < 4) return 1;
    int64_t ks = (512 + tiles - 1) / tiles;
    if (ks > stages / 2) ks = stages / 2;
    if (ks < 1) ks = 1;
    const int per = (int)((stages + ks - 1) / ks);
    return (int)((stages + per - 1) / per);
}

int conv16_stats_blocks(int B, int H, int W, int Cin, int Cout, int taps) {
    const int HW = H * W;
    if (Cout & 3) return 0;
    if (taps == 9 && conv16_uses_halo(B, H, W, Cin, 0, Cout, taps)) return HW / 128;     // C1/C2 split is irrelevant here
    if (conv16_ksplit(B, H, W, Cin, Cout, taps) != 1) {
        const int PB = sk_pixels_per_block(HW);
        return ((HW % PB) == 0 && Cout / 4 <= 256) ? HW / PB : 0;
    }
    return (HW % 128) == 0 ? HW / 128 : 0;
}

static int launch_flat16(const ConvArgs& a, hipStream_t s) {
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

int launch_splitk_reduce(const ConvArgs& a, hipStream_t s) {
    const int HW = a.H * a.W;
    if (a.stats) {
        const int PB = sk_pixels_per_block(HW);
        if (a.stats_nblk != HW / PB || (HW % PB) != 0 || a.Cout / 4 > 256) {
            set_error("splitk_reduce: inconsistent fused-stats geometry");
            return ERR_ARG;
        }
        FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_stats_kernel<OT>, dim3(HW / PB, a.B), dim3(256), 0, s,
                                                          a, PB));
        FLOWSE_LAUNCH_CHECK();
        return OK;
    }
    const unsigned per_sample = (unsigned)HW * (a.Cout / 4);
    FLOWSE_DT_SWITCH(a.out_dt, OT, hipLaunchKernelGGL(splitk_reduce_kernel<OT>, dim3((per_sample + 255) / 256, a.B),
                                                      dim3(256), 0, s, a));
    FLOWSE_LAUNCH_CHECK();
    return OK;
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
    if (a.in_dt != DT_F32) {                          // activations stored as bf16 / half
        if (a.ksplit <= 1 && !a.partial && !a.bias2 && !a.stats && a.out_dt == DT_F32 &&
            conv_supports_head4(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
            return launch_head4(a, s);
        if (a.ksplit <= 1 && a.out_dt == a.in_dt && conv16_uses_halo(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
            // the three-blocks-per-CU kernel wins once there are >= 1024 blocks (>= 4 / 3 rounds of 768); smaller grids
            // keep the two-blocks-per-CU form (FLOWSE_HALO16_PER_TAP=1 forces it everywhere: A-B hook)
            static const bool per_tap = getenv("FLOWSE_HALO16_PER_TAP") != nullptr;
            const int64_t blocks = ((int64_t)a.B * a.H * a.W / 128) * (a.Cout / 128);
            if (per_tap || blocks < 1024) return a.wq_f16 ? launch_halo_bf16<1, true>(a, s) : launch_halo_bf16<1>(a, s);
            return a.wq_f16 ? launch_halo16<true>(a, s) : launch_halo16<false>(a, s);
        }
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
    if (a.ksplit <= 1 && conv_supports_fused_gn(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        if (a.wq && (a.Cout % 128) == 0) {
            if (a.terms == 3 && !a.wq_f16) return launch_halo_bf16<3>(a, s);
            if (a.terms == 1) {
                const int64_t blocks = ((int64_t)a.B * a.H * a.W / 128) * (a.Cout / 128);
                if (blocks < 1024) return a.wq_f16 ? launch_halo_bf16<1, true>(a, s) : launch_halo_bf16<1>(a, s);
                return a.wq_f16 ? launch_halo16<true>(a, s) : launch_halo16<false>(a, s);
            }
            set_error("conv: 16-bit path needs terms = 1 (bf16 / f16) or 3 (bf16 only)");
            return ERR_ARG;
        }
        if (a.wino && conv_supports_wino(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps))
            return a.wino_f43 ? launch_f43(a, s) : launch_wino(a, s);
        if (a.Cout <= 32) return launch_halo<4, 1, 1, 1>(a, s);
        if (a.Cout <= 64) return launch_halo<2, 2, 2, 1>(a, s);
        // fewer than two 128x128 tiles per CU (single utterances): halve the N tile so that two blocks share every
        // CU and cover each other's barriers / prologues
        const int64_t tiles128 = ((int64_t)a.B * a.H * a.W / 128) * ((a.Cout + 127) / 128);
        if (tiles128 < 512) return launch_halo<2, 2, 2, 1>(a, s);
        return launch_halo<2, 2, 2, 2>(a, s);
    }
    if (a.ksplit > 1 && !a.partial) {
        set_error("conv: split-K needs a partial buffer");
        return ERR_ARG;
    }
    if (a.ksplit > 1 && a.wino && a.wino_f43 && conv_supports_wino(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps) &&
        a.ksplit == wino_plan(a.B, a.H, a.W, a.C1 + a.C2, a.Cout, a.taps)) {
        const int rc = launch_f43(a, s);              // gridDim.y = ksplit slices of chunks, raw partial tiles
        if (rc != OK || !with_reduce) return rc;
        return launch_splitk_reduce(a, s);
    }
    if (a.gn.mean) {
        set_error("conv: fused GroupNorm input requested for a shape the halo kernel does not cover");
        return ERR_ARG;
    }
    const int rc = a.Cout <= 32 ? launch_cfg<4, 1, 1, 1>(a, s)
                 : a.Cout <= 64 ? launch_cfg<2, 2, 2, 1>(a, s)
                 : conv_small_m((int64_t)a.B * a.H * a.W, a.Cout) ? launch_cfg<1, 4, 1, 1>(a, s) : launch_cfg<2, 2, 2, 2>(a, s);
    if (rc != OK || a.ksplit <= 1 || !with_reduce) return rc;
    return launch_splitk_reduce(a, s);
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
    static const bool off = getenv("FLOWSE_NO_CIN4_MFMA") != nullptr;      // test / A-B hook
    return !off && taps == 9 && Cout == 128 && ((H * W) % 128) == 0 && (int64_t)B * H * W >= 128 * 256 && !g_force_generic;
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
