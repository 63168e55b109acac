// Shared device helpers of the convolution kernels (gfx950 only): tile epilogues, GroupNorm-on-load, buffer loads.
// Internal to the library's conv_*.hip translation units.
#pragma once
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

// ---- launchers / policies shared between the conv_*.hip translation units (dispatch: conv_dispatch.hip)
bool conv_small_m(int64_t M, int Cout);
bool conv_force_generic();                 // FLOWSE_FORCE_GENERIC_CONV=1 (test hook): the generic flat kernel only
int f43_plan(int B, int H, int W, int Cin, int Cout, int taps);   // 0: not an F(4,3) shape, 1: whole K, >= 2: slices of chunks
inline int sk_pixels_per_block(int HW) { return HW <= 8 ? HW : HW <= 1024 ? 8 : HW <= 4096 ? 32 : 64; }
int launch_flat_fp32(const ConvArgs& a, hipStream_t s);        // conv_flat.hip: flat fp32 kernels (1x1, small 3x3, split-K slices)
int launch_halo_fp32(const ConvArgs& a, hipStream_t s);        // conv_halo.hip: direct LDS-halo 3x3 (exact fmaf chain)
int launch_head4(const ConvArgs& a, hipStream_t s);            // conv_halo.hip: 3x3 to four output channels
int launch_f43(const ConvArgs& a, hipStream_t s);              // conv_f43.hip: F(4,3) Winograd 3x3 (whole K or slices)
int launch_1x1_stream(const ConvArgs& a, hipStream_t s);       // conv_1x1.hip: fp32 1x1 on large images, every wave its own GEMM
int launch_smallm(const ConvArgs& a, hipStream_t s);           // conv_smallm.hip: <= 2048 pixels, K split inside the block
int launch_w2d(const ConvArgs& a, hipStream_t s);              // conv_w2d.hip: F(4,3) x F(2,3) Winograd 3x3 (whole K, large images)
int launch_halo_bf16x3(const ConvArgs& a, hipStream_t s);      // conv16.hip: split-bf16 operands, fp32 storage
int launch_halo16_any(const ConvArgs& a, hipStream_t s);       // conv16.hip: 16-bit operands, LDS-halo 3x3
int launch_flat16(const ConvArgs& a, hipStream_t s);           // conv16.hip: 16-bit storage, flat 1x1 / small 3x3
int launch_smallm16b(const ConvArgs& a, hipStream_t s);        // conv16_smallm.hip: 16-bit storage, <= 2048 pixels, K split inside the block
int launch_pc16(const ConvArgs& a, hipStream_t s);             // conv16_pc.hip: 16-bit storage, producer / consumer LDS-halo 3x3

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
        return;                                // splitk_reduce[_stats|_gn] sums the slices and runs the epilogue
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

// buffer loads: out-of-image taps (conv zero padding), rows past M and channels past Cout are redirected to an
// out-of-range buffer offset, for which the hardware returns 0 -- no masking VALU, no branches
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

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

// x * sigmoid(x) with v_exp_f32 / v_rcp_f32 (about 2 ulp)
__device__ __forceinline__ float fast_silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// GroupNorm affine (+ SiLU) of one staged channel quad, written on float2 halves so that hipcc emits the packed
// fp32 VALU forms (v_pk_add / v_pk_fma / v_pk_mul); only v_exp_f32 / v_rcp_f32 stay scalar.  `keep` = 0 zeroes the
// quad (halo pixel outside the image: zero padding applies AFTER the activation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

}  // namespace flowse
