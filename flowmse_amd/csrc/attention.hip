// Single-head self-attention over L = H*W tokens with C channels, fp32 on the CDNA4 matrix cores.
//
// Replaces (reference): the einsum / softmax / einsum core of AttnBlockpp.forward
// (flowmse/backbones/ncsnpp_utils/layerspp.py:82-86):
//     w = softmax_j( sum_c q[b,c,i] k[b,c,j] * C^-1/2 );  h[b,c,i] = sum_j w[i,j] v[b,c,j]
// q, k, v arrive token-major from one fused 1x1 projection: qkv[b][token][q(0:C) | k(C:2C) | v(2C:3C)].
//
// Flash-style, one 8-wave block per 32-query tile (the key tiles are dealt to the waves, whose online-softmax
// states are merged through LDS at the end), never materialising the L x L score matrix.  Both products are
// computed TRANSPOSED so that every per-query quantity (running max, running sum, rescale factor) is
// lane-local (lane & 31 = query):
//   S^T[key][query] = K Q^T   A = K rows (key = lane&31),       B = Q rows (query = lane&31)
//   O^T[ch][query]  = V^T P^T A = V[key(r, lane>>5)][ch = lane&31], B = P^T = the S^T accumulator registers:
//                     register r of the 32x32 C/D layout holds key (r&3)+8(r>>2) on lanes 0-31 and that key + 4
//                     on lanes 32-63, which is exactly the K=2 operand pair of v_mfma_f32_32x32x2_f32.
// Work is tiny (0.04 % of the network FLOPs), so the kernel favours simplicity.
#include "conv_common.h"

namespace flowse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_WAVES = 8;     // key tiles are dealt to 8 waves (L = 256 -> one 32-key tile each)
constexpr int ATT_SLOTS = 4;     // LDS tiles [32 queries][C + 4] the eight waves' O tiles are merged through

// The eight waves' rescaled O^T tiles -> O[query][channel] / l -> out.  Waves 0-3 write their tile into slot (wave & 3),
// waves 4-7 add theirs to it, then every thread sums the four slots in a fixed order, normalises and stores (bit-
// reproducible).  (Rounds 1-4 added the eight tiles one after the other into ONE slot, eight block barriers with a dependent
// LDS read-modify-write chain each: 12.8 of the kernel's 57 us at L = 256, C = 256 -- s_memrealtime stamps per phase.)
// Precondition: a block barrier since the last read of the query tile (slot 0 overlays it) and of sm_l.
template <int NCT, class OT>
__device__ __forceinline__ void att_merge_store(const f32x16 (&o)[NCT], float f, float l_tot, float* Os, float* sm_l,
                                                OT* __restrict__ out, int b, int L, int q0) {
    constexpr int C = 32 * NCT, OROW = C + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    float* slot = Os + (wave & (ATT_SLOTS - 1)) * (32 * OROW);
#pragma unroll 1
    for (int round = 0; round < ATT_WAVES / ATT_SLOTS; ++round) {
        if (wave / ATT_SLOTS == round) {
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4* p = reinterpret_cast<float4*>(slot + li * OROW + t * 32 + 8 * g + 4 * kh);
                    float4 v = make_float4(o[t][4 * g] * f, o[t][4 * g + 1] * f, o[t][4 * g + 2] * f, o[t][4 * g + 3] * f);
                    if (round > 0) {
                        const float4 old = *p;
                        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                    }
                    *p = v;
                }
        }
        // wave 0's row of sm_l takes l for every thread below -- in round 1: behind a barrier that every wave passes after its
        // last read of sm_l (l_tot)
        if (round == 1 && kh == 0 && wave == 0) sm_l[li] = l_tot;
        __syncthreads();
    }
    for (int i = tid; i < 32 * (C / 4); i += 64 * ATT_WAVES) {
        const int r = i / (C / 4), c4 = i - r * (C / 4);
        if (q0 + r >= L) continue;
        const float inv = 1.f / sm_l[r];
        const float* p = Os + r * OROW + c4 * 4;
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 32 * OROW);
        const float4 v2 = *reinterpret_cast<const float4*>(p + 2 * 32 * OROW), v3 = *reinterpret_cast<const float4*>(p + 3 * 32 * OROW);
        const float4 v = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z),
                                     (v0.w + v1.w) + (v2.w + v3.w));
        St<OT>::st4(out + ((int64_t)b * L + q0 + r) * C + c4 * 4, make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv));
    }
}

template <int NCT>   // C = 32 * NCT
__global__ __launch_bounds__(64 * ATT_WAVES) void attention_kernel(const float* __restrict__ qkv, int L,
                                                                   float* __restrict__ out, float scale) {
    constexpr int C = 32 * NCT;
    constexpr int QROW = C + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;                       // [32][QROW]: query tile; later slot 0 of the ATT_SLOTS merge tiles
    float* sm_m = sm + ATT_SLOTS * 32 * QROW;   // [ATT_WAVES][32] running max per wave
    float* sm_l = sm_m + ATT_WAVES * 32;  // [ATT_WAVES][32] running sum per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const int64_t rs = 3 * C;                                  // token stride
    const float* base = qkv + (int64_t)b * L * rs;

    // stage the query tile (zero rows beyond L)
    for (int i = tid; i < 32 * (C / 4); i += 64 * ATT_WAVES) {
        const int r = i / (C / 4), c4 = i - r * (C / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < L) v = *reinterpret_cast<const float4*>(base + (int64_t)(q0 + r) * rs + c4 * 4);
        *reinterpret_cast<float4*>(Qs + r * QROW + c4 * 4) = v;
    }
    __syncthreads();

    f32x16 o[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // the key tiles are dealt round-robin to the waves; each wave keeps its own online-softmax state
    for (int k0 = wave * 32; k0 < L; k0 += 32 * ATT_WAVES) {
        // ---- S^T tile
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = k0 + li;
        // (a key row past L reads row 0: its scores are masked to -inf below, its probabilities are exactly 0, so neither the K
        // nor the V operand needs a select -- a select on a loaded value pins the wait for that load to the spot where it was
        // issued, which is what had kept every one of these requests from running ahead: PV 31 us, S 10 us of 57.
        // Precondition: V row 0 finite, as for every key -- 0 x inf would be NaN; but row 0 is a real key of EVERY query, so a
        // non-finite V[0] poisons all outputs through its own strictly positive probability with or without this shortcut.)
        const float* kp = base + (int64_t)(krow < L ? krow : 0) * rs + C + kh * 4;
        const float* qp = Qs + li * QROW + kh * 4;
        // The loop is one chain of dependent MFMAs, so a load inside it is a full memory round trip that nothing hides, and
        // q / k / v were written by the previous launch (other XCDs' L2s: the round trip is ~2 us, not an L2 hit): the key
        // row runs through a ring of AR quads, each requested AR iterations (4 AR MFMAs) ahead of its use.  Round 5: AR 4 ->
        // 16 (a ring of four left ~1.5 us of every 0.12 us iteration exposed: the S tile took ~45 of the kernel's 60 us).
        constexpr int NQ = C / 8;
        constexpr int AR = NQ < 16 ? NQ : 16;
        float4 ak[AR];
#pragma unroll
        for (int u = 0; u < AR; ++u) ak[u] = *reinterpret_cast<const float4*>(kp + u * 8);
#pragma unroll 1
        for (int g = 0; g < NQ; g += AR) {
#pragma unroll
            for (int u = 0; u < AR; ++u) {
                const int j = g + u;
                const float4 a = ak[u];
                if (j + AR < NQ) ak[u] = *reinterpret_cast<const float4*>(kp + (j + AR) * 8);
                const float4 q = *reinterpret_cast<const float4*>(qp + j * 8);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q.x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q.y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q.z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q.w, s, 0, 0, 0);
            }
        }
        // ---- online softmax over keys (rows of S^T); this lane's query = li
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            s[r] = key < L ? s[r] * scale : -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);               // finite: every tile holds >= 1 valid key
        const float alpha = expf(m_run - m_new);            // first tile: exp(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // ---- O^T = alpha O^T + V^T P^T.  A channel tile's 16 V rows are requested together, TWO tiles ahead of their MFMAs
        // (requested at their use, every tile waited a full memory round trip: 8 x ~2 us).
        float vv[3][16];
        auto vload = [&](int t, float (&dst)[16]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                dst[r] = base[(int64_t)(key < L ? key : 0) * rs + 2 * C + t * 32 + li];
            }
        };
        vload(0, vv[0]);
        if (NCT > 1) vload(1, vv[1]);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            if (t + 2 < NCT) vload(t + 2, vv[(t + 2) % 3]);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[t % 3][r], s[r], o[t], 0, 0, 0);
        }
    }
    // ---- merge the per-wave states: m = max_w m_w, O = sum_w O_w e^{m_w - m}, l = sum_w l_w e^{m_w - m}
    if (kh == 0) {
        sm_m[wave * 32 + li] = m_run;
        sm_l[wave * 32 + li] = l_run;
    }
    __syncthreads();                       // also: every wave is done reading Qs
    float m_tot = -INFINITY;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) m_tot = fmaxf(m_tot, sm_m[w * 32 + li]);
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) l_tot += sm_l[w * 32 + li] * expf(sm_m[w * 32 + li] - m_tot);
    const float f = expf(m_run - m_tot);   // 0 for a wave that saw no key tile (m_run = -inf)
    att_merge_store<NCT, float>(o, f, l_tot, Qs, sm_l, out, b, L, q0);
}

template <int NCT>
static int launch_att(const float* qkv, int B, int L, float* out, hipStream_t s) {
    constexpr int C = 32 * NCT;
    constexpr size_t lds = (ATT_SLOTS * 32 * (C + 4) + 2 * ATT_WAVES * 32) * sizeof(float);
    // one block per CU: the four merge tiles need 135 KB at C = 256 -- gfx950's 160 KB of LDS, nothing smaller
    static_assert(lds <= 160 * 1024, "attention: the ATT_SLOTS merge tiles must fit the 160 KB of LDS of a gfx950 CU");
    if (const int rc = allow_lds<&attention_kernel<NCT>>(lds)) return rc;
    const dim3 grid((L + 31) / 32, B), block(64 * ATT_WAVES);
    hipLaunchKernelGGL(attention_kernel<NCT>, grid, block, lds, s, qkv, L, out, 1.0f / sqrtf((float)C));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}


// ---------------------------------------------------------------------------------------------------
// 16-bit form for the bf16 / fp16 storage modes (BASELINE configs 2 / 4: "MFMA attention"): q / k / v arrive as 16-bit
// tokens from the 16-bit qkv projection, both products run on v_mfma_f32_32x32x16_{bf16,f16} (one instruction = 16
// channels of Q K^T or 16 keys of P V: 16x fewer matrix instructions than the fp32 form), softmax state and the O
// accumulators stay fp32, the result is rounded once to the storage type.  Same transposed formulation and the same
// 8-wave split of the key tiles as attention_kernel.  P^T reaches the second product straight from the S^T accumulator
// registers: the instruction's K index j (0..7) of lane half kh is register 8 u + j of MFMA u = key
// ((8u + j) & 3) + 8 ((8u + j) >> 2) + 4 kh of the tile -- any order works as long as the V operand uses the same one.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));

template <int NCT, class ST>
__global__ __launch_bounds__(64 * ATT_WAVES) void attention16_kernel(const ST* __restrict__ qkv, int L, ST* __restrict__ out,
                                                                     float scale) {
    constexpr bool F16 = St<ST>::dt == DT_F16;
    constexpr int C = 32 * NCT;
    constexpr int QROWB = C * 2 + 16;                   // bytes per staged query row (16 B pad: conflict-free b128 reads)
    constexpr int OROW = C + 4;                         // floats per row of the merged O tile
    extern __shared__ __attribute__((aligned(16))) float sm[];
    char* Qs = reinterpret_cast<char*>(sm);             // [32][QROWB] 16-bit query tile; later (fp32) slot 0 of the merge tiles
    float* sm_m = sm + ATT_SLOTS * 32 * OROW;           // [ATT_WAVES][32]
    float* sm_l = sm_m + ATT_WAVES * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const int64_t rs = 3 * C;                           // token stride (elements)
    const ST* base = qkv + (int64_t)b * L * rs;

    for (int i = tid; i < 32 * (C / 8); i += 64 * ATT_WAVES) {          // query tile, 16 bytes per thread (zero rows beyond L)
        const int r = i / (C / 8), c8 = i - r * (C / 8);
        au32x4 v = {0u, 0u, 0u, 0u};
        if (q0 + r < L) v = *reinterpret_cast<const au32x4*>(base + (int64_t)(q0 + r) * rs + c8 * 8);
        *reinterpret_cast<au32x4*>(Qs + r * QROWB + c8 * 16) = v;
    }
    __syncthreads();

    f32x16 o[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = wave * 32; k0 < L; k0 += 32 * ATT_WAVES) {
        // ---- S^T tile: C / 16 MFMAs, the key row of this lane as 16-byte pieces (all requested up front: C / 16 <= 16)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = k0 + li;
        const ST* kp = base + (int64_t)(krow < L ? krow : 0) * rs + C + kh * 8;
        au32x4 ak[C / 16];
#pragma unroll
        for (int j = 0; j < C / 16; ++j) ak[j] = *reinterpret_cast<const au32x4*>(kp + j * 16);   // (no select: see the fp32 kernel)
#pragma unroll
        for (int j = 0; j < C / 16; ++j) {
            const au32x4 q = *reinterpret_cast<const au32x4*>(Qs + li * QROWB + (j * 16 + kh * 8) * 2);
            if (F16)
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(af16x8, ak[j]), __builtin_bit_cast(af16x8, q), s, 0, 0, 0);
            else
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ak[j]), __builtin_bit_cast(abf16x8, q), s, 0, 0, 0);
        }
        // ---- online softmax over keys (fp32), as in the fp32 kernel
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            s[r] = key < L ? s[r] * scale : -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // P^T operands: registers 8u .. 8u + 7 rounded to the storage type (the normaliser l above sums the UNROUNDED
        // probabilities, like the reference's softmax followed by a 16-bit matmul would)
        au32x4 pb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pb[u].x = St<ST>::pack2(s[8 * u + 0], s[8 * u + 1]);
            pb[u].y = St<ST>::pack2(s[8 * u + 2], s[8 * u + 3]);
            pb[u].z = St<ST>::pack2(s[8 * u + 4], s[8 * u + 5]);
            pb[u].w = St<ST>::pack2(s[8 * u + 6], s[8 * u + 7]);
        }
        // ---- O^T = alpha O^T + V^T P^T: per channel tile two MFMAs; A = V[key(u, j, kh)][channel 32 t + li], gathered
        // from global memory, requested TWO channel tiles ahead of its MFMAs (at its use every tile waited a full round trip)
        const unsigned short* vb = reinterpret_cast<const unsigned short*>(base) + 2 * C + li;
        unsigned short vv[3][16];
        auto vload = [&](int t, unsigned short (&dst)[16]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                dst[r] = vb[(int64_t)(key < L ? key : 0) * rs + t * 32];
            }
        };
        vload(0, vv[0]);
        if (NCT > 1) vload(1, vv[1]);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            if (t + 2 < NCT) vload(t + 2, vv[(t + 2) % 3]);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                au32x4 va;
                va.x = (unsigned)vv[t % 3][8 * u + 0] | ((unsigned)vv[t % 3][8 * u + 1] << 16);
                va.y = (unsigned)vv[t % 3][8 * u + 2] | ((unsigned)vv[t % 3][8 * u + 3] << 16);
                va.z = (unsigned)vv[t % 3][8 * u + 4] | ((unsigned)vv[t % 3][8 * u + 5] << 16);
                va.w = (unsigned)vv[t % 3][8 * u + 6] | ((unsigned)vv[t % 3][8 * u + 7] << 16);
                if (F16)
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(af16x8, va), __builtin_bit_cast(af16x8, pb[u]), o[t], 0, 0, 0);
                else
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, va), __builtin_bit_cast(abf16x8, pb[u]), o[t], 0, 0, 0);
            }
        }
    }
    // ---- merge the per-wave states (fp32), exactly as in the fp32 kernel
    if (kh == 0) {
        sm_m[wave * 32 + li] = m_run;
        sm_l[wave * 32 + li] = l_run;
    }
    __syncthreads();                       // also: every wave is done reading Qs
    float m_tot = -INFINITY;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) m_tot = fmaxf(m_tot, sm_m[w * 32 + li]);
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) l_tot += sm_l[w * 32 + li] * expf(sm_m[w * 32 + li] - m_tot);
    const float f = expf(m_run - m_tot);
    att_merge_store<NCT, ST>(o, f, l_tot, sm, sm_l, out, b, L, q0);
}

template <int NCT, class ST>
static int launch_att16(const void* qkv, int B, int L, void* out, hipStream_t s) {
    constexpr int C = 32 * NCT;
    constexpr size_t lds = (ATT_SLOTS * 32 * (C + 4) + 2 * ATT_WAVES * 32) * sizeof(float);  // the fp32 O tiles are the larger overlay
    static_assert(lds <= 160 * 1024, "attention16: the ATT_SLOTS merge tiles must fit the 160 KB of LDS of a gfx950 CU");
    if (const int rc = allow_lds<&attention16_kernel<NCT, ST>>(lds)) return rc;
    const dim3 grid((L + 31) / 32, B), block(64 * ATT_WAVES);
    hipLaunchKernelGGL((attention16_kernel<NCT, ST>), grid, block, lds, s, static_cast<const ST*>(qkv), L, static_cast<ST*>(out),
                       1.0f / sqrtf((float)C));
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

int launch_attention(const void* qkv, int B, int L, int C, void* out, hipStream_t s, int dt) {
    if (dt != DT_F32) {
#define FLOWSE_ATT16(NCT) (dt == DT_BF16 ? launch_att16<NCT, bf16_t>(qkv, B, L, out, s) : launch_att16<NCT, f16_t>(qkv, B, L, out, s))
        switch (C) {
            case 32:  return FLOWSE_ATT16(1);
            case 64:  return FLOWSE_ATT16(2);
            case 128: return FLOWSE_ATT16(4);
            case 256: return FLOWSE_ATT16(8);
            default: break;
        }
#undef FLOWSE_ATT16
        set_error("attention: unsupported channel count %d (32/64/128/256)", C);
        return ERR_SHAPE;
    }
    const float* q = static_cast<const float*>(qkv);
    float* o = static_cast<float*>(out);
    switch (C) {
        case 32:  return launch_att<1>(q, B, L, o, s);
        case 64:  return launch_att<2>(q, B, L, o, s);
        case 128: return launch_att<4>(q, B, L, o, s);
        case 256: return launch_att<8>(q, B, L, o, s);
        default:
            set_error("attention: unsupported channel count %d (32/64/128/256)", C);
            return ERR_SHAPE;
    }
}

}  // namespace flowse
