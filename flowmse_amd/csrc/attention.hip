// Single-head self-attention over L = H*W tokens with C channels, fp32 on the CDNA4 matrix cores.
//
// Replaces (reference): the einsum / softmax / einsum core of AttnBlockpp.forward
// (flowmse/backbones/ncsnpp_utils/layerspp.py:82-86):
//     w = softmax_j( sum_c q[b,c,i] k[b,c,j] * C^-1/2 );  h[b,c,i] = sum_j w[i,j] v[b,c,j]
// q, k, v arrive token-major from one fused 1x1 projection: qkv[b][token][q(0:C) | k(C:2C) | v(2C:3C)].
//
// Flash-style, one wave per 32-query tile, never materialising the L x L score matrix.  Both products are
// computed TRANSPOSED so that every per-query quantity (running max, running sum, rescale factor) is
// lane-local (lane & 31 = query):
//   S^T[key][query] = K Q^T   A = K rows (key = lane&31),       B = Q rows (query = lane&31)
//   O^T[ch][query]  = V^T P^T A = V[key(r, lane>>5)][ch = lane&31], B = P^T = the S^T accumulator registers:
//                     register r of the 32x32 C/D layout holds key (r&3)+8(r>>2) on lanes 0-31 and that key + 4
//                     on lanes 32-63, which is exactly the K=2 operand pair of v_mfma_f32_32x32x2_f32.
// Work is tiny (0.04 % of the network FLOPs), so the kernel favours simplicity.
#include "common.h"

namespace flowse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCT>   // C = 32 * NCT
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv, int L, float* __restrict__ out,
                                                       float scale) {
    constexpr int C = 32 * NCT;
    constexpr int QROW = C + 4;
    __shared__ __attribute__((aligned(16))) float Qs[32 * QROW];
    const int lane = threadIdx.x;
    const int li = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const int64_t rs = 3 * C;                                  // token stride
    const float* base = qkv + (int64_t)b * L * rs;

    // stage the query tile (zero rows beyond L)
    for (int i = lane; i < 32 * (C / 4); i += 64) {
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

    for (int k0 = 0; k0 < L; k0 += 32) {
        // ---- S^T tile
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = k0 + li;
        const bool kok = krow < L;
        const float* kp = base + (int64_t)(kok ? krow : 0) * rs + C + kh * 4;
        const float* qp = Qs + li * QROW + kh * 4;
#pragma unroll 4
        for (int j = 0; j < C / 8; ++j) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok) a = *reinterpret_cast<const float4*>(kp + j * 8);
            const float4 q = *reinterpret_cast<const float4*>(qp + j * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q.w, s, 0, 0, 0);
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
        // ---- O^T = alpha O^T + V^T P^T
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = 0.f;
                if (key < L) v = base[(int64_t)key * rs + 2 * C + t * 32 + li];
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[r], o[t], 0, 0, 0);
            }
        }
    }
    // ---- store O[query][channel]: this lane's query = li, channels t*32 + (r&3) + 8(r>>2) + 4kh
    const int qrow = q0 + li;
    if (qrow < L) {
        const float inv = 1.f / l_run;
        float* op = out + ((int64_t)b * L + qrow) * C;
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = t * 32 + 8 * g + 4 * kh;
                *reinterpret_cast<float4*>(op + c) = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv,
                                                                 o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
            }
    }
}

int launch_attention(const float* qkv, int B, int L, int C, float* out, hipStream_t s) {
    const dim3 grid((L + 31) / 32, B), block(64);
    const float scale = 1.0f / sqrtf((float)C);
    switch (C) {
        case 32:  hipLaunchKernelGGL(attention_kernel<1>, grid, block, 0, s, qkv, L, out, scale); break;
        case 64:  hipLaunchKernelGGL(attention_kernel<2>, grid, block, 0, s, qkv, L, out, scale); break;
        case 128: hipLaunchKernelGGL(attention_kernel<4>, grid, block, 0, s, qkv, L, out, scale); break;
        case 256: hipLaunchKernelGGL(attention_kernel<8>, grid, block, 0, s, qkv, L, out, scale); break;
        default:
            set_error("attention: unsupported channel count %d (32/64/128/256)", C);
            return ERR_SHAPE;
    }
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
