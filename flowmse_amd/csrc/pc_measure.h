// Measurement-only instrumentation of conv3x3_pc16_kernel (builds with -DFLOWSE_MEASURE, tools/pc16_ts.py): s_memtime
// accumulators per block and role.  Never part of the shipped library.
#pragma once
#include <hip/hip_runtime.h>

namespace flowse {
__device__ unsigned long long g_pc_ts[256 * 16];
}
extern "C" int flowse_debug_pc_ts(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(flowse::g_pc_ts), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#define PC_TS_DECL unsigned long long pc_t0 = 0, pc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PC_TS_START pc_t0 = __builtin_amdgcn_s_memtime();
// whole-kernel brackets: shader-clock ticks (slot 5) and the constant 100 MHz counter (slot 6) from kernel entry to the end of
// the role's loop -- calibrates ticks against wall time and shows what the per-phase accumulators do not cover
#define PC_TS_ENTRY const unsigned long long pc_e0 = __builtin_amdgcn_s_memtime(), pc_r0 = __builtin_amdgcn_s_memrealtime();
#define PC_TS_EXIT                                                  \
    pc_acc[5] = __builtin_amdgcn_s_memtime() - pc_e0;               \
    pc_acc[6] = __builtin_amdgcn_s_memrealtime() - pc_r0;         \
    pc_acc[7] = pc_r0;                                              /* absolute entry time (100 MHz): start skew between blocks */
// add the time since the last mark to accumulator K
#define PC_TS_ADD(K)                                               \
    {                                                              \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        pc_acc[K] += now - pc_t0;                                  \
        pc_t0 = now;                                               \
    }
#define PC_TS_FLUSH(BASE)                                                                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)                                          \
        for (int k = 0; k < 8; ++k) flowse::g_pc_ts[blockIdx.x * 16 + (BASE) + k] = pc_acc[k];
