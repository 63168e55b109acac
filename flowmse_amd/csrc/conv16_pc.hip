// Producer / consumer LDS-halo 3x3 convolution for the 16-bit storage modes (bf16 / fp16 activations, BASELINE configs
// 2 and 4): Conv_k(act(GroupNorm_k(x))) of ResnetBlockBigGANpp (flowmse/backbones/ncsnpp_utils/layerspp.py:246-249,
// 265-267 = ddpm_conv3x3, layers.py:118-124, behind nn.GroupNorm + SiLU), concat input of ncsnpp.py:337, temb bias
// (layerspp.py:262-263), skip (x + h)/sqrt(2) (:271-274) and the next GroupNorm's partial statistics.
//
// Structure (ONE persistent block per CU, eight waves, tiles of 16 x 16 pixels x 128 output channels dealt to the blocks in
// XCD-contiguous ranges):
//   * Waves 0-3 ("consumers", one per SIMD): A fragments from the LDS halo, B fragments STRAIGHT FROM L2 -- the weights are
//     kept a second time in MFMA fragment order (pc16_weights_kernel: one 1 KB line per wave-level B operand) and travel
//     through a ring of three register sets, requested three steps ahead -- and 16 MFMAs per (tap, 32-channel chunk) step.
//     Every other instruction of a step sits in the gap behind ONE MFMA (FLOWSE_PC_STEP).  No weight tile in LDS, hence no
//     barrier per step.
//   * Waves 4-7 ("producers", one per SIMD) stage halos: raw 16-byte pieces (8 channels of a pixel) requested two chunks
//     ahead, GroupNorm + SiLU (affine folded to one FMA, v_exp / v_rcp, one rounding to the operand type) one chunk ahead of the
//     barrier that needs them, into one of THREE halo buffers, in scalar fp32 instructions -- and (round 6) DRAIN the finished
//     tile: see "hand-off of a finished tile" below.
//   * ONE barrier per chunk (X): the consumers are done with the previous chunk's buffer, the staged chunks are visible.
//   * Output stage (round 6): the MFMAs run transposed, the accumulators start at the biases, and the consumers only scale,
//     round and write runs of four channels into a hand-off tile in LDS (pc16_out_hand); the producers store it (16-byte
//     write-through stores) and take the GroupNorm partial statistics of what is stored during the next tile's first two
//     chunk intervals.  Residuals are requested into the idle B ring during the tile's last three steps.  No block barrier
//     per tile.
// LDS: 3 halo buffers [18][18 px x 80 B + 96] + statistics scratch + the hand-off tile [256 px][64 NJ ch + 16 B] = 153 KB.
//
// How it got here (round 4; tools/pc16_ts.py = s_memtime accumulators per role against the 100 MHz counter, [8,.,256,256]
// 128 -> 128 with GroupNorm + SiLU input and residual; ideal MFMA time 8 tiles x 36 steps x 512 = 147 k cycles per block):
//   weights through three LDS slots, one barrier per step, halo bursts with the consumers parked: ~370 k cycles per block
//     (the speed of the all-waves-equal kernel of rounds 2-3);
//   B fragments from L2, one barrier per chunk:                                   338 k
//   + packed output stage with scalar-offset addressing, early residual requests: 324 k (9.7 k -> 6.4 k per tile's stage)
//   + one filler per MFMA gap (the compiler's order clumped 4 ds_read + 4 buffer_load + scalar adds behind one MFMA:
//     41 cycles per MFMA, now 36):                                                 324 k, producers now critical
//   + SCALAR fp32 in the staging waves: beside an MFMA stream a packed fp32 instruction costs ~20 cycles more than the two
//     scalar ones it replaces (MI355X_MICROARCH.md) -- 26 k -> 12 k cycles of staging per tile:             265 k = 0.56
//   What did NOT change anything: per-chunk window arithmetic hoisted / division-free cursors / parameters folded at use /
//   a fourth halo buffer (the producers' "requests" segment stays ~2 k cycles per chunk: vector-memory issue behind the
//   consumers' B loads -- B fragments are ~145 KB per chunk and CU through the vector-memory path, half of it the same
//   lines for the two pixel halves).
// Wall time moved less than cycles (204 -> 176-190 us): the part clocks itself down as the MFMA duty rises (1.66 GHz at
// 324 k cycles, 1.41 GHz at 265 k, same shape, different boxes 1.4-2.0 GHz) -- DVFS gives a cycle saving partly back.
#include "conv_common.h"
#ifdef FLOWSE_MEASURE
#include "pc_measure.h"
#else
#define PC_TS_DECL
#define PC_TS_START
#define PC_TS_ADD(K)
#define PC_TS_FLUSH(BASE)
#define PC_TS_ENTRY
#define PC_TS_EXIT
#endif

#ifndef FLOWSE_PC16_STORE_AUX
// Cache policy of the output stores: 16 = sc1, WRITE-THROUGH.  A plain store leaves its line dirty in the XCD's L2 and what is
// still dirty when the kernel ends is written back at the kernel boundary with every CU idle (MI355X_MICROARCH.md, row
// "boundary": + B / 6 TB/s behind B dirty bytes); written through, the bytes leave under the kernel's own compute, and the next
// kernel reads them from HBM / MALL either way (the XCDs' L2s are not coherent with each other).  A-B (tools/ab.py, bf16,
// [8,1,256,256]): plain 53.65 k, nt 54.5 k, sc1 55.2 k, sc0 sc1 55.2 k frames/s; -3.1 us per launch.  Only for 16-byte stores
// (a scalar sc1 store is one fabric write each).  The same policy on the fp32 Winograd kernels' stores measured +-0.0 %.
// Loads stay on the default policy: `nt` on the raw halo pieces measured -5 % (the halo overlap and the second channel block
// live on L2 hits), `nt` on the residual loads +-0.0 %.
#define FLOWSE_PC16_STORE_AUX 16
#endif

namespace flowse {

namespace {
constexpr int PC_ROWB = 80;                        // bytes per halo pixel / weight row: 32 x 16 bit + 16 pad (conflict-free b128 reads)
constexpr int PC_HPITCH = 18 * PC_ROWB + 96;       // halo image row: 1536 B = 0 mod 256
constexpr int PC_HBUF_X = 18 * PC_HPITCH;            // one halo buffer: 18 rows = 27 KB
constexpr int PC_NHBUF = 3;                        // halo buffers: the staging runs two chunks ahead of the MFMAs
constexpr int PCF_STEP = 2 * 64 * 16;                // fragment-order weights: bytes of one (32-channel block, chunk, tap) = 2 halves x 1 KB
constexpr int PCF_CHUNK = 9 * PCF_STEP;
constexpr int PC_RED = 2 * 4 * 64 * 2 * 4;         // statistics scratch of the drain: [2 strips][4 producer waves][64 channel pairs][mean, M2]
constexpr int PC_PIECES = 6;                       // 16-byte halo pieces per producer thread and chunk (324 x 4 / 256)
// hand-off tile of a finished (16 x 16 pixel, 64 NJ channel) item: [256 pixels][64 NJ channels] in the storage type, + 16 B per pixel
constexpr int pc_opitch(int nj) { return 128 * nj + 16; }
constexpr int pc_lds(int nj) { return PC_NHBUF * PC_HBUF_X + PC_RED + 256 * pc_opitch(nj); }

// position of one (tile, channel block) work item
struct PcItem {
    int b, y0, x0, n0;
};
}  // namespace

// ---- hand-off of a finished tile (round 6).  The MFMAs run TRANSPOSED (A = weight fragment, B = pixel fragment), so a
// lane of a consumer wave holds, per 32 x 32 accumulator tile, four runs of four CONSECUTIVE output channels of ONE pixel
// (register r: channel (r & 3) + 8 (r >> 2) + 4 kh, pixel li).  The accumulators start at bias + per-sample bias + shortcut
// bias (pc_init_acc), so the consumers' whole output stage is: [+ residual] * scale, ONE rounding to the storage type, one
// ds_write_b64 per run into the hand-off tile O[pixel][channel] -- ~160 instructions per wave and tile where the round-4/5
// stage (fp32 transposition through wave-private LDS tiles, bias / scale / statistics in the MFMA waves, a block barrier
// for the statistics exchange) took ~1 600 and a fifth of a four-chunk tile's time with the matrix pipe idle.  The
// PRODUCER waves drain O during the next tile's first two chunk intervals: 16-byte pieces LDS -> global (write-through),
// and the GroupNorm partial statistics of exactly what is stored, per channel PAIR with v_dot2c on the packed values
// (a pair never straddles a group: group sizes are even; both channels of a pair get (mean, M2 / 2), which merges to the
// pair's exact moments downstream).
template <bool F16>
__device__ __forceinline__ unsigned pc_pack2(float a, float b) {
    return St<typename std::conditional<F16, f16_t, bf16_t>::type>::pack2(a, b);
}
struct PcOut {
    __amdgpu_buffer_rsrc_t rs_res;                         // the residual's sample
    unsigned voff;                                         // this lane's byte offset: its pixel, its first channel
    int rowb;                                              // bytes per image row
};
// scalar byte offset of output round (t, i): image rows 8 t + 2 i (+ li >> 4) of the tile
__device__ __forceinline__ unsigned pc_osoff(const PcOut& o, int t, int i) {
    return (unsigned)__builtin_amdgcn_readfirstlane((8 * t + 2 * i) * o.rowb);
}
typedef unsigned int pc_u32x2 __attribute__((ext_vector_type(2)));
// Residuals and the next tile's B-fragment ring share the ring's 48 registers (explicitly: left to the allocator, early
// residual requests were spilled to scratch behind an s_waitcnt vmcnt(0)).  A round (t, i) needs 4 NJ eight-byte pieces
// (piece p = 4 j + q: channels 32 j + 8 q + 4 kh .. + 3 of the lane's pixel) = one ring entry.  Round r = 2 t + i:
//   residual of round 0 / 1 / 2 arrives in wb[0] / wb[1] / wb[2], requested by the caller after the tile's last MFMAs on that
//   entry (taps 6 / 7 / 8 of its last chunk); round 3's goes into wb[0] when round 0 has consumed it.
#define FLOWSE_PC_RQ(R, P) wb[R][NJ == 2 ? ((P) >> 2) : ((P) >> 1)][NJ == 2 ? (((P) >> 1) & 1) : 0]
#define FLOWSE_PC_RSET(R, P, D)                                                                                      \
    {                                                                                                                \
        u32x4 rq_ = __builtin_bit_cast(u32x4, FLOWSE_PC_RQ(R, P));                                                   \
        if ((P) & 1) { rq_.z = (D).x; rq_.w = (D).y; } else { rq_.x = (D).x; rq_.y = (D).y; }                        \
        FLOWSE_PC_RQ(R, P) = __builtin_bit_cast(bf16x8, rq_);                                                        \
    }
template <class OT, bool RES, int NJ>
__device__ __forceinline__ void pc16_out_hand(const ConvArgs& a, f32x16 (&acc)[2][2][NJ], char* Ow, bf16x8 (&wb)[3][2][NJ],
                                              const PcOut& po) {
    constexpr int OP = pc_opitch(NJ);
    const f32x2 scale = {a.scale, a.scale};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int p = 0; p < 4 * NJ; ++p) {
                const int j = p >> 2, q = p & 3;
                f32x2 x0 = {acc[t][i][j][4 * q], acc[t][i][j][4 * q + 1]}, x1 = {acc[t][i][j][4 * q + 2], acc[t][i][j][4 * q + 3]};
                if (RES) {
                    const u32x4 rr = __builtin_bit_cast(u32x4, FLOWSE_PC_RQ((2 * t + i) % 3, p));
                    float r0, r1, r2, r3;
                    St<OT>::unpack2((p & 1) ? rr.z : rr.x, r0, r1);
                    St<OT>::unpack2((p & 1) ? rr.w : rr.y, r2, r3);
                    x0 += f32x2{r0, r1};
                    x1 += f32x2{r2, r3};
                }
                x0 *= scale;
                x1 *= scale;
                const pc_u32x2 w = {St<OT>::pack2(x0.x, x0.y), St<OT>::pack2(x1.x, x1.y)};
                *reinterpret_cast<pc_u32x2*>(Ow + ((8 * t + 2 * i) * 16) * OP + (32 * j + 8 * q) * 2) = w;
            }
            if (RES && t == 0 && i == 0) {                 // round 3's residual into the registers round 0 just read
#pragma unroll
                for (int p = 0; p < 4 * NJ; ++p) {
                    const pc_u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(
                        po.rs_res, po.voff + (unsigned)((32 * (p >> 2) + 8 * (p & 3)) * 2), pc_osoff(po, 1, 1), 0);
                    FLOWSE_PC_RSET(0, p, d)
                }
            }
        }
}

template <int GN, bool F16, int NJ>
__global__ __launch_bounds__(512, 2) void conv3x3_pc16_kernel(ConvArgs a) {
    constexpr int NCH = 64 * NJ;                           // output channels per block (NJ 32-channel tiles per consumer wave)
    using T16 = typename std::conditional<F16, f16_t, bf16_t>::type;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    PC_TS_ENTRY
    char* Hs = reinterpret_cast<char*>(smem);              // [PC_NHBUF][18][PC_HPITCH]
    float* red = reinterpret_cast<float*>(Hs + PC_NHBUF * PC_HBUF_X);     // [2 strips][4 producer waves][NCH / 2][mean, M2]
    char* Ob = Hs + PC_NHBUF * PC_HBUF_X + PC_RED;         // hand-off tile [256 pixels][pc_opitch(NJ)]
    constexpr int OP = pc_opitch(NJ);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int C1 = a.C1, C2 = a.C2, Cin = C1 + C2;
    const int nchunks = Cin / KC;
    // folded 1x1 shortcut (ConvArgs::sc1): `ns` extra ONE-step chunks per tile after the nine-tap chunks -- 32 raw channels of
    // the shortcut's input each, staged UNSHIFTED (tile pixel (py, px) in halo slot (py, px)) so that the consumers' tap-0
    // addressing reads the centre pixel, no GroupNorm; B fragments = the 1x1 weights in the same fragment order
    const int ns = (a.SC1 + a.SC2) / KC;
    const int nct = nchunks + ns;                          // chunks per tile in this block's stream
    const int n_ntiles = a.Cout / NCH;
    const int tiles_x = W >> 4, tiles_img = tiles_x * (H >> 4);
    // ---- this block's items: XCD x (= blockIdx & 7, where the dispatcher puts the block) walks the contiguous item range
    // [I x / 8, I (x + 1) / 8) with its gridDim / 8 blocks interleaved, so neighbouring tiles (shared halo rows, the same
    // weights) run at the same time behind one L2.  Placement is a speed matter only.
    const int I = a.B * tiles_img * n_ntiles;
    const int G8 = (int)gridDim.x >> 3;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int i_lo = (int)((int64_t)I * xcd / 8), i_hi = (int)((int64_t)I * (xcd + 1) / 8);
    const int first = i_lo + slot;
    if (first >= i_hi) return;                             // (whole block: nobody reaches a barrier)
    const int n_items = (i_hi - first + G8 - 1) / G8;
    const int Ctot = n_items * nct;                        // chunks in this block's stream
    auto item_at = [&](int k) {                            // k-th item of this block
        const int it = first + k * G8;
        const int mt = it / n_ntiles, nt = it - mt * n_ntiles;
        PcItem p;
        p.b = mt / tiles_img;
        const int tt = mt - p.b * tiles_img;
        const int ty = tt / tiles_x;
        p.y0 = ty * 16;
        p.x0 = (tt - ty * tiles_x) * 16;
        p.n0 = nt * NCH;
        return p;
    };

    // ---- halo staging (the producers' job; in the prologue the consumer waves stage chunks 1 and 2 with the same piece map
    // while the producers stage chunk 0: a one-item block spent a quarter of its time with its MFMA waves parked at the
    // first barrier -- tools/pc16_ts.py: ~18 k of ~70 k cycles)
    const int ltid = tid & 255;                            // staging thread: producers always, consumers in the prologue
    const int octet = ltid & 3;                            // this thread's 8-channel group inside a 32-channel chunk
    const int hp0 = ltid >> 2;                             // halo pixels hp0 + 64 q
    // Per piece, fixed for the whole launch: its LDS byte offset inside a halo buffer and its byte offset inside the 18 x 18
    // window of either source tensor.  Per (tile, chunk) only a descriptor, a scalar offset and six selects remain: the
    // per-chunk window arithmetic and bounds tests (~100 VALU instructions) cost the staging waves as much as half a
    // normalisation burst (tools/pc16_ts.py: "requests").
    int hlds[PC_PIECES];
    unsigned offA[PC_PIECES], offB[PC_PIECES];
    unsigned pixo[PC_PIECES];                              // window pixel offset hy W + hx (the folded shortcut's sources)
    unsigned smask = 0;                                    // pieces inside the 16 x 16 tile when the window is staged unshifted
    unsigned hyx[PC_PIECES / 2];                           // window coordinates (hy | hx << 8), two per register: the per-tile bounds mask
#pragma unroll
    for (int q = 0; q < PC_PIECES; ++q) {
        const int hp = hp0 + 64 * q;
        const int hy = hp / 18, hx = hp - hy * 18;
        hlds[q] = hy * PC_HPITCH + hx * PC_ROWB + octet * 16;
        offA[q] = (unsigned)(((hy * W + hx) * C1 + octet * 8) * 2);
        offB[q] = (unsigned)(((hy * W + hx) * C2 + octet * 8) * 2);
        pixo[q] = (unsigned)(hy * W + hx);
        if (hy < 16 && hx < 16) smask |= 1u << q;
        const unsigned pk = (unsigned)hy | ((unsigned)hx << 8);
        if (q & 1) hyx[q >> 1] |= pk << 16; else hyx[q >> 1] = pk;
    }
    const bool last_valid = hp0 + 64 * (PC_PIECES - 1) < 324;      // piece 5 exists for 16 threads only
    const T16* in1p = reinterpret_cast<const T16*>(a.in1);
    const T16* in2p = reinterpret_cast<const T16*>(a.in2);
    const T16* sc1p = reinterpret_cast<const T16*>(a.sc1);
    const T16* sc2p = reinterpret_cast<const T16*>(a.sc2);
    const int SC1 = a.SC1, SC2 = a.SC2;
    const int wpix = 17 * W + 18;

    // stream position of the requests, advanced by one chunk per HLOAD (no division per request: every instruction of these
    // waves competes with the MFMA stream for issue, ~10 cycles apiece): chunk rq_chunk of item rq_k, the item's coordinates
    // and the in-image mask of this thread's pieces.  Beyond the stream's end the last chunk is requested again.
    int rq_g = -1, rq_k = 0, rq_chunk = -1;
    PcItem rq_p = item_at(0);
    unsigned rq_mask = 0;
    auto rq_item = [&]() {
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < PC_PIECES; ++q) {
            const unsigned hy = (hyx[q >> 1] >> ((q & 1) * 16)) & 0xffu, hx = (hyx[q >> 1] >> ((q & 1) * 16 + 8)) & 0xffu;
            const bool in = (q < PC_PIECES - 1 || last_valid) && (unsigned)(rq_p.y0 - 1 + (int)hy) < (unsigned)H &&
                            (unsigned)(rq_p.x0 - 1 + (int)hx) < (unsigned)W;
            m |= in ? (1u << q) : 0u;
        }
        rq_mask = m;
    };
    rq_item();
    auto rq_next = [&]() {
        if (rq_g + 1 >= Ctot && rq_g >= 0) return;         // (uniform) clamp: stay on the last chunk
        ++rq_g;
        if (++rq_chunk == nct) {
            rq_chunk = 0;
            ++rq_k;
            rq_p = item_at(rq_k);
            rq_item();
        }
    };
    // raw halo pieces (8 consecutive channels of a pixel per piece; out-of-image pixels read 0)
    u32x4 ra[PC_PIECES], rb[PC_PIECES];                    // two chunks in flight
    unsigned hin_a = 0, hin_b = 0;                         // in-image bits of the pieces held in ra / rb
    // GroupNorm parameters of the thread's 8 channels, two chunks' worth
    struct PcAff {
        float4 m[2], s[2], b[2];
    };
    PcAff pa, pb;
    // requests of the next stream chunk: raw pieces into R, their in-image bits into HIN, the GroupNorm parameters into Q
#define FLOWSE_PC_HLOAD(R, HIN, Q)                                                                                   \
    {                                                                                                                \
        rq_next();                                                                                                   \
        const int chunk = rq_chunk;                                                                                  \
        const bool sc = chunk >= nchunks;                  /* a chunk of the folded shortcut: raw, unshifted */     \
        const int c0 = (sc ? chunk - nchunks : chunk) * KC;                                                          \
        const bool second = c0 >= (sc ? SC1 : C1);                                                                   \
        const unsigned cs = (unsigned)(sc ? (second ? SC2 : SC1) : (second ? C2 : C1));                              \
        const int64_t wbase = ((int64_t)rq_p.b * H + rq_p.y0 - (sc ? 0 : 1)) * W + rq_p.x0 - (sc ? 0 : 1);           \
        const T16* srcp = sc ? (second ? sc2p : sc1p) : (second ? in2p : in1p);                                      \
        const uint64_t wsel = reinterpret_cast<uint64_t>(srcp + wbase * (int64_t)cs);                                \
        const uint64_t wuni = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wsel) |              \
                              ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wsel >> 32)) << 32); \
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(                                       \
            reinterpret_cast<T16*>(wuni), 0, __builtin_amdgcn_readfirstlane(wpix * (int)cs * 2), 0x00020000);        \
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((c0 - (second ? (sc ? SC1 : C1) : 0)) * 2);   \
        const unsigned msk = sc ? smask : rq_mask;                                                                   \
        _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) {                                                      \
            const unsigned offr = second ? offB[q] : offA[q];                                                        \
            const unsigned offs = (pixo[q] * cs + (unsigned)octet * 8u) * 2u;                                        \
            const unsigned off = ((msk >> q) & 1u) ? (sc ? offs : offr) : OOB;                                       \
            R[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0);                                        \
        }                                                                                                            \
        HIN = msk | (sc ? 0x80000000u : 0u);                                                                         \
        if (!sc) load_params(rq_p.b, chunk, Q);                                                                      \
    }
    // GroupNorm parameters of (sample b, chunk): REQUESTED here (raw mean / scale / beta of the thread's 8 channels), folded by
    // the burst that uses them a chunk later -- folded on arrival, every request waited ~1 900 cycles for these loads
    auto load_params = [&](int b, int chunk, PcAff& q) {
        if (!GN) return;
        const int cg = chunk * KC + octet * 8;
        const float* mp = a.gn.mean + (int64_t)b * Cin + cg;
        const float* sp = a.gn.scale + (int64_t)b * Cin + cg;
        const float* bp = a.gn.beta + cg;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            q.m[h] = *reinterpret_cast<const float4*>(mp + 4 * h);
            q.s[h] = *reinterpret_cast<const float4*>(sp + 4 * h);
            q.b[h] = *reinterpret_cast<const float4*>(bp + 4 * h);
        }
    };
    // The raw pieces RX (parameters Q) of one chunk are normalised into the halo buffer at byte offset HB, stage by stage over
    // ALL pieces (48 values): a lone wave hides no latency by itself -- piece after piece the dependent unpack -> fma -> exp ->
    // rcp -> mul chains ran at ~10 cycles per instruction, 48 independent values per stage keep the VALU issuing.  SCALAR
    // fp32 instructions on purpose (this translation unit is built with -fno-slp-vectorize): the staging now runs beside the
    // consumers' MFMA stream, where a packed fp32 instruction costs ~20 cycles more than the two scalar ones it replaces
    // (MI355X_MICROARCH.md; here: 26 k vs ... cycles of staging per tile, tools/pc16_ts.py).
#define FLOWSE_PC_BURST(RX, HINX, Q, HB)                                                                             \
    if (GN && !((HINX) >> 31)) {                                                                                     \
        /* folded affine: y = (x - mean) scale + beta = x sc + sh; SiLU exponent -log2(e) y = x ec + eh */           \
        float sc[8], sh[8], ec[8], eh[8];                                                                            \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                              \
            const float mm[4] = {Q.m[h].x, Q.m[h].y, Q.m[h].z, Q.m[h].w}, ss[4] = {Q.s[h].x, Q.s[h].y, Q.s[h].z, Q.s[h].w}; \
            const float bb[4] = {Q.b[h].x, Q.b[h].y, Q.b[h].z, Q.b[h].w};                                            \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                          \
                sc[4 * h + e] = ss[e];                                                                               \
                sh[4 * h + e] = __builtin_fmaf(-mm[e], ss[e], bb[e]);                                                \
                ec[4 * h + e] = -1.44269504088896341f * sc[4 * h + e];                                               \
                eh[4 * h + e] = -1.44269504088896341f * sh[4 * h + e];                                               \
            }                                                                                                        \
        }                                                                                                            \
        float v[PC_PIECES][8], z[PC_PIECES][8];                                                                      \
        _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) {                                                      \
            const unsigned wsrc[4] = {RX[q].x, RX[q].y, RX[q].z, RX[q].w};                                           \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) St<T16>::unpack2(wsrc[e], v[q][2 * e], v[q][2 * e + 1]);   \
        }                                                                                                            \
        if (GN == 2) {                                                                                               \
            _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)      \
                z[q][e] = __builtin_fmaf(v[q][e], ec[e], eh[e]);                                                     \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)          \
            v[q][e] = __builtin_fmaf(v[q][e], sc[e], sh[e]);                                                         \
        if (GN == 2) {                                                                                               \
            _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)      \
                z[q][e] = __builtin_amdgcn_exp2f(z[q][e]);                                                           \
            _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)      \
                z[q][e] += 1.f;                                                                                      \
            _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)      \
                z[q][e] = __builtin_amdgcn_rcpf(z[q][e]);                                                            \
            _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) _Pragma("unroll") for (int e = 0; e < 8; ++e)      \
                v[q][e] *= z[q][e];                                                                                  \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q) {                                                      \
            const unsigned keepm = (((HINX) >> q) & 1u) ? 0xffffffffu : 0u;   /* zero padding AFTER the activation */ \
            u32x4 t;                                                                                                 \
            t.x = pc_pack2<F16>(v[q][0], v[q][1]) & keepm;                                                           \
            t.y = pc_pack2<F16>(v[q][2], v[q][3]) & keepm;                                                           \
            t.z = pc_pack2<F16>(v[q][4], v[q][5]) & keepm;                                                           \
            t.w = pc_pack2<F16>(v[q][6], v[q][7]) & keepm;                                                           \
            if (q < PC_PIECES - 1 || last_valid) *reinterpret_cast<u32x4*>(Hs + (HB) + hlds[q]) = t;                 \
        }                                                                                                            \
    } else {                                                                                                         \
        _Pragma("unroll") for (int q = 0; q < PC_PIECES; ++q)                                                        \
            if (q < PC_PIECES - 1 || last_valid) *reinterpret_cast<u32x4*>(Hs + (HB) + hlds[q]) = RX[q];             \
    }

    if (wave < 4) {
        // ====================================================== consumers: A fragments from LDS, B fragments from L2, MFMA
        // their share of the prologue: chunk 1 -> buffer 1 (the producers: chunk 0, the requests of chunks 2 and 3).  Rounds 5 /
        // 6a had them stage chunk 2 as well: two normalisation bursts (~5.4 k cycles) in front of the first MFMA of every launch;
        // chunk 2 is only needed at the SECOND chunk barrier and the producers are idle during chunk 0 (all buffers full)
        rq_next();
        FLOWSE_PC_HLOAD(ra, hin_a, pa)
        FLOWSE_PC_BURST(ra, hin_a, pa, PC_HBUF_X)
        const int lane = tid & 63;
        const int wm = wave >> 1, wn = wave & 1;
        const int li = lane & 31, kh = lane >> 5;
        int abase[2];                                      // sub-tile 0; sub-tile t adds 8 image rows
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int py = 2 * (wm * 2 + i) + (li >> 4), px = li & 15;
            abase[i] = (py + 1) * PC_HPITCH + (px + 1) * PC_ROWB + kh * 16;
        }
        // weights in fragment order (pc16_weights_kernel): one buffer_load_b128 per wave = one MFMA B operand, 1 KB contiguous.
        // ONE descriptor spans the 3x3 fragments and (folded shortcut) the 1x1 fragments: byte offsets o1 / o2 from the lower one
        const char* wf1 = static_cast<const char*>(a.wfrag);
        const char* wf2 = ns ? static_cast<const char*>(a.wfrag_sc) : wf1;
        const char* wlow = wf2 < wf1 ? wf2 : wf1;
        const unsigned o1 = (unsigned)(wf1 - wlow), o2 = (unsigned)(wf2 - wlow);
        const unsigned e1 = o1 + (unsigned)(a.Cout * 9 * Cin * 2), e2 = o2 + (unsigned)(a.Cout * (a.SC1 + a.SC2) * 2);
        const __amdgpu_buffer_rsrc_t rsrcw =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wlow), 0, (int)(e1 > e2 ? e1 : e2), 0x00020000);
        const unsigned bvo = (unsigned)lane * 16u;
        const unsigned jstride = (unsigned)nchunks * PCF_CHUNK;   // bytes between two 32-channel blocks: nine-tap chunks ...
        const unsigned jshort = (unsigned)ns * PCF_STEP;          // ... and the shortcut's one-step chunks
        // byte offsets of chunk c of 32-channel block nb (w0) and of block nb + 1 (w1): no division, a handful of scalar
        // instructions per chunk
        auto chunk_off = [&](int nb, int c, unsigned& w0, unsigned& w1) {
            if (c < nchunks) {
                w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(o1 + (unsigned)(nb * nchunks + c) * PCF_CHUNK));
                w1 = w0 + jstride;
            } else {
                w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(o2 + (unsigned)(nb * ns + (c - nchunks)) * PCF_STEP));
                w1 = w0 + jshort;
            }
        };
        // The accumulators of an item START at bias + per-sample bias + shortcut bias of the lane's channels (transposed
        // product: register r of tile j = channel 32 j + (r & 3) + 8 (r >> 2) + 4 kh of pixel li), so the output stage has no
        // bias to add.  Branch-free: an absent table is a buffer of 0 records (loads return 0).
        f32x16 acc[2][2][NJ];
        auto init_acc = [&](const PcItem& p) {
            const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.Cout * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.bias2 ? a.bias2 + (int64_t)p.b * a.bias2_stride : a.bias2), 0, a.bias2 ? a.Cout * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias_x), 0, a.bias_x ? a.Cout * 4 : 0, 0x00020000);
            const unsigned c0 = (unsigned)((p.n0 + wn * (32 * NJ) + 4 * kh) * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned co = c0 + (unsigned)((32 * j + 8 * q) * 4);
                    const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rb1, co, 0, 0);
                    const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rb2, co, 0, 0);
                    const u32x4 vx = __builtin_amdgcn_raw_buffer_load_b128(rbx, co, 0, 0);
                    const float e[4] = {__uint_as_float(v1.x) + __uint_as_float(v2.x) + __uint_as_float(vx.x),
                                        __uint_as_float(v1.y) + __uint_as_float(v2.y) + __uint_as_float(vx.y),
                                        __uint_as_float(v1.z) + __uint_as_float(v2.z) + __uint_as_float(vx.z),
                                        __uint_as_float(v1.w) + __uint_as_float(v2.w) + __uint_as_float(vx.w)};
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[t][i][j][4 * q + r] = e[r];
                }
        };
        // A step = two half-steps of 16 channels (mh = 0, 1), 8 MFMAs each.  The A fragments of half-step h + 1 are requested
        // before the MFMAs of half-step h (two sets of 4); the B fragments of a whole step travel through a ring of three
        // register sets, i.e. they are requested three steps (>= 1 500 cycles) before their MFMAs.
        bf16x8 xa[2][2], ya[2][2];                         // [t][i]
        bf16x8 wb[3][2][NJ];                               // [ring][mh][j]

#define FLOWSE_PC_LOADA(FA, HOFF, TAP, MH)                                                                           \
    {                                                                                                                \
        constexpr int tapoff = ((TAP) / 3 - 1) * PC_HPITCH + ((TAP) % 3 - 1) * PC_ROWB + (MH) * 32;                  \
        const char* Hb = Hs + (HOFF) + tapoff;                                                                       \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int i = 0; i < 2; ++i)                  \
            FA[t][i] = *reinterpret_cast<const bf16x8*>(Hb + abase[i] + t * 8 * PC_HPITCH);                          \
    }
#define FLOWSE_PC_WLOAD(RING, S0, S1, TAPV)                                                                          \
    _Pragma("unroll") for (int mh = 0; mh < 2; ++mh) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                  \
        wb[RING][mh][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(                          \
            rsrcw, bvo, (j ? (S1) : (S0)) + (unsigned)((TAPV) * PCF_STEP + mh * 1024), 0));
        // One step at tap TAP of the current chunk (halo buffer offset hoff, B ring entry R = TAP % 3): 16 MFMAs with every other
        // instruction of the step placed in the gap behind ONE of them -- a lone MFMA wave hides at most ~5 issue slots per
        // 32-cycle MFMA (MI355X_MICROARCH.md), and in clumps (four ds_read + four buffer_load + their scalar adds behind one
        // MFMA, as the compiler's own order had it) the loop ran at 41 cycles per MFMA (tools/pc16_ts.py).  Gap after MFMA
        //   1-4   this tap's second-half A fragments (set Y),        7, 8   refill of ring entry R, first half (step + 3),
        //   9-12  the next step's first-half A fragments (set X),    15, 16 refill of ring entry R, second half.
        // Branch-free: across a chunk boundary the next halo is complete since the barrier of THIS chunk; in a tile's last
        // chunk the requests for "the next chunk" are redundant (the ring is re-requested after the output stage, entries 0-2
        // meanwhile take the residuals: FLOWSE_PC_RESLOAD).
#define FLOWSE_PC_A1(FA, T_, I_, HOFF, TAP, MH)                                                                      \
    FA[T_][I_] = *reinterpret_cast<const bf16x8*>(Hs + (HOFF) + (((TAP) / 3 - 1) * PC_HPITCH + ((TAP) % 3 - 1) * PC_ROWB + \
                                                                (MH) * 32 + (T_) * 8 * PC_HPITCH) + abase[I_]);      \
    __builtin_amdgcn_sched_barrier(0);
#define FLOWSE_PC_W1(R, MH, J_, S0, S1, TAPV)                                                                        \
    wb[R][MH][J_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(                                \
        rsrcw, bvo + (unsigned)(((TAPV) & 1) * PCF_STEP + (MH) * 1024),                                              \
        ((J_) ? (S1) : (S0)) + (unsigned)(((TAPV) >> 1) * 2 * PCF_STEP), 0));                                        \
    __builtin_amdgcn_sched_barrier(0);
#define FLOWSE_PC_M1(FA, T_, I_, J_, R, MH)                                                                          \
    if (F16)                                                                                                         \
        acc[T_][I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb[R][MH][J_]),           \
                                                                __builtin_bit_cast(f16x8, FA[T_][I_]), acc[T_][I_][J_], 0, 0, 0); \
    else                                                                                                             \
        acc[T_][I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[R][MH][J_], FA[T_][I_], acc[T_][I_][J_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define FLOWSE_PC_REFILL(R, MH, J_, TAP)                                                                             \
    if constexpr ((TAP) + 3 < 9) { FLOWSE_PC_W1(R, MH, J_, wc0, wc1, (TAP) + 3) } else { FLOWSE_PC_W1(R, MH, J_, wn0, wn1, (TAP) + 3 - 9) }
#define FLOWSE_PC_NEXTA(T_, I_, TAP)                                                                                 \
    if constexpr ((TAP) < 8) { FLOWSE_PC_A1(xa, T_, I_, hoff, (TAP) + 1, 0) } else { FLOWSE_PC_A1(xa, T_, I_, hnext, 0, 0) }
        // NJ = 1 (64-channel blocks): eight MFMAs per step, every A fragment requested four MFMAs ahead of its use
#define FLOWSE_PC_STEP_N1(TAP)                                                                                       \
    {                                                                                                                \
        constexpr int R = (TAP) % 3;                                                                                 \
        FLOWSE_PC_M1(xa, 0, 0, 0, R, 0) FLOWSE_PC_A1(ya, 0, 0, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 0, 1, 0, R, 0) FLOWSE_PC_A1(ya, 0, 1, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 1, 0, 0, R, 0) FLOWSE_PC_A1(ya, 1, 0, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 1, 1, 0, R, 0) FLOWSE_PC_A1(ya, 1, 1, hoff, (TAP), 1) FLOWSE_PC_REFILL(R, 0, 0, TAP)        \
        FLOWSE_PC_M1(ya, 0, 0, 0, R, 1) FLOWSE_PC_NEXTA(0, 0, TAP)                                                   \
        FLOWSE_PC_M1(ya, 0, 1, 0, R, 1) FLOWSE_PC_NEXTA(0, 1, TAP)                                                   \
        FLOWSE_PC_M1(ya, 1, 0, 0, R, 1) FLOWSE_PC_NEXTA(1, 0, TAP)                                                   \
        FLOWSE_PC_M1(ya, 1, 1, 0, R, 1) FLOWSE_PC_NEXTA(1, 1, TAP) FLOWSE_PC_REFILL(R, 1, 0, TAP)                    \
    }
#define FLOWSE_PC_SSTEP_N1(R)                                                                                        \
    {                                                                                                                \
        FLOWSE_PC_M1(xa, 0, 0, 0, R, 0) FLOWSE_PC_A1(ya, 0, 0, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 0, 1, 0, R, 0) FLOWSE_PC_A1(ya, 0, 1, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 1, 0, 0, R, 0) FLOWSE_PC_A1(ya, 1, 0, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 1, 1, 0, R, 0) FLOWSE_PC_A1(ya, 1, 1, hoff, 0, 1) FLOWSE_PC_W1(R, 0, 0, wn0, wn1, 0)        \
        FLOWSE_PC_M1(ya, 0, 0, 0, R, 1) FLOWSE_PC_A1(xa, 0, 0, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 0, 1, 0, R, 1) FLOWSE_PC_A1(xa, 0, 1, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 1, 0, 0, R, 1) FLOWSE_PC_A1(xa, 1, 0, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 1, 1, 0, R, 1) FLOWSE_PC_A1(xa, 1, 1, hnext, 0, 0) FLOWSE_PC_W1(R, 1, 0, wn0, wn1, 0)       \
    }
#define FLOWSE_PC_STEP(TAP)                                                                                          \
    if constexpr (NJ == 1) FLOWSE_PC_STEP_N1(TAP) else                                                               \
    {                                                                                                                \
        constexpr int R = (TAP) % 3;                                                                                 \
        FLOWSE_PC_M1(xa, 0, 0, 0, R, 0) FLOWSE_PC_A1(ya, 0, 0, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 0, 0, 1, R, 0) FLOWSE_PC_A1(ya, 0, 1, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 0, 1, 0, R, 0) FLOWSE_PC_A1(ya, 1, 0, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 0, 1, 1, R, 0) FLOWSE_PC_A1(ya, 1, 1, hoff, (TAP), 1)                                       \
        FLOWSE_PC_M1(xa, 1, 0, 0, R, 0)                                                                              \
        FLOWSE_PC_M1(xa, 1, 0, 1, R, 0)                                                                              \
        FLOWSE_PC_M1(xa, 1, 1, 0, R, 0) FLOWSE_PC_REFILL(R, 0, 0, TAP)                                               \
        FLOWSE_PC_M1(xa, 1, 1, 1, R, 0) FLOWSE_PC_REFILL(R, 0, 1, TAP)                                               \
        FLOWSE_PC_M1(ya, 0, 0, 0, R, 1) FLOWSE_PC_NEXTA(0, 0, TAP)                                                   \
        FLOWSE_PC_M1(ya, 0, 0, 1, R, 1) FLOWSE_PC_NEXTA(0, 1, TAP)                                                   \
        FLOWSE_PC_M1(ya, 0, 1, 0, R, 1) FLOWSE_PC_NEXTA(1, 0, TAP)                                                   \
        FLOWSE_PC_M1(ya, 0, 1, 1, R, 1) FLOWSE_PC_NEXTA(1, 1, TAP)                                                   \
        FLOWSE_PC_M1(ya, 1, 0, 0, R, 1)                                                                              \
        FLOWSE_PC_M1(ya, 1, 0, 1, R, 1)                                                                              \
        FLOWSE_PC_M1(ya, 1, 1, 0, R, 1) FLOWSE_PC_REFILL(R, 1, 0, TAP)                                               \
        FLOWSE_PC_M1(ya, 1, 1, 1, R, 1) FLOWSE_PC_REFILL(R, 1, 1, TAP)                                               \
    }
        // One step of a folded-shortcut chunk (ring entry R = its ordinal % 3): the STEP schedule with tap-0 addressing -- the
        // chunk's pixels are staged unshifted --, the refill of entry R comes from the shortcut step three ahead (wn0 / wn1, the
        // last one again past the end) and the next A fragments from the next chunk's buffer.
#define FLOWSE_PC_SSTEP(R)                                                                                           \
    if constexpr (NJ == 1) FLOWSE_PC_SSTEP_N1(R) else                                                                \
    {                                                                                                                \
        FLOWSE_PC_M1(xa, 0, 0, 0, R, 0) FLOWSE_PC_A1(ya, 0, 0, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 0, 0, 1, R, 0) FLOWSE_PC_A1(ya, 0, 1, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 0, 1, 0, R, 0) FLOWSE_PC_A1(ya, 1, 0, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 0, 1, 1, R, 0) FLOWSE_PC_A1(ya, 1, 1, hoff, 0, 1)                                           \
        FLOWSE_PC_M1(xa, 1, 0, 0, R, 0)                                                                              \
        FLOWSE_PC_M1(xa, 1, 0, 1, R, 0)                                                                              \
        FLOWSE_PC_M1(xa, 1, 1, 0, R, 0) FLOWSE_PC_W1(R, 0, 0, wn0, wn1, 0)                                           \
        FLOWSE_PC_M1(xa, 1, 1, 1, R, 0) FLOWSE_PC_W1(R, 0, 1, wn0, wn1, 0)                                           \
        FLOWSE_PC_M1(ya, 0, 0, 0, R, 1) FLOWSE_PC_A1(xa, 0, 0, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 0, 0, 1, R, 1) FLOWSE_PC_A1(xa, 0, 1, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 0, 1, 0, R, 1) FLOWSE_PC_A1(xa, 1, 0, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 0, 1, 1, R, 1) FLOWSE_PC_A1(xa, 1, 1, hnext, 0, 0)                                          \
        FLOWSE_PC_M1(ya, 1, 0, 0, R, 1)                                                                              \
        FLOWSE_PC_M1(ya, 1, 0, 1, R, 1)                                                                              \
        FLOWSE_PC_M1(ya, 1, 1, 0, R, 1) FLOWSE_PC_W1(R, 1, 0, wn0, wn1, 0)                                           \
        FLOWSE_PC_M1(ya, 1, 1, 1, R, 1) FLOWSE_PC_W1(R, 1, 1, wn0, wn1, 0)                                           \
    }
#define FLOWSE_PC_RESLOAD(R)                               /* residual of output round R into ring entry R (see pc16_out_hand) */ \
    if (tile_end && has_res) {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < 4 * NJ; ++p) {                                                         \
            const pc_u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(                                                 \
                po.rs_res, po.voff + (unsigned)((32 * (p >> 2) + 8 * (p & 3)) * 2), pc_osoff(po, (R) >> 1, (R) & 1), 0); \
            FLOWSE_PC_RSET(R, p, d)                                                                                  \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
        PC_TS_DECL
        PC_TS_START
        int cit = 0;                                       // chunk inside the current tile
        int kitem = 0;                                     // ordinal of the current item
        const bool has_res = a.res != nullptr;
        PcItem pit = item_at(0);
        PcOut po;
        po.rowb = W * a.Cout * 2;
        auto set_out = [&]() {                             // residual descriptor and lane offset of tile `pit`
            const int64_t sb = (int64_t)pit.b * H * W * a.Cout;
            const T16* rbase = reinterpret_cast<const T16*>(has_res ? a.res : a.out) + sb;
            po.rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<T16*>(rbase), 0, H * po.rowb, 0x00020000);
            po.voff = (unsigned)((((pit.y0 + 4 * wm + (li >> 4)) * W + pit.x0 + (li & 15)) * a.Cout + pit.n0 + wn * (32 * NJ) +
                                  4 * kh) * 2);
        };
        set_out();
        init_acc(pit);
        // this lane's corner of the hand-off tile: pixel (4 wm + (li >> 4), li & 15) of output round (0, 0), its first channel
        char* const Ow = Ob + ((4 * wm + (li >> 4)) * 16 + (li & 15)) * OP + (wn * (32 * NJ) + 4 * kh) * 2;
        const bool small_k = nct < 3;                      // fewer than three chunk intervals per tile: drained at the tile's end
        int hoff = 0;                                      // byte offset of the current chunk's halo buffer
        int nb_cur = (pit.n0 >> 5) + wn * NJ;              // this wave's first 32-channel block in the current / the next item
        auto nb_of = [&](int k) { return ((first + k * G8) % n_ntiles) * (2 * NJ) + wn * NJ; };
        int nb_next = nb_of(min(1, n_items - 1));
        unsigned wc0, wc1, wn0, wn1;                       // scalar byte offsets: this chunk's fragments (block j = 0 / 1), the next one's
        chunk_off(nb_cur, 0, wc0, wc1);
        FLOWSE_PC_WLOAD(0, wc0, wc1, 0) FLOWSE_PC_WLOAD(1, wc0, wc1, 1) FLOWSE_PC_WLOAD(2, wc0, wc1, 2)
        __syncthreads();                                   // (X 0)
        FLOWSE_PC_LOADA(xa, 0, 0, 0)
        // One folded-shortcut chunk (ordinal S of the tile, ring entry R): its own chunk barrier, one step.  The refill of entry R
        // comes from the shortcut step three ahead (the last one again past the end: never used).
#define FLOWSE_PC_SCHUNK(R, S)                                                                                       \
    {                                                                                                                \
        const int hnext = hoff == (PC_NHBUF - 1) * PC_HBUF_X ? 0 : hoff + PC_HBUF_X;                                 \
        __syncthreads();                                   /* (X) */                                                \
        PC_TS_ADD(1)                                                                                                 \
        chunk_off(nb_cur, nchunks + min((S) + 3, ns - 1), wn0, wn1);                                                 \
        FLOWSE_PC_SSTEP(R)                                                                                           \
        PC_TS_ADD(0)                                                                                                 \
        hoff = hnext;                                                                                                \
    }
        for (int gc = 0; gc < Ctot;) {
            const bool tile_end = cit == nchunks - 1;      // the tile's last NINE-TAP chunk
            {
                const int hnext = hoff == (PC_NHBUF - 1) * PC_HBUF_X ? 0 : hoff + PC_HBUF_X;
                if (gc > 0) __syncthreads();               // (X gc) the halos of chunks gc and gc + 1 are in LDS; the producers may
                PC_TS_ADD(1)                               //        overwrite the buffer of chunk gc - 1.   1: chunk barrier
                chunk_off(nb_cur, cit, wc0, wc1);
                if (!tile_end || ns) chunk_off(nb_cur, cit + 1, wn0, wn1);     // (cit + 1 = nchunks: the shortcut's first steps)
                else chunk_off(nb_next, 0, wn0, wn1);
                FLOWSE_PC_STEP(0) FLOWSE_PC_STEP(1) FLOWSE_PC_STEP(2) FLOWSE_PC_STEP(3) FLOWSE_PC_STEP(4) FLOWSE_PC_STEP(5)
                FLOWSE_PC_STEP(6) FLOWSE_PC_RESLOAD(0) FLOWSE_PC_STEP(7) FLOWSE_PC_RESLOAD(1) FLOWSE_PC_STEP(8) FLOWSE_PC_RESLOAD(2)
                PC_TS_ADD(0)                               // 0: fragments + MFMA issue
                hoff = hnext;
            }
            ++cit;
            ++gc;
            if (tile_end) {
                if (ns) {                                  // the folded shortcut: ns one-step chunks, ring entries 0, 1, 2, 0, ...
                    for (int sq = 0; sq < ns; sq += 3) {
                        FLOWSE_PC_SCHUNK(0, sq)
                        if (sq + 1 < ns) {
                            FLOWSE_PC_SCHUNK(1, sq + 1)
                            if (sq + 2 < ns) FLOWSE_PC_SCHUNK(2, sq + 2)
                        }
                    }
                    gc += ns;
                }
                // the tile is complete: rounded values into the hand-off tile (the producers drain it under the next tile's first
                // chunks: the next chunk barrier publishes it, and they are done before this wave writes it again), next tile
                if (has_res) pc16_out_hand<T16, true, NJ>(a, acc, Ow, wb, po);
                else pc16_out_hand<T16, false, NJ>(a, acc, Ow, wb, po);
                if (small_k) {                             // (launch-uniform) no interval to drain under: the producers drain here
                    __syncthreads();
                    __syncthreads();
                }
                cit = 0;
                ++kitem;
                pit = item_at(min(kitem, n_items - 1));
                nb_cur = nb_next;
                nb_next = nb_of(min(kitem + 1, n_items - 1));
                set_out();
                // the next tile's first operands are requested BEFORE the biases are waited for (one memory round trip in front
                // of the tile's first MFMA instead of two)
                if (has_res || ns) {                       // (else the ring still holds the refills of taps 6-8: the next tile's
                    chunk_off(nb_cur, 0, wc0, wc1);        //  steps 0-2)
                    FLOWSE_PC_WLOAD(0, wc0, wc1, 0) FLOWSE_PC_WLOAD(1, wc0, wc1, 1) FLOWSE_PC_WLOAD(2, wc0, wc1, 2)
                }
                FLOWSE_PC_LOADA(xa, hoff, 0, 0)            // (in LDS since the barrier of the finished chunk; a stale read after the last tile is never used)
                __builtin_amdgcn_sched_barrier(0);
                init_acc(pit);
                PC_TS_ADD(3)                               // 3: hand-off + the next tile's set-up
            }
        }
#undef FLOWSE_PC_SCHUNK
        if (!small_k) {                                    // the last tile: published by the first barrier, drained before the second
            __syncthreads();
            __syncthreads();
        }
        PC_TS_EXIT
        if (wave == 0) { PC_TS_FLUSH(0) }                  // slots 0-7 of the block
#undef FLOWSE_PC_RESLOAD
#undef FLOWSE_PC_SSTEP
#undef FLOWSE_PC_SSTEP_N1
#undef FLOWSE_PC_STEP_N1
#undef FLOWSE_PC_STEP
#undef FLOWSE_PC_NEXTA
#undef FLOWSE_PC_REFILL
#undef FLOWSE_PC_M1
#undef FLOWSE_PC_W1
#undef FLOWSE_PC_A1
#undef FLOWSE_PC_WLOAD
#undef FLOWSE_PC_LOADA
        return;
    }

    // =================================================================================== producers: halo staging + drain
    // ---- prologue: chunk 0 -> buffer 0 (chunk 1: the consumer waves, above); raw pieces of chunks 2 and 3 in flight
    FLOWSE_PC_HLOAD(ra, hin_a, pa)
    rq_next();                                             // (chunk 1)
    FLOWSE_PC_HLOAD(rb, hin_b, pb)                         // chunk 2
    FLOWSE_PC_BURST(ra, hin_a, pa, 0)
    FLOWSE_PC_HLOAD(ra, hin_a, pa)                         // chunk 3
    PC_TS_DECL
    PC_TS_START
    // ---- drain of the hand-off tile O (written by the consumers at the end of a tile, published by the next chunk barrier):
    // a thread owns channel octet d_o of the pixels (row 8 T + RPP k + (d_pg >> 4), column d_pg & 15), k < NPH, of strip T
    // (= the 8 x 16-pixel statistics block).  Strip 0 leaves in the next tile's first chunk interval, strip 1 in its second;
    // each strip's partial statistics meet in `red` and are finished by NCH threads one interval later.
    constexpr int OCT = 8 * NJ;                            // 16-byte pieces per pixel
    constexpr int NPH = 4 * NJ;                            // pieces per thread and strip
    constexpr int RPP = 2 / NJ;                            // image rows per pass of the 256 threads
    const int d_o = ltid & (OCT - 1), d_pg = ltid / OCT;
    const int d_lds = ((d_pg >> 4) * 16 + (d_pg & 15)) * OP + d_o * 16;
    const int d_rowb = W * a.Cout * 2;
    const int pwave = wave - 4;
    PcItem dp = item_at(0);                                // the item being drained
    __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T16*>(a.out), 0, 0, 0x00020000);
    unsigned d_voff = 0;
    auto drain_item = [&](int k) {
        dp = item_at(k);
        d_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T16*>(a.out) + (int64_t)dp.b * H * W * a.Cout, 0, H * d_rowb, 0x00020000);
        d_voff = (unsigned)((((dp.y0 + (d_pg >> 4)) * W + dp.x0 + (d_pg & 15)) * a.Cout + dp.n0 + d_o * 8) * 2);
    };
    // One strip: NPH 16-byte pieces LDS -> global (write-through, see FLOWSE_PC16_STORE_AUX), and the moments of what is
    // stored per channel PAIR: S1 = sum x, S2 = sum x^2 over the thread's 2 NPH values with ONE v_dot2c each per packed word
    // (fp32 accumulation of exact products; half: about the thread's first value as pivot -- 22-bit squares do not sum exactly),
    // equal-count Chan merges over the lanes that share the octet (64 values), then `red`.
#define FLOWSE_PC_DRAIN(TS)                                                                                          \
    {                                                                                                                \
        u32x4 pc[NPH];                                                                                               \
        _Pragma("unroll") for (int k = 0; k < NPH; ++k)                                                              \
            pc[k] = *reinterpret_cast<const u32x4*>(Ob + d_lds + ((8 * (TS) + RPP * k) * 16) * OP);                  \
        _Pragma("unroll") for (int k = 0; k < NPH; ++k)                                                              \
            __builtin_amdgcn_raw_buffer_store_b128(pc[k], d_rs, d_voff,                                              \
                (unsigned)__builtin_amdgcn_readfirstlane((8 * (TS) + RPP * k) * d_rowb), FLOWSE_PC16_STORE_AUX);     \
        if (a.stats) {                                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                          \
                const unsigned w0 = e == 0 ? pc[0].x : e == 1 ? pc[0].y : e == 2 ? pc[0].z : pc[0].w;               \
                float s1 = 0.f, s2 = 0.f, pv = 0.f;                                                                  \
                if (F16) {                                                                                           \
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));                                       \
                    const unsigned pk = (w0 & 0xffffu) * 0x10001u;                                                   \
                    pv = (float)__builtin_bit_cast(_Float16, (unsigned short)(w0 & 0xffffu));                        \
                    _Pragma("unroll") for (int k = 0; k < NPH; ++k) {                                                \
                        const unsigned wk = e == 0 ? pc[k].x : e == 1 ? pc[k].y : e == 2 ? pc[k].z : pc[k].w;       \
                        const h2_t d = __builtin_bit_cast(h2_t, wk) - __builtin_bit_cast(h2_t, pk);                  \
                        s1 = __builtin_amdgcn_fdot2(d, __builtin_bit_cast(h2_t, 0x3c003c00u), s1, false);            \
                        s2 = __builtin_amdgcn_fdot2(d, d, s2, false);                                                \
                    }                                                                                                \
                } else {                                                                                             \
                    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));                                         \
                    _Pragma("unroll") for (int k = 0; k < NPH; ++k) {                                                \
                        const unsigned wk = e == 0 ? pc[k].x : e == 1 ? pc[k].y : e == 2 ? pc[k].z : pc[k].w;       \
                        s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, wk), __builtin_bit_cast(b2_t, 0x3f803f80u), s1, false); \
                        s2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, wk), __builtin_bit_cast(b2_t, wk), s2, false); \
                    }                                                                                                \
                }                                                                                                    \
                constexpr float inv = 1.f / (2 * NPH);                                                               \
                const float mr = s1 * inv;                 /* exact: a power of two */                               \
                float m2 = fmaxf(__builtin_fmaf(-mr, s1, s2), 0.f);                                                  \
                float mean = pv + mr;                                                                                \
                float cnt = (float)(2 * NPH);                                                                        \
                _Pragma("unroll") for (int off = OCT; off < 64; off <<= 1) {                                         \
                    const float mo = __shfl_xor(mean, off), qo = __shfl_xor(m2, off);                                \
                    const float dd = mo - mean;                                                                      \
                    m2 = m2 + qo + dd * dd * (0.5f * cnt);                                                           \
                    mean = 0.5f * (mean + mo);                                                                       \
                    cnt *= 2.f;                                                                                      \
                }                                                                                                    \
                if ((tid & 63) < OCT) {                                                                              \
                    float* dst = red + ((((TS) * 4 + pwave) * (NCH / 2)) + d_o * 4 + e) * 2;                         \
                    dst[0] = mean;                                                                                   \
                    dst[1] = m2;                                                                                     \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }
    // the four producer waves' partials (64 values each) of strip TS -> the (mean, M2) of the strip's 128 pixels, per channel:
    // both channels of a pair get (mean, M2 / 2) of the pair's 256 values
#define FLOWSE_PC_FINAL(TS)                                                                                          \
    if (a.stats && ltid < NCH) {                                                                                     \
        const float* r0 = red + (((TS) * 4 + 0) * (NCH / 2) + (ltid >> 1)) * 2;                                      \
        const float* r1 = r0 + (NCH / 2) * 2;                                                                        \
        const float* r2 = r1 + (NCH / 2) * 2;                                                                        \
        const float* r3 = r2 + (NCH / 2) * 2;                                                                        \
        const float da = r1[0] - r0[0], db = r3[0] - r2[0];                                                          \
        const float ma = 0.5f * (r0[0] + r1[0]), mb = 0.5f * (r2[0] + r3[0]);                                        \
        const float qa = r0[1] + r1[1] + da * da * 32.f, qb = r2[1] + r3[1] + db * db * 32.f;                        \
        const float dc = mb - ma;                                                                                    \
        const int tile0 = (dp.y0 >> 3) * tiles_x + (dp.x0 >> 4);                                                     \
        float* dst = a.stats + (((int64_t)dp.b * a.stats_nblk + tile0 + (TS) * tiles_x) * a.Cout + dp.n0 + ltid) * 2; \
        dst[0] = 0.5f * (ma + mb);                                                                                   \
        dst[1] = 0.5f * (qa + qb + dc * dc * 64.f);                                                                  \
    }
    const bool small_k = nct < 3;                          // fewer than three chunk intervals per tile: drained at the tile's end
    int cit = 0;                                           // chunk (of the tile) of the current interval
    int kt = 0;                                            // ordinal of the tile the consumers work on
    int hb = (PC_NHBUF - 1) * PC_HBUF_X;                   // buffer (byte offset) of chunk g + 2
    // One chunk interval g: after the chunk barrier the consumers work on chunk g and are done with chunk g - 1, whose buffer
    // takes chunk g + 2 (raw pieces RX, requested two intervals ago); then the raw pieces of chunk g + 4 are requested into the
    // same registers.  Three halo buffers (round 6: the fourth made room for the hand-off tile; rounds 4-5 measured three and
    // four equal -- per chunk the staging waves need as long as the MFMA waves, ~2 000 of their cycles in the request segment:
    // twelve vector-memory instructions queued behind the consumers' 36 B-fragment loads per wave and chunk).  The drain of
    // the previous tile rides in front: stores first, so that they are older than this interval's requests.
#define FLOWSE_PC_LCHUNK(STAGE, RX, HINX, Q)                                                                         \
    {                                                                                                                \
        const bool tile_end = cit == nct - 1;                                                                        \
        __syncthreads();                                   /* (X g) */                                              \
        PC_TS_ADD(1)                                       /* 1: chunk barrier */                                   \
        if (!small_k && kt > 0) {                                                                                    \
            if (cit == 0) {                                                                                          \
                drain_item(kt - 1);                                                                                  \
                FLOWSE_PC_DRAIN(0)                                                                                   \
            } else if (cit == 1) {                                                                                   \
                FLOWSE_PC_FINAL(0)                                                                                   \
                FLOWSE_PC_DRAIN(1)                                                                                   \
            } else if (cit == 2) {                                                                                   \
                FLOWSE_PC_FINAL(1)                                                                                   \
            }                                                                                                        \
        }                                                                                                            \
        PC_TS_ADD(3)                                       /* 3: drain */                                           \
        if (STAGE) {                                                                                                 \
            if (g + 2 < Ctot) { FLOWSE_PC_BURST(RX, HINX, Q, hb) }                                                   \
            hb = hb == (PC_NHBUF - 1) * PC_HBUF_X ? 0 : hb + PC_HBUF_X;                                              \
            PC_TS_ADD(4)                                   /* 4: halo burst */                                      \
            FLOWSE_PC_HLOAD(RX, HINX, Q)                    /* chunk g + 4 */                                        \
            PC_TS_ADD(0)                                   /* 0: requests */                                        \
        }                                                                                                            \
        ++cit;                                                                                                       \
        if (tile_end) {                                                                                              \
            cit = 0;                                                                                                 \
            if (small_k) {                                 /* (launch-uniform) the consumers' hand-off is complete after the first barrier */ \
                __syncthreads();                                                                                     \
                drain_item(kt);                                                                                      \
                FLOWSE_PC_DRAIN(0)                                                                                   \
                FLOWSE_PC_DRAIN(1)                                                                                   \
                __syncthreads();                                                                                     \
                FLOWSE_PC_FINAL(0)                                                                                   \
                FLOWSE_PC_FINAL(1)                                                                                   \
            }                                                                                                        \
            ++kt;                                                                                                    \
        }                                                                                                            \
    }
    for (int g = 0; g < Ctot; g += 2) {
        FLOWSE_PC_LCHUNK(true, rb, hin_b, pb)              // rb holds chunk g + 2 (interval 0: chunk 2, needed at the second barrier)
        if (g + 1 < Ctot) {
            const int g1 = g + 1;
            {
                const int g = g1;
                FLOWSE_PC_LCHUNK(true, ra, hin_a, pa)      // ra holds chunk g + 2
            }
        }
    }
    if (!small_k) {                                        // the last tile
        __syncthreads();
        drain_item(n_items - 1);
        FLOWSE_PC_DRAIN(0)
        FLOWSE_PC_DRAIN(1)
        __syncthreads();
        FLOWSE_PC_FINAL(0)
        FLOWSE_PC_FINAL(1)
    }
#undef FLOWSE_PC_LCHUNK
#undef FLOWSE_PC_FINAL
#undef FLOWSE_PC_DRAIN
#undef FLOWSE_PC_BURST
    PC_TS_EXIT
    if (wave == 4) { PC_TS_FLUSH(8) }                      // slots 8-15 of the block
#undef FLOWSE_PC_HLOAD
}

// Weights of a 3x3 conv in MFMA fragment order for the consumers of conv3x3_pc16_kernel:
//   dst[nb][chunk][tap][mh][lane = kh * 32 + li][8] = w[nb * 32 + li][tap][chunk * 32 + mh * 16 + kh * 8 + 0..7]
// (w = [Cout][taps][Cin] in the 16-bit operand type; taps = 1: the 1x1 convs, for the 16-bit small-image kernel), i.e. one 1 KB line per wave-level B operand of v_mfma_f32_32x32x16.
__global__ __launch_bounds__(256) void pc16_weights_kernel(const uint16_t* __restrict__ w, int Cout, int Cin,
                                                          uint16_t* __restrict__ dst, int taps) {
    const int nchunks = Cin / KC;
    const int64_t n16 = (int64_t)Cout * taps * Cin / 8;    // 16-byte units
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n16) return;
    const int lane = (int)(u & 63);
    int64_t r = u >> 6;
    const int mh = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % taps);
    r /= taps;
    const int chunk = (int)(r % nchunks);
    const int nb = (int)(r / nchunks);
    const int li = lane & 31, kh = lane >> 5;
    const uint16_t* src = w + ((int64_t)(nb * 32 + li) * taps + tap) * Cin + chunk * KC + mh * 16 + kh * 8;
    *reinterpret_cast<uint4*>(dst + u * 8) = *reinterpret_cast<const uint4*>(src);
}

int launch_pc16_weights(const void* w16, int Cout, int Cin, void* dst, hipStream_t s, int taps) {
    if ((Cin % KC) || (Cout % 32) || (taps != 1 && taps != 9) || (int64_t)Cout * taps * Cin * 2 >= (1LL << 31)) {
        set_error("pc16_weights: unsupported Cout=%d Cin=%d", Cout, Cin);
        return ERR_SHAPE;
    }
    const int64_t n16 = (int64_t)Cout * taps * Cin / 8;
    hipLaunchKernelGGL(pc16_weights_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s,
                       static_cast<const uint16_t*>(w16), Cout, Cin, static_cast<uint16_t*>(dst), taps);
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

// The producer / consumer form takes a 3x3 on 16-bit activations when its 16 x 16-pixel tiling applies and the launch has at
// least 64 (tile, 128-channel block) items: everything from 32 x 32 up at batch 8 (64 one-tile blocks there: 39 us against
// 43 us for the per-tap kernel's 128 blocks).
bool conv16_uses_pc(int B, int H, int W, int C1, int C2, int Cout, int taps) {
    if (taps != 9 || (H & 15) || (W & 15) || (C1 % KC) || (C2 % KC) || (Cout % 128)) return false;
    const int64_t cmax = C1 > C2 ? C1 : C2;
    if ((int64_t)(17 * W + 18) * cmax * 2 >= (1LL << 31) || (int64_t)Cout * 9 * (C1 + C2) * 2 >= (1LL << 31) ||
        (int64_t)H * W * Cout * 2 >= (1LL << 31)) return false;
    return ((int64_t)B * H * W / 256) * (Cout / 128) >= 64;
}

// FLOWSE_PC16_NARROW=0 / 1: never / always 64-channel blocks (A-B hook); default: below 3/4 of an item per CU
static int g_pc_narrow = getenv("FLOWSE_PC16_NARROW") ? atoi(getenv("FLOWSE_PC16_NARROW")) : -1;
void pc16_set_channel_blocks(int mode) { g_pc_narrow = mode < 0 ? -1 : (mode != 0); }
static bool pc16_narrow(int64_t items128, int cus) {
    if (g_pc_narrow >= 0) return g_pc_narrow != 0;
    return items128 * 4 < (int64_t)cus * 3;
}

int launch_pc16(const ConvArgs& a, hipStream_t s) {
    if (a.in_dt == DT_F32 || a.in_dt != a.out_dt || a.terms != 1 || a.partial || !a.wq || !a.wfrag ||
        (a.wq_f16 ? DT_F16 : DT_BF16) != a.in_dt || !conv16_uses_pc(a.B, a.H, a.W, a.C1, a.C2, a.Cout, a.taps)) {
        set_error("pc16: 16-bit storage in = out = operand type, fragment-order weights, 16 x 16-pixel tiles, no split-K form");
        return ERR_ARG;
    }
    if (a.sc1 || a.sc2 || a.SC1 || a.SC2 || a.wfrag_sc) {  // folded 1x1 shortcut
        const int64_t scn = (int64_t)a.SC1 + a.SC2;
        const char* w1 = static_cast<const char*>(a.wfrag);
        const char* w2 = static_cast<const char*>(a.wfrag_sc);
        const int64_t b1 = (int64_t)a.Cout * 9 * (a.C1 + a.C2) * 2, b2 = (int64_t)a.Cout * scn * 2;
        const int64_t span = w2 ? (w1 < w2 ? (w2 - w1) + b2 : (w1 - w2) + b1) : 0;
        if (!a.sc1 || !w2 || a.res || a.partial2 || a.SC1 <= 0 || (a.SC1 % KC) || (a.SC2 % KC) || (a.SC2 != 0) != (a.sc2 != nullptr) ||
            scn < 3 * KC || (int64_t)(17 * a.W + 18) * (a.SC1 > a.SC2 ? a.SC1 : a.SC2) * 2 >= (1LL << 31) ||
            span >= (1LL << 31) || (b1 > span ? b1 : span) >= (1LL << 31)) {
            set_error("pc16: folded shortcut needs sc1 (+ sc2) with 32-aligned channel counts (>= 96 in all), fragment-order 1x1 "
                      "weights within 2 GB of the 3x3 ones, and no residual");
            return ERR_ARG;
        }
    }
    int dev = 0, cus = 256;
    FLOWSE_HIP(hipGetDevice(&dev));
    static int cu_cache[64] = {0};
    if (!cu_cache[dev & 63]) {
        hipDeviceProp_t prop;
        FLOWSE_HIP(hipGetDeviceProperties(&prop, dev));
        cu_cache[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    cus = cu_cache[dev & 63];
    // 64-channel blocks (NJ = 1) when the 128-channel items would leave a quarter of the CUs or more idle (the 32 x 32 level
    // at batch 8: 64 items -> 256): half the MFMAs per A fragment read from LDS, but four times the CUs at work
    const int64_t items128 = ((int64_t)a.B * a.H * a.W / 256) * (a.Cout / 128);
    const bool narrow = pc16_narrow(items128, cus);
    const int64_t items = narrow ? 2 * items128 : items128;
    int grid = (int)(items < cus ? items : cus);
    grid &= ~7;                                            // a multiple of the 8 XCDs (>= 64 items: never 0)
    const int gn = a.gn.mean ? (a.gn_silu ? 2 : 1) : 0;
#define FLOWSE_LPC(GNF, F16)                                                                                 \
    {                                                                                                        \
        if (narrow) {                                                                                        \
            if (const int rc = allow_lds<&conv3x3_pc16_kernel<GNF, F16, 1>>(pc_lds(1))) return rc;           \
            hipLaunchKernelGGL((conv3x3_pc16_kernel<GNF, F16, 1>), dim3(grid), dim3(512), pc_lds(1), s, a);  \
        } else {                                                                                             \
            if (const int rc = allow_lds<&conv3x3_pc16_kernel<GNF, F16, 2>>(pc_lds(2))) return rc;           \
            hipLaunchKernelGGL((conv3x3_pc16_kernel<GNF, F16, 2>), dim3(grid), dim3(512), pc_lds(2), s, a);  \
        }                                                                                                    \
    }
    if (a.wq_f16) {
        if (gn == 2) FLOWSE_LPC(2, true) else if (gn == 1) FLOWSE_LPC(1, true) else FLOWSE_LPC(0, true)
    } else {
        if (gn == 2) FLOWSE_LPC(2, false) else if (gn == 1) FLOWSE_LPC(1, false) else FLOWSE_LPC(0, false)
    }
#undef FLOWSE_LPC
    FLOWSE_LAUNCH_CHECK();
    return OK;
}

}  // namespace flowse
