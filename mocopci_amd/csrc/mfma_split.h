// mfma_split.h -- fp32 products on the bf16 matrix pipe (gfx950), shared by the fused MFMA layers.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value splits EXACTLY into three bf16 pieces,
//   x = x1 + x2 + x3,   x1 = top 16 bits of x,  x2 = top 16 bits of (x - x1),  x3 = x - x1 - x2
// (8 significant bits each; both subtractions are exact, x3 has at most 8 bits left so its top 16 bits are all of it), and
//   a.b = a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 + O(2^-24 |a.b|).
// Six v_mfma_f32_32x32x16_bf16 (exact 8x8-bit products, fp32 accumulation) per 16 k-values replace eight f32-input MFMAs:
// 6/16 of the matrix-pipe time, same accuracy class (tests compare against the CPU oracle at unchanged tolerances; the fusion
// kernel keeps its f32-MFMA build behind MCP_FUSION_F32_MFMA=1 for A/B runs: 4 ulp apart).
// Accumulator-as-operand chaining is unchanged by the split: k-step s of a 32-channel input tile takes registers 8s..8s+7
// of both lane halves, i.e. k = 8h + i  <->  channel chan_of(8s + i, h); the weight image is laid out to match.
#pragma once
#include "common.h"

#ifndef MCP_SPLIT_SCHED_MASK
#define MCP_SPLIT_SCHED_MASK 0x0000
#endif

typedef float mcp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mcp_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int mcp_chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ uint32_t mcp_top16(float x) { return __float_as_uint(x) & 0xFFFF0000u; }
// (hi16(odd) << 16) | hi16(even): two bf16 pieces in one dword, element 2j in the low half
__device__ __forceinline__ uint32_t mcp_pack_hi(float even, float odd) {
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}

struct McpSplit3 {  // the three bf16 pieces of one k-step's 8 values per lane (4 dwords each)
    uint4 p1, p2, p3;
};
__device__ __forceinline__ McpSplit3 mcp_split8(const float *v) {
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r1[i] = v[i] - __uint_as_float(mcp_top16(v[i]));      // exact: x and its top 16 bits share sign and exponent
        r2[i] = r1[i] - __uint_as_float(mcp_top16(r1[i]));    // exact; at most 8 significant bits remain
    }
    McpSplit3 o;
#ifdef MCP_SPLIT_DIAG_NOSPLIT  // timing-only diagnostic build (wrong results): no residual arithmetic, one packing pass
    o.p1 = make_uint4(mcp_pack_hi(v[0], v[1]), mcp_pack_hi(v[2], v[3]), mcp_pack_hi(v[4], v[5]), mcp_pack_hi(v[6], v[7]));
    o.p2 = o.p1; o.p3 = o.p1;
    return o;
#endif
    o.p1 = make_uint4(mcp_pack_hi(v[0], v[1]), mcp_pack_hi(v[2], v[3]), mcp_pack_hi(v[4], v[5]), mcp_pack_hi(v[6], v[7]));
    o.p2 = make_uint4(mcp_pack_hi(r1[0], r1[1]), mcp_pack_hi(r1[2], r1[3]), mcp_pack_hi(r1[4], r1[5]), mcp_pack_hi(r1[6], r1[7]));
    o.p3 = make_uint4(mcp_pack_hi(r2[0], r2[1]), mcp_pack_hi(r2[2], r2[3]), mcp_pack_hi(r2[4], r2[5]), mcp_pack_hi(r2[6], r2[7]));
    return o;
}
// registers 8s..8s+7 of an accumulator tile
__device__ __forceinline__ McpSplit3 mcp_split_kstep(const mcp_f32x16 &a, int s) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a[8 * s + i];
    return mcp_split8(v);
}
__device__ __forceinline__ mcp_f32x16 mcp_mfma_bf16(uint4 a, uint4 b, mcp_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mcp_bf16x8, a), __builtin_bit_cast(mcp_bf16x8, b), c, 0, 0, 0);
}
// acc += W(k-step) . X(k-step); w points at this lane's entry of piece 1, pieces are 64 uint4 apart; small terms first
__device__ __forceinline__ mcp_f32x16 mcp_mfma_split(const uint4 *w, const McpSplit3 &x, mcp_f32x16 acc) {
    const uint4 w1 = w[0], w2 = w[64], w3 = w[128];
    acc = mcp_mfma_bf16(w3, x.p1, acc);
    acc = mcp_mfma_bf16(w1, x.p3, acc);
    acc = mcp_mfma_bf16(w2, x.p2, acc);
    acc = mcp_mfma_bf16(w2, x.p1, acc);
    acc = mcp_mfma_bf16(w1, x.p2, acc);
    acc = mcp_mfma_bf16(w1, x.p1, acc);
    return acc;
}
// One output tile over KSTEPS k-steps: acc += sum_s W(s) . X(s).  ws points at this lane's entry of (k-step 0, piece 1); k-steps
// are 3 * 64 uint4 apart.  The weight pieces of step s+1 are read while the six MFMAs of step s run, and a scheduling barrier
// per step keeps the compiler from hoisting every LDS read of the (unrolled) layer to its top, which costs ~100 registers.
template <int KSTEPS>
__device__ __forceinline__ mcp_f32x16 mcp_tile_split(const uint4 *ws, const McpSplit3 *xs, mcp_f32x16 acc) {
    uint4 w1 = ws[0], w2 = ws[64], w3 = ws[128];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        uint4 n1 = w1, n2 = w2, n3 = w3;
#ifndef MCP_SPLIT_DIAG_NOLOAD  // timing-only diagnostic build (wrong results): every k-step reuses step 0's weights
        if (s + 1 < KSTEPS) {
            const uint4 *nx = ws + (size_t)(s + 1) * 3 * 64;
            n1 = nx[0]; n2 = nx[64]; n3 = nx[128];
        }
#endif
        acc = mcp_mfma_bf16(w3, xs[s].p1, acc);
        acc = mcp_mfma_bf16(w1, xs[s].p3, acc);
        acc = mcp_mfma_bf16(w2, xs[s].p2, acc);
        acc = mcp_mfma_bf16(w2, xs[s].p1, acc);
        acc = mcp_mfma_bf16(w1, xs[s].p2, acc);
        acc = mcp_mfma_bf16(w1, xs[s].p1, acc);
        __builtin_amdgcn_sched_barrier(MCP_SPLIT_SCHED_MASK);  // nothing moves across: a mask that let VALU/MFMA cross (0xE) produced WRONG results with this compiler (fusion test) and no speed-up
        w1 = n1; w2 = n2; w3 = n3;
    }
    return acc;
}

// Weight image of one (out x cin) layer, cin a multiple of 32: entry ((t * ksteps + s) * 3 + piece) * 64 + lane holds the 8 bf16
// W[32t + (lane&31)][32 (s>>1) + chan_of(8 (s&1) + i, lane>>5)], i = 0..7, ksteps = cin / 16.  Writes entries e, e+stride, ...
__device__ __forceinline__ void mcp_split_weights(uint4 *dst, const float *__restrict__ w, int cin, int out_tiles, int first, int stride) {
    const int ksteps = cin / 16;
    for (int e = first; e < out_tiles * ksteps * 64; e += stride) {
        const int lane = e & 63, s = (e >> 6) % ksteps, t = (e >> 6) / ksteps;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[(size_t)(32 * t + (lane & 31)) * cin + 32 * (s >> 1) + mcp_chan_of(8 * (s & 1) + i, lane >> 5)];
        const McpSplit3 sp = mcp_split8(v);
        uint4 *o = dst + (size_t)(t * ksteps + s) * 3 * 64 + lane;
        o[0] = sp.p1;
        o[64] = sp.p2;
        o[128] = sp.p3;
    }
}
