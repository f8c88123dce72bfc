// mfma_grad.h -- pieces shared by the backward kernels of the fused MFMA layers (fusion_grad, fusion_bn, cross_grad, ptblock_grad).
//
// Their weight gradients contract over the NEIGHBOUR axis, which is the MFMA column (lane & 31) of every activation / gradient tile
// in the forward's layout.  An operand therefore passes once through a per-wave LDS "transposition tile": written in accumulator
// layout (row = channel, column = neighbour), read back with 8 consecutive neighbours per lane -- the A / B operand layout of
// v_mfma_f32_32x32x16_bf16 with k = neighbour -- and split three ways there.  Rows are MCP_TS floats apart: 16-byte aligned, and the
// 8-float reads of the 32 channel rows spread over the banks.
#pragma once
#include "mfma_split.h"

constexpr int MCP_TS = 36;

// Image of A[m][k] = w[k * m_total + m] (the transpose of a row-major (k_total, m_total) matrix) in mcp_split_weights' layout.
__device__ __forceinline__ void mcp_split_weights_transposed(uint4 *dst, const float *__restrict__ w, int m_total, int k_total, int first, int stride) {
    const int ksteps = k_total / 16, out_tiles = m_total / 32;
    for (int e = first; e < out_tiles * ksteps * 64; e += stride) {
        const int lane = e & 63, s = (e >> 6) % ksteps, t = (e >> 6) / ksteps;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[(size_t)(32 * (s >> 1) + mcp_chan_of(8 * (s & 1) + i, lane >> 5)) * m_total + 32 * t + (lane & 31)];
        const McpSplit3 sp = mcp_split8(v);
        uint4 *o = dst + (size_t)(t * ksteps + s) * 3 * 64 + lane;
        o[0] = sp.p1;
        o[64] = sp.p2;
        o[128] = sp.p3;
    }
}

// 8 consecutive neighbours of one channel row of a transposition tile
__device__ __forceinline__ void mcp_read8(const float *row, float *v) {
    const float4 a = reinterpret_cast<const float4 *>(row)[0], b = reinterpret_cast<const float4 *>(row)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float mcp_sum8(const float *v) { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }

// one accumulator-layout tile (32 channels x 32 neighbours) into rows 0..31 of a tile buffer: row = channel, column = neighbour
__device__ __forceinline__ void mcp_write_tile(float *tb, const mcp_f32x16 &v, int col, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[mcp_chan_of(r, h) * MCP_TS + col] = v[r];
}
// two tiles (64 channels)
__device__ __forceinline__ void mcp_write_tiles(float *tb, const mcp_f32x16 *v, int col, int h) {
    mcp_write_tile(tb, v[0], col, h);
    mcp_write_tile(tb + 32 * MCP_TS, v[1], col, h);
}

// acc += A . B with both operands split three ways (small terms first, as mcp_mfma_split)
__device__ __forceinline__ mcp_f32x16 mcp_mfma_split6(const McpSplit3 &a, const McpSplit3 &b, mcp_f32x16 acc) {
    acc = mcp_mfma_bf16(a.p3, b.p1, acc);
    acc = mcp_mfma_bf16(a.p1, b.p3, acc);
    acc = mcp_mfma_bf16(a.p2, b.p2, acc);
    acc = mcp_mfma_bf16(a.p2, b.p1, acc);
    acc = mcp_mfma_bf16(a.p1, b.p2, acc);
    acc = mcp_mfma_bf16(a.p1, b.p1, acc);
    return acc;
}
