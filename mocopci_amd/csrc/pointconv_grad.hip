// pointconv_grad.hip -- backward of the fused PointConv grouping + WeightNet + aggregation (mcp_pointconv_agg; mocopci.py:1218-1266
// group / group_query, :1289-1300 WeightNet, :1330-1335 the matmul) for gfx950.  The reference differentiates the layer with autograd
// over the materialised (B,S,32,3+D) grouping and three (B,8,32,S) WeightNet activations.  Here a lane owns one (centre, neighbour)
// pair -- a wave two centres -- and everything of that pair stays in its registers:
//   * the WeightNet 3 -> 8 -> 8 -> 8 again, in the forward's fma order (same ReLU masks);
//   * with A = dL/dout of the centre ((3+D) x 8, staged once per wave in LDS and read as broadcasts):
//       d[dxyz | feat][k][c] = sum_m A[c][m] w[k][m]      -> grad_rows (B,S,32,D), and the coordinate part joins the WeightNet's
//       dw[k][m]             = sum_c [dxyz | feat][k][c] A[c][m]
//     one pass over the gathered feature row serves both (16 fma per channel);
//   * back through the WeightNet to dxyz: grad_gxyz (B,S,32,3) per pair, dL/dnew_xyz = -sum_k per centre;
//   * the 176 WeightNet weight gradients are sums over pairs: per-lane accumulators over the wave's pairs, one butterfly per wave at
//     the end, waves in wave order, workgroups in workgroup order by a second kernel -- fixed orders, bit-reproducible.
// grad_rows / grad_gxyz are scattered into dL/ds_points and dL/ds_xyz by the caller's deterministic segmented reduction.
#include "common.h"

namespace {

constexpr int K = 32, WN = 8, WAVES = 4;
// weight-gradient vector: dW0 (8,3) | db0 (8) | dW1 (8,8) | db1 (8) | dW2 (8,8) | db2 (8)
constexpr int G_W0 = 0, G_B0 = 24, G_W1 = 32, G_B1 = 96, G_W2 = 104, G_B2 = 168, G_FLOATS = 176;
constexpr int MAX_D = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float half_sum(float v) {  // over the 32 lanes that share lane >> 5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64 * WAVES, 1) void pointconv_agg_grad_kernel(long long total, int n, int s, int d, const float *__restrict__ s_xyz,
                                                                        const float *__restrict__ new_xyz, const float *__restrict__ s_points,
                                                                        const int *__restrict__ idx, const float *__restrict__ w0,
                                                                        const float *__restrict__ b0, const float *__restrict__ w1,
                                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                                        const float *__restrict__ b2, const float *__restrict__ gout,
                                                                        float *__restrict__ d_new_xyz, float *__restrict__ d_gxyz,
                                                                        float *__restrict__ d_rows, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // per wave: dL/dout of its two centres, 2 x (3 + d) x 8 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = lane >> 5, k = lane & 31;
    const int cin = d + 3, arow = cin * WN;  // floats per centre
    float *la = lds + (size_t)wave * 2 * arow;
    const float4 *mine = reinterpret_cast<const float4 *>(la + pl * arow);
    const bool f32 = mcp_fits32(total);
    float acc[G_FLOATS];
#pragma unroll
    for (int e = 0; e < G_FLOATS; ++e) acc[e] = 0.f;

    const long long pairs = (total + 1) >> 1;
    const McpUnits units = mcp_units_by_xcd(pairs, WAVES);   // XCD x takes the x-th eighth of the centre pairs (common.h)
    for (long long it = units.first + wave; it < units.limit; it += units.stride) {
        const long long p = 2 * it + pl;
        const bool valid = p < total;
        const long long pc = valid ? p : total - 1;  // a lane without a centre works on the last one with a zero gradient
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < 2 * (arow >> 2); e += 64) {
            const int half = e >= (arow >> 2), o = e - half * (arow >> 2);
            const long long pp = 2 * it + half;
            reinterpret_cast<float4 *>(la)[e] = pp < total ? reinterpret_cast<const float4 *>(gout + pp * arow)[o] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        const long long bb = mcp_div(pc, s, f32);
        const int id = idx[pc * K + k];
        const float *q = s_xyz + ((long long)bb * n + id) * 3;
        const float x0 = q[0] - new_xyz[pc * 3 + 0], x1 = q[1] - new_xyz[pc * 3 + 1], x2 = q[2] - new_xyz[pc * 3 + 2];
        // every lane walks its own gathered row: with one wave per SIMD (176 accumulators per lane) nothing hides a load, and two loads
        // in flight (the old unroll-by-2 loop) left 65 % of the wave's time in s_waitcnt (round 5 counters).  The row is read in chunks
        // of CHUNK float4: the first chunk's loads are issued here, in front of the WeightNet arithmetic, every later chunk's while the
        // one before it is used.
        constexpr int CHUNK = 8;
        const float4 *row = reinterpret_cast<const float4 *>(s_points + ((long long)bb * n + id) * d);
        const int quads = d >> 2;
        float4 cur[CHUNK], nxt[CHUNK];
#pragma unroll
        for (int i = 0; i < CHUNK; ++i) cur[i] = i < quads ? row[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- the WeightNet again (pointconv_agg_kernel's fma order) ----
        float h0[WN], h1[WN], h2[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j)
            h0[j] = fmaxf(__builtin_fmaf(w0[j * 3 + 2], x2, __builtin_fmaf(w0[j * 3 + 1], x1, __builtin_fmaf(w0[j * 3], x0, b0[j]))), 0.f);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float a = b1[j];
#pragma unroll
            for (int i = 0; i < WN; ++i) a = __builtin_fmaf(w1[j * WN + i], h0[i], a);
            h1[j] = fmaxf(a, 0.f);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float a = b2[j];
#pragma unroll
            for (int i = 0; i < WN; ++i) a = __builtin_fmaf(w2[j * WN + i], h1[i], a);
            h2[j] = fmaxf(a, 0.f);
        }
        // ---- one pass over [dxyz | gathered feature row]: the pair's input gradient and dL/dw ----
        float dw[WN];
#pragma unroll
        for (int m = 0; m < WN; ++m) dw[m] = 0.f;
        auto channel = [&](float v, int c) {  // returns d v, accumulates dw
            const float4 a = mine[2 * c], b = mine[2 * c + 1];
            float g = a.x * h2[0];
            g = __builtin_fmaf(a.y, h2[1], g); g = __builtin_fmaf(a.z, h2[2], g); g = __builtin_fmaf(a.w, h2[3], g);
            g = __builtin_fmaf(b.x, h2[4], g); g = __builtin_fmaf(b.y, h2[5], g); g = __builtin_fmaf(b.z, h2[6], g); g = __builtin_fmaf(b.w, h2[7], g);
            dw[0] = __builtin_fmaf(v, a.x, dw[0]); dw[1] = __builtin_fmaf(v, a.y, dw[1]); dw[2] = __builtin_fmaf(v, a.z, dw[2]);
            dw[3] = __builtin_fmaf(v, a.w, dw[3]); dw[4] = __builtin_fmaf(v, b.x, dw[4]); dw[5] = __builtin_fmaf(v, b.y, dw[5]);
            dw[6] = __builtin_fmaf(v, b.z, dw[6]); dw[7] = __builtin_fmaf(v, b.w, dw[7]);
            return g;
        };
        float dg0 = channel(x0, 0), dg1 = channel(x1, 1), dg2 = channel(x2, 2);
        {
            float4 *orow = reinterpret_cast<float4 *>(d_rows + (pc * K + k) * d);
            for (int c0 = 0; c0 < quads; c0 += CHUNK) {
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) nxt[i] = c0 + CHUNK + i < quads ? row[c0 + CHUNK + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) {
                    const int c4 = c0 + i;
                    if (c4 < quads) {   // wave-uniform
                        const float4 f = cur[i];
                        float4 o;
                        o.x = channel(f.x, 3 + 4 * c4 + 0);
                        o.y = channel(f.y, 3 + 4 * c4 + 1);
                        o.z = channel(f.z, 3 + 4 * c4 + 2);
                        o.w = channel(f.w, 3 + 4 * c4 + 3);
                        if (valid) orow[c4] = o;
                    }
                }
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) cur[i] = nxt[i];
            }
        }
        // ---- back through the WeightNet ----
        float dz2[WN], dz1[WN], dz0[WN];
#pragma unroll
        for (int m = 0; m < WN; ++m) {
            dz2[m] = h2[m] > 0.f ? dw[m] : 0.f;
            acc[G_B2 + m] += dz2[m];
#pragma unroll
            for (int i = 0; i < WN; ++i) acc[G_W2 + m * WN + i] = __builtin_fmaf(dz2[m], h1[i], acc[G_W2 + m * WN + i]);
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            float a = w2[i] * dz2[0];
#pragma unroll
            for (int m = 1; m < WN; ++m) a = __builtin_fmaf(w2[m * WN + i], dz2[m], a);
            dz1[i] = h1[i] > 0.f ? a : 0.f;
            acc[G_B1 + i] += dz1[i];
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[G_W1 + i * WN + j] = __builtin_fmaf(dz1[i], h0[j], acc[G_W1 + i * WN + j]);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float a = w1[j] * dz1[0];
#pragma unroll
            for (int i = 1; i < WN; ++i) a = __builtin_fmaf(w1[i * WN + j], dz1[i], a);
            dz0[j] = h0[j] > 0.f ? a : 0.f;
            acc[G_B0 + j] += dz0[j];
            acc[G_W0 + j * 3 + 0] = __builtin_fmaf(dz0[j], x0, acc[G_W0 + j * 3 + 0]);
            acc[G_W0 + j * 3 + 1] = __builtin_fmaf(dz0[j], x1, acc[G_W0 + j * 3 + 1]);
            acc[G_W0 + j * 3 + 2] = __builtin_fmaf(dz0[j], x2, acc[G_W0 + j * 3 + 2]);
            dg0 = __builtin_fmaf(w0[j * 3 + 0], dz0[j], dg0);
            dg1 = __builtin_fmaf(w0[j * 3 + 1], dz0[j], dg1);
            dg2 = __builtin_fmaf(w0[j * 3 + 2], dz0[j], dg2);
        }
        if (valid) {
            float *o = d_gxyz + (pc * K + k) * 3;
            o[0] = dg0; o[1] = dg1; o[2] = dg2;
        }
        const float sx = half_sum(dg0), sy = half_sum(dg1), sz = half_sum(dg2);
        if (valid && k == 0) {
            d_new_xyz[pc * 3 + 0] = -sx;
            d_new_xyz[pc * 3 + 1] = -sy;
            d_new_xyz[pc * 3 + 2] = -sz;
        }
    }
    // ---- weight gradients: butterfly per wave, waves in wave order, the workgroup's vector to the second pass ----
    __syncthreads();
    float *red = lds;  // [WAVES][176], over the staging rows (no longer needed)
#pragma unroll
    for (int e = 0; e < G_FLOATS; ++e) {
        const float v = wave_sum(acc[e]);
        if (lane == 0) red[wave * G_FLOATS + e] = v;
    }
    __syncthreads();
    if (tid < G_FLOATS) {
        float v = red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += red[w * G_FLOATS + tid];
        partial[(size_t)blockIdx.x * G_FLOATS + tid] = v;
    }
}

__global__ __launch_bounds__(256) void pointconv_grad_reduce_kernel(const float *__restrict__ partial, int parts, float *__restrict__ out) {
    const int e = threadIdx.x;
    if (e >= G_FLOATS) return;
    float v = 0.f;
    for (int g = 0; g < parts; ++g) v += partial[(size_t)g * G_FLOATS + e];
    out[e] = v;
}

unsigned grad_grid(long long total) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = (total + 2 * WAVES - 1) / (2 * WAVES), cap = cus;  // one resident workgroup per CU: 176 accumulators per lane, one wave per SIMD
    return (unsigned)(want < cap ? want : cap);
}

}  // namespace

MCP_EXPORT int mcp_pointconv_agg_grad_floats(void) { return G_FLOATS; }

MCP_EXPORT size_t mcp_pointconv_agg_grad_workspace_bytes(int b, int s) {
    if (b <= 0 || s <= 0) return 0;
    return (size_t)grad_grid((long long)b * s) * G_FLOATS * sizeof(float);
}

MCP_EXPORT int mcp_pointconv_agg_grad(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points, const int *idx,
                                      const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                                      const float *grad_out, float *grad_new_xyz, float *grad_gxyz, float *grad_rows, float *grad_weights,
                                      void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && s_xyz && new_xyz && s_points && idx && w0 && b0 && w1 && b1 && w2 && b2 && grad_out && grad_new_xyz &&
                   grad_gxyz && grad_rows && grad_weights && workspace);
    if (k != K || d < 4 || (d & 3) || d > MAX_D) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)s_points) | ((uintptr_t)grad_out) | ((uintptr_t)grad_rows)) & 15) return MCP_ERR_BAD_ARG;
    const long long total = (long long)b * s;
    const unsigned grid = grad_grid(total);
    if (workspace_bytes < (size_t)grid * G_FLOATS * sizeof(float)) return MCP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t lds = (size_t)WAVES * 2 * (d + 3) * WN * sizeof(float);
    if (lds < (size_t)WAVES * G_FLOATS * sizeof(float)) lds = (size_t)WAVES * G_FLOATS * sizeof(float);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pointconv_agg_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    mcp_prof_begin(MCP_KERNEL_POINTCONV, st);
    hipLaunchKernelGGL(pointconv_agg_grad_kernel, dim3(grid), dim3(64 * WAVES), lds, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2,
                       grad_out, grad_new_xyz, grad_gxyz, grad_rows, static_cast<float *>(workspace));
    hipLaunchKernelGGL(pointconv_grad_reduce_kernel, dim3(1), dim3(256), 0, st, static_cast<const float *>(workspace), (int)grid, grad_weights);
    mcp_prof_end(MCP_KERNEL_POINTCONV, st);
    return mcp_launch_status();
}
