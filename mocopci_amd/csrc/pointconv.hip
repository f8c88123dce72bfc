// pointconv.hip -- fused grouping + WeightNet + neighbour aggregation of PointConv / PointConvD for gfx950.
//
// Reference (mocopci.py:1218-1266 group/group_query, :1289-1300 WeightNet, :1330-1335 / :1381-1387):
//   idx -> two K5 gathers (+ permute copies) -> cat [dxyz | feats] (B,S,32,3+D) -> WeightNet 3->8->8->8
//   (three Conv2d+ReLU launches on (B,8,32,S)) -> batched matmul (3+D,32)x(32,8) per point -> (B,S,(3+D)*8).
// At level 0 the intermediates are 280 MiB per tensor.  Here one workgroup owns PPB points:
//   phase 1: one thread per (point, neighbour) gathers the neighbour coordinate, forms dxyz and runs
//            the 3->8->8->8 MLP in registers (weights arrive as scalar loads), leaving dxyz and the
//            8 kernel weights in LDS;
//   phase 2: one thread per (point, channel) walks the 32 neighbours: the gathered feature rows are
//            read with lanes on consecutive channels (coalesced row segments), the 8 weights are LDS
//            broadcasts, and the (3+D) x 8 aggregate is accumulated as an ascending-k fma chain and
//            written as two float4.
// The following Linear((3+D)*8 -> C_out) is a plain GEMM and stays on the BLAS library.
#include "common.h"

namespace {

constexpr int K = 32, WN = 8, PPB = 8;  // phase 1 uses PPB * K = 256 threads; phase 2 all THREADS of the workgroup

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pointconv_agg_kernel(long long total, int n, int s, int d, const float *__restrict__ s_xyz,
                                                                const float *__restrict__ new_xyz, const float *__restrict__ s_points,
                                                                const int *__restrict__ idx, const float *__restrict__ w0,
                                                                const float *__restrict__ b0, const float *__restrict__ w1,
                                                                const float *__restrict__ b1, const float *__restrict__ w2,
                                                                const float *__restrict__ b2, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float wl[PPB][K][WN];  // kernel weights per (point, neighbour)
    __shared__ float gx[PPB][K][3];                                // dxyz
    __shared__ int il[PPB][K];
    const int tid = threadIdx.x;
    const int cin = d + 3;
    const bool f32 = mcp_fits32(total);
    const bool off32 = (long long)n * d < (1LL << 31);
    for (long long p0 = (long long)blockIdx.x * PPB; p0 < total; p0 += (long long)gridDim.x * PPB) {
        __syncthreads();
        {   // ---- phase 1 ----
            const int pl = tid >> 5, k = tid & 31;
            const long long p = p0 + pl;
            if (tid < PPB * K && p < total) {
                const long long bb = mcp_div(p, s, f32);
                const int id = idx[p * K + k];
                const float *q = s_xyz + ((long long)bb * n + id) * 3;
                const float x0 = q[0] - new_xyz[p * 3 + 0], x1 = q[1] - new_xyz[p * 3 + 1], x2 = q[2] - new_xyz[p * 3 + 2];
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
#pragma unroll
                for (int j = 0; j < WN; ++j) wl[pl][k][j] = h2[j];
                gx[pl][k][0] = x0; gx[pl][k][1] = x1; gx[pl][k][2] = x2;
                il[pl][k] = id;
            }
        }
        __syncthreads();
        // ---- phase 2 ----
        // Feature channels and coordinate channels are separate loops: one loop with `c < 3 ? LDS : global` per neighbour made the
        // compiler merge the two sources into FLAT loads behind a pointer select -- a divergent branch per neighbour and waits
        // that drain the LDS and the global queue together.
        for (int it = tid; it < PPB * d; it += THREADS) {
            const int pl = it / d, c = it - pl * d;
            const long long p = p0 + pl;
            if (p >= total) break;
            const long long bb = mcp_div(p, s, f32);
            float acc[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[j] = 0.f;
            const float *fb = s_points + (long long)bb * n * d + c;
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                // row offsets within one batch element fit 32 bits whenever n * d does (checked by the host: off32)
                const float f = off32 ? fb[(unsigned)il[pl][k] * (unsigned)d] : fb[(long long)il[pl][k] * d];
                const float4 wa = *reinterpret_cast<const float4 *>(&wl[pl][k][0]);
                const float4 wb = *reinterpret_cast<const float4 *>(&wl[pl][k][4]);
                acc[0] = __builtin_fmaf(f, wa.x, acc[0]); acc[1] = __builtin_fmaf(f, wa.y, acc[1]);
                acc[2] = __builtin_fmaf(f, wa.z, acc[2]); acc[3] = __builtin_fmaf(f, wa.w, acc[3]);
                acc[4] = __builtin_fmaf(f, wb.x, acc[4]); acc[5] = __builtin_fmaf(f, wb.y, acc[5]);
                acc[6] = __builtin_fmaf(f, wb.z, acc[6]); acc[7] = __builtin_fmaf(f, wb.w, acc[7]);
            }
            float4 *o = reinterpret_cast<float4 *>(out + (p * cin + 3 + c) * WN);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        if (tid < PPB * 3) {  // the three coordinate channels of every point: dxyz from LDS
            const int pl = tid / 3, c = tid - pl * 3;
            const long long p = p0 + pl;
            if (p < total) {
                float acc[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[j] = 0.f;
#pragma unroll 8
                for (int k = 0; k < K; ++k) {
                    const float f = gx[pl][k][c];
                    const float4 wa = *reinterpret_cast<const float4 *>(&wl[pl][k][0]);
                    const float4 wb = *reinterpret_cast<const float4 *>(&wl[pl][k][4]);
                    acc[0] = __builtin_fmaf(f, wa.x, acc[0]); acc[1] = __builtin_fmaf(f, wa.y, acc[1]);
                    acc[2] = __builtin_fmaf(f, wa.z, acc[2]); acc[3] = __builtin_fmaf(f, wa.w, acc[3]);
                    acc[4] = __builtin_fmaf(f, wb.x, acc[4]); acc[5] = __builtin_fmaf(f, wb.y, acc[5]);
                    acc[6] = __builtin_fmaf(f, wb.z, acc[6]); acc[7] = __builtin_fmaf(f, wb.w, acc[7]);
                }
                float4 *o = reinterpret_cast<float4 *>(out + (p * cin + c) * WN);
                o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
    }
}

}  // namespace

MCP_EXPORT int mcp_pointconv_agg(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points,
                                 const int *idx, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                                 const float *b2, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && d > 0 && s_xyz && new_xyz && s_points && idx && w0 && b0 && w1 && b1 && w2 && b2 && out);
    if (k != K) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)out) & 15) return MCP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)b * s;
    // one workgroup per PPB points (no persistent loop in practice): the dispatcher balances around whatever else holds CUs, and
    // the two barriers per group overlap across resident workgroups (measured 1.10 -> 0.99 ms per step against a 4096-workgroup cap)
    const unsigned grid = (unsigned)min((total + PPB - 1) / PPB, 1LL << 20);
    mcp_prof_begin(MCP_KERNEL_POINTCONV, st);
    // wide layers of the small levels (few workgroups, PPB * d channel sums each): 1024 threads walk the (point, channel) items of a
    // group in 2-3 passes instead of 8-17 -- these launches are latency chains on an otherwise idle chip (49 -> 20 us at level 4, 31 -> 26 us at level 3; at d = 128 the wider group is slower: 29 -> 44 us)
    if (d >= 256 && total <= 8192)
        hipLaunchKernelGGL(pointconv_agg_kernel<1024>, dim3(grid), dim3(1024), 0, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1,
                           b1, w2, b2, out);
    else
        hipLaunchKernelGGL(pointconv_agg_kernel<256>, dim3(grid), dim3(256), 0, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1,
                           b1, w2, b2, out);
    mcp_prof_end(MCP_KERNEL_POINTCONV, st);
    return mcp_launch_status();
}
