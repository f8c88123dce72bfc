// emd.hip -- approximate Earth Mover's Distance (metric of test.py:90) for gfx950.
//
// Reference: models/EMD/cuda/emd_kernel.cu:29-162 (approxmatch: 10-level soft auction, level = -4^j for
// j = 7..-1 then 0) and :204-247 (matchcost).  The reference launches <<<32,512>>> with a grid-stride over
// the batch, i.e. ONE workgroup does all 30*n*m exponentials of a batch element, and it materialises
// match (B,m,n) -- 256 MiB per element at 8192 points, read-modify-written once per level.
// Here every pass of every level is a chip-wide launch (lane = one point of one set, the other set
// streamed through LDS tiles of (x,y,z,weight)), and because the cost  sum match*d^2  is linear in the
// per-level transfers w it is accumulated where w is produced: match is written only if the caller asks
// for it (the reference's approxmatch_forward API).  Per-lane summation order over the streamed set is the
// reference's (ascending index), so the iteration reproduces the reference up to __expf's rounding.
#include "common.h"

namespace {

constexpr int BLK = 256, TILE = 1024;

struct EmdWs {
    float *remainL, *remainR, *ratioL, *ratioR, *costk;
};
__device__ __forceinline__ EmdWs ws_of(float *ws, int b, int n, int m) {
    float *base = ws + (size_t)b * (3 * (size_t)n + 2 * (size_t)m);
    return EmdWs{base, base + n, base + n + m, base + 2 * (size_t)n + m, base + 2 * (size_t)n + 2 * (size_t)m};
}
__device__ __forceinline__ float d2(float x1, float y1, float z1, float x2, float y2, float z2) {
    return mcp_sqdist3(x2, y2, z2, x1, y1, z1);  // (x2-x1)^2 + (y2-y1)^2 + (z2-z1)^2, shared canon
}

__global__ __launch_bounds__(BLK) void emd_init_kernel(int n, int m, float multiL, float multiR, float *__restrict__ ws) {
    const EmdWs w = ws_of(ws, blockIdx.y, n, m);
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < n) { w.remainL[i] = multiL; w.costk[i] = 0.f; }
    if (i < m) w.remainR[i] = multiR;
}

// PASS 1: ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d) * remainR[l])             emd_kernel.cu:56-84
// PASS 3: w = exp(level*d)*ratioL[k]*ratioR[l]; match[l][k] += w; remainL[k] -= sum w    emd_kernel.cu:120-151
//         (+ cost_k += w*d, the fused matchcost)
template <int PASS>
__global__ __launch_bounds__(BLK) void emd_left_kernel(float level, int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2, float *__restrict__ ws,
                                                       float *__restrict__ match) {
    __shared__ float4 tile[TILE];
    const int b = blockIdx.y;
    const EmdWs w = ws_of(ws, b, n, m);
    const int k = blockIdx.x * BLK + threadIdx.x;
    const bool live = k < n;
    const float *p1 = xyz1 + ((size_t)b * n + (live ? k : 0)) * 3;
    const float x1 = p1[0], y1 = p1[1], z1 = p1[2];
    const float *p2 = xyz2 + (size_t)b * m * 3;
    const float *wsrc = PASS == 1 ? w.remainR : w.ratioR;
    const float rl = (PASS == 3 && live) ? w.ratioL[k] : 0.f;
    float suml = PASS == 1 ? 1e-9f : 0.f, cost = 0.f;
    float *mrow = match ? match + (size_t)b * n * m + k : nullptr;
    for (int l0 = 0; l0 < m; l0 += TILE) {
        const int lend = min(m, l0 + TILE) - l0;
        __syncthreads();
        for (int l = threadIdx.x; l < lend; l += BLK)
            tile[l] = make_float4(p2[(size_t)(l0 + l) * 3], p2[(size_t)(l0 + l) * 3 + 1], p2[(size_t)(l0 + l) * 3 + 2], wsrc[l0 + l]);
        __syncthreads();
        if (live) {
            for (int l = 0; l < lend; ++l) {
                const float4 t = tile[l];
                const float d = d2(x1, y1, z1, t.x, t.y, t.z);
                if (PASS == 1) {
                    suml += __expf(level * d) * t.w;
                } else {
                    const float ww = __expf(level * d) * rl * t.w;
                    if (mrow) mrow[(size_t)(l0 + l) * n] += ww;
                    suml += ww;
                    cost += ww * d;
                }
            }
        }
    }
    if (live) {
        if (PASS == 1) {
            w.ratioL[k] = w.remainL[k] / suml;
        } else {
            w.remainL[k] = fmaxf(0.0f, w.remainL[k] - suml);
            w.costk[k] += cost;
        }
    }
}

// PASS 2: sumr = remainR[l] * sum_k exp(level*d)*ratioL[k]; ratioR, remainR update           emd_kernel.cu:86-118
__global__ __launch_bounds__(BLK) void emd_right_kernel(float level, int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, float *__restrict__ ws) {
    __shared__ float4 tile[TILE];
    const int b = blockIdx.y;
    const EmdWs w = ws_of(ws, b, n, m);
    const int l = blockIdx.x * BLK + threadIdx.x;
    const bool live = l < m;
    const float *p2 = xyz2 + ((size_t)b * m + (live ? l : 0)) * 3;
    const float x2 = p2[0], y2 = p2[1], z2 = p2[2];
    const float *p1 = xyz1 + (size_t)b * n * 3;
    float sumr = 0.f;
    for (int k0 = 0; k0 < n; k0 += TILE) {
        const int kend = min(n, k0 + TILE) - k0;
        __syncthreads();
        for (int k = threadIdx.x; k < kend; k += BLK)
            tile[k] = make_float4(p1[(size_t)(k0 + k) * 3], p1[(size_t)(k0 + k) * 3 + 1], p1[(size_t)(k0 + k) * 3 + 2], w.ratioL[k0 + k]);
        __syncthreads();
        if (live) {
            for (int k = 0; k < kend; ++k) {
                const float4 t = tile[k];
                sumr += __expf(level * d2(t.x, t.y, t.z, x2, y2, z2)) * t.w;
            }
        }
    }
    if (live) {
        const float rr = w.remainR[l];
        sumr *= rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        w.ratioR[l] = consumption * rr;
        w.remainR[l] = fmaxf(0.0f, rr - sumr);
    }
}

// cost[b] = sum_k costk[b][k]   (fixed-shape tree: deterministic)
__global__ __launch_bounds__(BLK) void emd_reduce_kernel(int n, int m, const float *__restrict__ ws_c, float *__restrict__ cost) {
    __shared__ float part[BLK];
    float *ws = const_cast<float *>(ws_c);
    const EmdWs w = ws_of(ws, blockIdx.x, n, m);
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += BLK) s += w.costk[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int h = BLK / 2; h > 0; h >>= 1) {
        if (threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[blockIdx.x] = part[0];
}

}  // namespace

MCP_EXPORT int mcp_emd(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *cost, float *workspace,
                       mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && m > 0 && xyz1 && xyz2 && cost && workspace);
    hipStream_t s = (hipStream_t)stream;
    float multiL, multiR;  // emd_kernel.cu:32-38 (integer division)
    if (n >= m) { multiL = 1.f; multiR = (float)(n / m); } else { multiL = (float)(m / n); multiR = 1.f; }
    if (match) {
        hipError_t e = hipMemsetAsync(match, 0, sizeof(float) * (size_t)b * n * m, s);
        if (e != hipSuccess) return (int)e;
    }
    const int big = n > m ? n : m;
    hipLaunchKernelGGL(emd_init_kernel, dim3(mcp_divup(big, BLK), b), dim3(BLK), 0, s, n, m, multiL, multiR, workspace);
    for (int j = 7; j >= -2; --j) {
        float level = -powf(4.0f, (float)j);
        if (j == -2) level = 0.f;
        hipLaunchKernelGGL(emd_left_kernel<1>, dim3(mcp_divup(n, BLK), b), dim3(BLK), 0, s, level, n, m, xyz1, xyz2, workspace,
                           (float *)nullptr);
        hipLaunchKernelGGL(emd_right_kernel, dim3(mcp_divup(m, BLK), b), dim3(BLK), 0, s, level, n, m, xyz1, xyz2, workspace);
        hipLaunchKernelGGL(emd_left_kernel<3>, dim3(mcp_divup(n, BLK), b), dim3(BLK), 0, s, level, n, m, xyz1, xyz2, workspace, match);
    }
    hipLaunchKernelGGL(emd_reduce_kernel, dim3(b), dim3(BLK), 0, s, n, m, workspace, cost);
    return mcp_launch_status();
}
