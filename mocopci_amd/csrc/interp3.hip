// interp3.hip -- fused 3-NN inverse-distance interpolation (UpsampleFlow, mocopci.py:1485-1502;
// the interpolation half of PointWarping, :1472-1479) on channel-last tensors.
// The reference runs knn_point(3) (materialised (B,N,S) matrix + topk), two K5 gathers with their
// permute copies, a norm, and a weighted sum: 10+ launches per call, 59 calls per forward.
// Here: the K<=4 path of knn.hip, one tiny weight kernel, one vectorised row-blend kernel; the
// (idx3, w3) pair is returned so repeated calls on the same (dense, sparse) pair skip the search.
#include "common.h"

namespace {
constexpr int BLK = 256;

// weights from the DISTANCE of the gathered differences (not the squared expansion distance):
//   dist = clamp(||sparse[idx]-dense||, 1e-10); w = (1/dist) / sum(1/dist)      mocopci.py:1495-1498
__global__ __launch_bounds__(BLK) void interp3_weights_kernel(int n, int s, const float *__restrict__ dense,
                                                              const float *__restrict__ sparse, const int *__restrict__ idx3,
                                                              float *__restrict__ w3) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    const float *x = dense + ((size_t)b * n + p) * 3;
    const int *id = idx3 + ((size_t)b * n + p) * 3;
    const float x0 = x[0], x1 = x[1], x2 = x[2];
    float inv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *y = sparse + ((size_t)b * s + id[j]) * 3;
        const float dx = y[0] - x0, dy = y[1] - x1, dz = y[2] - x2;
        float nr = sqrtf((dx * dx + dy * dy) + dz * dz);
        nr = nr < 1e-10f ? 1e-10f : nr;
        inv[j] = 1.0f / nr;
    }
    const float nrm = (inv[0] + inv[1]) + inv[2];
    float *w = w3 + ((size_t)b * n + p) * 3;
    w[0] = inv[0] / nrm;
    w[1] = inv[1] / nrm;
    w[2] = inv[2] / nrm;
}

// out[b,p,:] = (w0*f[i0,:] + w1*f[i1,:]) + w2*f[i2,:]   (rounded products, sequential sum: torch.sum over dim 2)
template <typename VT, int VW>
__global__ __launch_bounds__(BLK) void interp3_apply_kernel(int n, int s, int cv, long long total, const VT *__restrict__ feat,
                                                            const int *__restrict__ idx3, const float *__restrict__ w3,
                                                            VT *__restrict__ out) {
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, cv, mcp_fits32(total));  // b*n + p
        const int col = (int)(g - row * cv);
        const int b = (int)mcp_div(row, n, mcp_fits32(total));
        const int *id = idx3 + row * 3;
        const float *w = w3 + row * 3;
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        const VT f0 = feat[((long long)b * s + id[0]) * cv + col];
        const VT f1 = feat[((long long)b * s + id[1]) * cv + col];
        const VT f2 = feat[((long long)b * s + id[2]) * cv + col];
        VT o;
        const float *a0 = reinterpret_cast<const float *>(&f0);
        const float *a1 = reinterpret_cast<const float *>(&f1);
        const float *a2 = reinterpret_cast<const float *>(&f2);
        float *oo = reinterpret_cast<float *>(&o);
#pragma unroll
        for (int e = 0; e < VW; ++e) oo[e] = (w0 * a0[e] + w1 * a1[e]) + w2 * a2[e];
        out[g] = o;
    }
}
// Backward of the blend w.r.t. the sparse features: grad_feat[b, idx3[b,p,j], :] += w3[b,p,j] * grad_out[b,p,:]
// (channel-last counterpart of K9, three_interpolate_grad_kernel, interpolate_gpu.cu:126-150)
__global__ __launch_bounds__(BLK) void interp3_apply_grad_kernel(int n, int s, int c, long long total, const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx3, const float *__restrict__ w3,
                                                                 float *__restrict__ grad_feat) {
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, c, mcp_fits32(total));  // b*n + p
        const int col = (int)(g - row * c);
        const int b = (int)mcp_div(row, n, mcp_fits32(total));
        const float go = grad_out[g];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            atomicAdd(grad_feat + ((long long)b * s + idx3[row * 3 + j]) * c + col, w3[row * 3 + j] * go);
    }
}
// Deterministic backward of the blend (autograd over mocopci.py:1480-1481 / :1500-1501; the reference's own K9,
// interpolate_gpu.cu:120-161, scatters with atomicAdd).  Two kernels, no (B,N,3,C) tensor in between:
//   grad_w3[b,p,j]  = <grad_out[b,p,:], feat[b, idx3[b,p,j], :]>              SUB lanes per (p, j), butterfly sum (fixed order)
//   grad_feat[b,q,:] = sum over the (p, j) with idx3[b,p,j] = q, in ascending position 3p + j, of w3[b,p,j] grad_out[b,p,:]
//                      from the CSR form of idx3 (mcp_scatter_segments): one thread per (row, channel quad) walks its segment
constexpr int I3_SUB = 8;
__global__ __launch_bounds__(BLK) void interp3_grad_w_kernel(int n, int s, int c, long long triples, const float *__restrict__ grad_out,
                                                             const float *__restrict__ feat, const int *__restrict__ idx3,
                                                             float *__restrict__ grad_w3, int vec) {
    const int l = threadIdx.x & (I3_SUB - 1);
    long long t = ((long long)blockIdx.x * BLK + threadIdx.x) / I3_SUB;   // (b*n + p) * 3 + j
    const long long stride = (long long)gridDim.x * (BLK / I3_SUB);
    for (; t < triples; t += stride) {
        const long long row = mcp_div(t, 3, mcp_fits32(triples));
        const int b = (int)mcp_div(row, n, mcp_fits32(triples));
        const float *g = grad_out + row * c, *f = feat + ((long long)b * s + idx3[t]) * c;
        float acc = 0.f;
        if (vec) {
            for (int q = l; q < (c >> 2); q += I3_SUB) {
                const float4 a = reinterpret_cast<const float4 *>(g)[q], v = reinterpret_cast<const float4 *>(f)[q];
                acc = __builtin_fmaf(a.w, v.w, __builtin_fmaf(a.z, v.z, __builtin_fmaf(a.y, v.y, __builtin_fmaf(a.x, v.x, acc))));
            }
        } else {
            for (int q = l; q < c; q += I3_SUB) acc = __builtin_fmaf(g[q], f[q], acc);
        }
#pragma unroll
        for (int o = I3_SUB / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, I3_SUB);
        if (l == 0) grad_w3[t] = acc;
    }
}

template <bool VEC4>
__global__ __launch_bounds__(BLK) void interp3_grad_feat_sorted_kernel(int n, int s, int c, long long total, const float *__restrict__ grad_out,
                                                                       const float *__restrict__ w3, const int *__restrict__ order,
                                                                       const int *__restrict__ seg, float *__restrict__ grad_feat) {
    const int cw = VEC4 ? c / 4 : c;
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, cw, mcp_fits32(total));  // b * s + destination
        const int col = (int)(g - row * cw);
        const int b = (int)mcp_div(row, s, mcp_fits32(total)), d = (int)(row - (long long)b * s);
        const int *sg = seg + (long long)b * (s + 1) + d;
        const int lo = sg[0], hi = sg[1];
        const int *ord = order + (long long)b * 3 * n;
        const float *wb = w3 + (long long)b * 3 * n;
        const float *src = grad_out + (long long)b * n * c;
        if (VEC4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = lo; j < hi; ++j) {
                const int pos = ord[j];
                const float w = wb[pos];
                const float4 v = reinterpret_cast<const float4 *>(src + (long long)(pos / 3) * c)[col];
                acc.x = __builtin_fmaf(w, v.x, acc.x); acc.y = __builtin_fmaf(w, v.y, acc.y);
                acc.z = __builtin_fmaf(w, v.z, acc.z); acc.w = __builtin_fmaf(w, v.w, acc.w);
            }
            reinterpret_cast<float4 *>(grad_feat + row * c)[col] = acc;
        } else {
            float acc = 0.f;
            for (int j = lo; j < hi; ++j) {
                const int pos = ord[j];
                acc = __builtin_fmaf(wb[pos], src[(long long)(pos / 3) * c + col], acc);
            }
            grad_feat[row * c + col] = acc;
        }
    }
}
}  // namespace

MCP_EXPORT int mcp_interp3_apply_grad_sorted(int b, int n, int s, int c, const float *feat, const int *idx3, const float *w3,
                                             const float *grad_out, const int *order, const int *seg, float *grad_feat, float *grad_w3,
                                             mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && c > 0 && idx3 && w3 && grad_out && (grad_feat || grad_w3));
    MCP_CHECK_ARGS(!grad_feat || (order && seg));
    MCP_CHECK_ARGS(!grad_w3 || feat);
    hipStream_t st = (hipStream_t)stream;
    if (grad_w3) {
        const long long triples = (long long)b * n * 3;
        const int vec = (c % 4 == 0) && !((((uintptr_t)grad_out) | ((uintptr_t)feat)) & 15);
        const unsigned grid = (unsigned)min((triples * I3_SUB + BLK - 1) / BLK, 16384LL);
        hipLaunchKernelGGL(interp3_grad_w_kernel, dim3(grid), dim3(BLK), 0, st, n, s, c, triples, grad_out, feat, idx3, grad_w3, vec);
        const int rc = mcp_launch_status();
        if (rc != MCP_OK) return rc;
    }
    if (grad_feat) {
        const bool vec4 = (c % 4 == 0) && !((((uintptr_t)grad_out) | ((uintptr_t)grad_feat)) & 15);
        const long long total = (long long)b * s * (vec4 ? c / 4 : c);
        const unsigned grid = (unsigned)min((total + BLK - 1) / BLK, 16384LL);
        if (vec4)
            hipLaunchKernelGGL(interp3_grad_feat_sorted_kernel<true>, dim3(grid), dim3(BLK), 0, st, n, s, c, total, grad_out, w3, order, seg, grad_feat);
        else
            hipLaunchKernelGGL(interp3_grad_feat_sorted_kernel<false>, dim3(grid), dim3(BLK), 0, st, n, s, c, total, grad_out, w3, order, seg, grad_feat);
    }
    return mcp_launch_status();
}

MCP_EXPORT int mcp_interp3_apply_grad(int b, int n, int s, int c, const float *grad_out, const int *idx3, const float *w3,
                                      float *grad_feat, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && c > 0 && grad_out && idx3 && w3 && grad_feat);
    const long long total = (long long)b * n * c;
    const unsigned grid = (unsigned)min((total + BLK - 1) / BLK, 8192LL);
    hipLaunchKernelGGL(interp3_apply_grad_kernel, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, s, c, total, grad_out, idx3, w3,
                       grad_feat);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_interp3_apply(int b, int n, int s, int c, const float *feat, const int *idx3, const float *w3, float *out,
                                 mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && c > 0 && feat && idx3 && w3 && out);
    hipStream_t st = (hipStream_t)stream;
    const bool al16 = ((((uintptr_t)feat) | ((uintptr_t)out)) & 15) == 0;
    if (c % 4 == 0 && al16) {
        const int cv = c / 4;
        const long long total = (long long)b * n * cv;
        const unsigned grid = (unsigned)min((total + BLK - 1) / BLK, 8192LL);
        hipLaunchKernelGGL((interp3_apply_kernel<float4, 4>), dim3(grid), dim3(BLK), 0, st, n, s, cv, total,
                           reinterpret_cast<const float4 *>(feat), idx3, w3, reinterpret_cast<float4 *>(out));
    } else {
        const long long total = (long long)b * n * c;
        const unsigned grid = (unsigned)min((total + BLK - 1) / BLK, 8192LL);
        hipLaunchKernelGGL((interp3_apply_kernel<float, 1>), dim3(grid), dim3(BLK), 0, st, n, s, c, total, feat, idx3, w3, out);
    }
    return mcp_launch_status();
}

MCP_EXPORT int mcp_interp3_weights(int b, int n, int s, const float *dense, const float *sparse, const int *idx3, float *w3,
                                   mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && dense && sparse && idx3 && w3);
    hipLaunchKernelGGL(interp3_weights_kernel, dim3(mcp_divup(n, BLK), b), dim3(BLK), 0, (hipStream_t)stream, n, s, dense, sparse, idx3,
                       w3);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_interp3(int b, int n, int s, int c, const float *dense, const float *sparse, const float *feat, float *out,
                           int *idx3, float *w3, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && c > 0 && dense && sparse && feat && out && idx3 && w3);
    hipStream_t st = (hipStream_t)stream;
    mcp_prof_begin(MCP_KERNEL_INTERP3, st);
    int rc = mcp_knn(b, n, s, 3, MCP_DIST_EXPANSION, dense, sparse, idx3, nullptr, stream);
    if (rc == MCP_OK) rc = mcp_interp3_weights(b, n, s, dense, sparse, idx3, w3, stream);
    if (rc == MCP_OK) rc = mcp_interp3_apply(b, n, s, c, feat, idx3, w3, out, stream);
    mcp_prof_end(MCP_KERNEL_INTERP3, st);
    return rc;
}
