// pointnet2_ops.hip -- gather / group / ball_query / three_nn / three_interpolate (+ grads)
// for gfx950.  Replaces pointnet2/src/{sampling_gpu.cu:8-83, group_points_gpu.cu,
// ball_query_gpu.cu, interpolate_gpu.cu}.  All are HBM/L2-bound index or copy kernels:
// one index load is shared across channels, outputs are written coalesced, the per-query
// scans (ball_query, three_nn) stream the reference points through LDS tiles that every
// lane reads as a broadcast.
#include "common.h"

namespace {

constexpr int BLK = 256;

// K2  out[b,c,m] = points[b,c,idx[b,m]]           sampling_gpu.cu:8-24
__global__ __launch_bounds__(BLK) void gather_points_kernel(int c, int n, int m, const float *__restrict__ points,
                                                            const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= m) return;
    const int src = idx[(size_t)b * m + p];
    const float *pb = points + (size_t)b * c * n;
    float *ob = out + (size_t)b * c * m;
    for (int ci = 0; ci < c; ++ci) ob[(size_t)ci * m + p] = pb[(size_t)ci * n + src];
}

// K3  grad_points[b,c,idx[b,m]] += grad_out[b,c,m]  sampling_gpu.cu:46-63
__global__ __launch_bounds__(BLK) void gather_points_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx, float *__restrict__ grad_points) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= m) return;
    const int dst = idx[(size_t)b * m + p];
    const float *gb = grad_out + (size_t)b * c * m;
    float *pb = grad_points + (size_t)b * c * n;
    for (int ci = 0; ci < c; ++ci) atomicAdd(pb + (size_t)ci * n + dst, gb[(size_t)ci * m + p]);
}

// K5  out[b,c,s,k] = points[b,c,idx[b,s,k]]        group_points_gpu.cu:47-66
// blockIdx.y walks channel chunks of CCH so small-C calls still fill the chip.
constexpr int CCH = 8;
__global__ __launch_bounds__(BLK) void group_points_kernel(int c, int n, int sk, const float *__restrict__ points,
                                                           const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.z;
    const int e = blockIdx.x * BLK + threadIdx.x;
    if (e >= sk) return;
    const int c0 = blockIdx.y * CCH;
    const int c1 = min(c, c0 + CCH);
    const int src = idx[(size_t)b * sk + e];
    const float *pb = points + (size_t)b * c * n;
    float *ob = out + (size_t)b * c * sk;
    for (int ci = c0; ci < c1; ++ci) ob[(size_t)ci * sk + e] = pb[(size_t)ci * n + src];
}

// K6  group_points_gpu.cu:8-25
__global__ __launch_bounds__(BLK) void group_points_grad_kernel(int c, int n, int sk, const float *__restrict__ grad_out,
                                                                const int *__restrict__ idx, float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int e = blockIdx.x * BLK + threadIdx.x;
    if (e >= sk) return;
    const int c0 = blockIdx.y * CCH;
    const int c1 = min(c, c0 + CCH);
    const int dst = idx[(size_t)b * sk + e];
    const float *gb = grad_out + (size_t)b * c * sk;
    float *pb = grad_points + (size_t)b * c * n;
    for (int ci = c0; ci < c1; ++ci) atomicAdd(pb + (size_t)ci * n + dst, gb[(size_t)ci * sk + e]);
}

// Deterministic K3 / K6 / K9: the channel-major scatter-adds of the reference API as segmented reductions.  The
// caller sorts the T scatter positions of a batch element by destination (stable: equal destinations keep ascending position order)
// and passes the permutation `order` (B,T) and the CSR offsets seg (B,N+1).  One thread per (destination, channel) sums its segment
// in that order -- the order of a sequential loop over the positions, so the result equals the oracle's loop bit for bit, on every
// run; no atomics, no pre-zeroing.  WEIGHTED (K9): position t = 3 * p + k scatters grad_out[.., p] * weight[b, p, k].
template <bool WEIGHTED>
__global__ __launch_bounds__(BLK) void scatter_cm_sorted_kernel(int c, int n, int t, int src_n, const float *__restrict__ grad_out,
                                                                const int *__restrict__ order, const int *__restrict__ seg,
                                                                const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int d = blockIdx.x * BLK + threadIdx.x;   // destination point: consecutive threads write consecutive floats
    if (d >= n) return;
    const int c0 = blockIdx.y * CCH;
    const int c1 = min(c, c0 + CCH);
    const int *sg = seg + (size_t)b * (n + 1) + d;
    const int lo = sg[0], hi = sg[1];
    const int *ord = order + (size_t)b * t;
    const float *wb = WEIGHTED ? weight + (size_t)b * t : nullptr;
    for (int ci = c0; ci < c1; ++ci) {
        const float *src = grad_out + ((size_t)b * c + ci) * src_n;
        float acc = 0.f;
        for (int j = lo; j < hi; ++j) {
            const int pos = ord[j];
            acc += WEIGHTED ? src[pos / 3] * wb[pos] : src[pos];
        }
        grad_points[((size_t)b * c + ci) * n + d] = acc;
    }
}

// Channel-last row gather: out[b,t,:] = points[b, idx[b,t], :]   (index_points_group, mocopci.py:1204-1215)
template <typename VT, int VW>
__global__ __launch_bounds__(BLK) void group_rows_kernel(int n, int cv, long long total, int t, const VT *__restrict__ points,
                                                         const int *__restrict__ idx, VT *__restrict__ out) {
    // cv = C / VW vector columns per row; total = B * t * cv
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, cv, mcp_fits32(total));
        const int col = (int)(g - row * cv);
        const int b = (int)mcp_div(row, t, mcp_fits32(total));
        const int src = idx[row];
        out[g] = points[((long long)b * n + src) * cv + col];
    }
}

// First layer of the cost-volume cross() in its unfused form (pointconv_util.py:762-770): gather the K neighbour rows of
// every centre, add the centre's own row, LeakyReLU:  out[b,s,k,:] = leaky(points[b, idx[b,s,k], :] + centre[b,s,:]).
__global__ __launch_bounds__(BLK) void group_rows_add_leaky_kernel(int n, int cv, long long total, int sk, int k, float slope,
                                                                   const float4 *__restrict__ points, const int *__restrict__ idx,
                                                                   const float4 *__restrict__ centre, float4 *__restrict__ out) {
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, cv, mcp_fits32(total));  // b*S*K + s*K + j
        const int col = (int)(g - row * cv);
        const int b = (int)mcp_div(row, sk, mcp_fits32(total));
        const float4 p = points[((long long)b * n + idx[row]) * cv + col], c = centre[mcp_div(row, k, mcp_fits32(total)) * cv + col];
        float4 o = make_float4(p.x + c.x, p.y + c.y, p.z + c.z, p.w + c.w);
        o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
        o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
        out[g] = o;
    }
}

// Backward of the row gather: grad_points[b, idx[b,t], :] += grad_out[b,t,:]  (the channel-last counterpart of K6,
// group_points_grad_kernel, group_points_gpu.cu:49-75: same atomicAdd scatter, whole rows instead of strided scalars)
__global__ __launch_bounds__(BLK) void group_rows_grad_kernel(int n, int c, long long total, int t, const float *__restrict__ grad_out,
                                                              const int *__restrict__ idx, float *__restrict__ grad_points) {
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, c, mcp_fits32(total));
        const int col = (int)(g - row * c);
        const int b = (int)mcp_div(row, t, mcp_fits32(total));
        atomicAdd(grad_points + ((long long)b * n + idx[row]) * c + col, grad_out[g]);
    }
}

// Deterministic backward of the row gather ("deterministic segmented reduction instead of atomicAdd").  The
// caller sorts the T gather positions of every batch element by destination row (stable, so equal destinations keep their
// original order) and passes the permutation `order` (B,T) plus the CSR offsets seg (B,N+1).  One thread per (destination
// row, channel quad / channel) walks its segment in that fixed order: the same bits on every run, no atomics, no pre-zeroing
// (rows nobody gathered from are written as zeros).
template <bool VEC4>
__global__ __launch_bounds__(BLK) void group_rows_grad_sorted_kernel(int n, int c, int t, long long total, const float *__restrict__ grad_out,
                                                                     const int *__restrict__ order, const int *__restrict__ seg,
                                                                     float *__restrict__ grad_points) {
    const int cw = VEC4 ? c / 4 : c;
    long long g = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long stride = (long long)gridDim.x * BLK;
    for (; g < total; g += stride) {
        const long long row = mcp_div(g, cw, mcp_fits32(total));  // b * n + destination
        const int col = (int)(g - row * cw);
        const int b = (int)mcp_div(row, n, mcp_fits32(total)), d = (int)(row - (long long)b * n);
        const int *sg = seg + (long long)b * (n + 1) + d;
        const int lo = sg[0], hi = sg[1];
        const int *ord = order + (long long)b * t;
        const float *src = grad_out + (long long)b * t * c;
        if (VEC4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = lo; j < hi; ++j) {
                const float4 v = reinterpret_cast<const float4 *>(src + (long long)ord[j] * c)[col];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            reinterpret_cast<float4 *>(grad_points + row * c)[col] = acc;
        } else {
            float acc = 0.f;
            for (int j = lo; j < hi; ++j) acc += src[(long long)ord[j] * c + col];
            grad_points[row * c + col] = acc;
        }
    }
}

// Narrow rows (c <= 4: the coordinate gradients, c = 3): the kernel above gives each CHANNEL a thread that walks the whole segment
// alone -- three dependent 4-byte gathers per entry, one entry after the other (108 us per launch in the round-5 training trace, 31
// launches per step).  Here NARROW_SUB lanes share a destination row: lane l takes entries lo + l, lo + l + NARROW_SUB, ... with all c
// channels of an entry, and the lanes' partial sums are added by a butterfly (a + b is commutative bit for bit, so every lane ends
// with the same bits): a fixed order again, an eighth of the serial chain.
constexpr int NARROW_SUB = 8;
__global__ __launch_bounds__(BLK) void group_rows_grad_sorted_narrow_kernel(int n, int c, int t, long long rows, const float *__restrict__ grad_out,
                                                                            const int *__restrict__ order, const int *__restrict__ seg,
                                                                            float *__restrict__ grad_points) {
    const int l = threadIdx.x & (NARROW_SUB - 1);
    long long row = ((long long)blockIdx.x * BLK + threadIdx.x) / NARROW_SUB;   // b * n + destination
    const long long stride = (long long)gridDim.x * (BLK / NARROW_SUB);
    for (; row < rows; row += stride) {
        const int b = (int)mcp_div(row, n, mcp_fits32(rows)), d = (int)(row - (long long)b * n);
        const int *sg = seg + (long long)b * (n + 1) + d;
        const int lo = sg[0], hi = sg[1];
        const int *ord = order + (long long)b * t;
        const float *src = grad_out + (long long)b * t * c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = lo + l; j < hi; j += NARROW_SUB) {
            const float *e = src + (long long)ord[j] * c;
            a0 += e[0];
            if (c > 1) a1 += e[1];
            if (c > 2) a2 += e[2];
            if (c > 3) a3 += e[3];
        }
#pragma unroll
        for (int o = NARROW_SUB / 2; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o, NARROW_SUB);
            a1 += __shfl_xor(a1, o, NARROW_SUB);
            a2 += __shfl_xor(a2, o, NARROW_SUB);
            a3 += __shfl_xor(a3, o, NARROW_SUB);
        }
        if (l < c) grad_points[row * c + l] = l == 0 ? a0 : l == 1 ? a1 : l == 2 ? a2 : a3;
    }
}

// K4  ball_query_gpu.cu:9-45.  The reference gives one thread per centre a serial scan of all N points with an
// early break.  Here a WAVE owns a centre: lane l tests points 64*i + l (coalesced reads), the hit mask of each step
// is a ballot, and a hit's output slot is (hits so far) + (hits in lower lanes) -- exactly the reference's index
// order -- so the scan stops, wave-uniformly, as soon as nsample hits exist.  Slots beyond the hit count are filled
// with the first hit (the reference writes it to every slot when it is found); centres with no hit keep the
// caller's zeros.
constexpr int BQ_WAVES = 4;
__global__ __launch_bounds__(64 * BQ_WAVES) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                                   const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                                   int *__restrict__ idx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p = blockIdx.x * BQ_WAVES + wave;
    if (p >= m) return;  // whole wave; no barriers below
    const float *q = new_xyz + ((size_t)b * m + p) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float *rb = xyz + (size_t)b * n * 3;
    int *o = idx + ((size_t)b * m + p) * nsample;
    int cnt = 0, first = -1;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) hit = mcp_sqdist3(qx, qy, qz, rb[(size_t)k * 3 + 0], rb[(size_t)k * 3 + 1], rb[(size_t)k * 3 + 2]) < radius2;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (mask) {
            if (first < 0) first = base + (int)__builtin_ctzll(mask);
            const int rank = cnt + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (hit && rank < nsample) o[rank] = k;
            cnt += (int)__builtin_popcountll(mask);
        }
    }
    if (first >= 0)
        for (int l = min(cnt, nsample) + lane; l < nsample; l += 64) o[l] = first;
}

// QueryAndGroup.forward (pointnet2/pointnet2_utils.py:231-264) as ONE launch: the reference runs ball_query, transposes the cloud,
// groups it, subtracts the centres, groups the features and concatenates -- three kernels, two transposed copies, a cat.  Here a
// wave owns a centre: the ball query above leaves its nsample hits in an LDS row, then lane (channel offset, slot) gathers
// 64 / nsample output channels per pass -- relative coordinates first (xyz is read in its (B,N,3) layout: no transposed copy),
// then the (B,C,N) feature rows -- and writes the (B, 3+C, M, nsample) layout directly.  nsample <= 64.
__global__ __launch_bounds__(64 * BQ_WAVES) void query_and_group_kernel(int n, int m, int c, float radius2, int nsample, int xyz_ch,
                                                                        const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                        const float *__restrict__ features, float *__restrict__ out) {
    __shared__ int s_idx[BQ_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p = blockIdx.x * BQ_WAVES + wave;
    if (p >= m) return;  // whole wave; no workgroup barriers below
    const float *q = new_xyz + ((size_t)b * m + p) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float *rb = xyz + (size_t)b * n * 3;
    s_idx[wave][lane] = 0;  // a centre with no hit groups point 0 (the reference's pre-zeroed idx, pointnet2_utils.py:218)
    __builtin_amdgcn_wave_barrier();
    int cnt = 0, first = -1;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) hit = mcp_sqdist3(qx, qy, qz, rb[(size_t)k * 3 + 0], rb[(size_t)k * 3 + 1], rb[(size_t)k * 3 + 2]) < radius2;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (mask) {
            if (first < 0) first = base + (int)__builtin_ctzll(mask);
            const int rank = cnt + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (hit && rank < nsample) s_idx[wave][rank] = k;
            cnt += (int)__builtin_popcountll(mask);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (first >= 0 && lane >= min(cnt, nsample) && lane < nsample) s_idx[wave][lane] = first;
    __builtin_amdgcn_wave_barrier();
    const int cpl = 64 / nsample;                 // output channels per pass
    const int slot = lane % nsample, coff = lane / nsample;
    const int ct = xyz_ch + (features ? c : 0);   // output channels
    const int id = s_idx[wave][slot];
    const float centre[3] = {qx, qy, qz};
    if (coff < cpl) {
        for (int ch = coff; ch < ct; ch += cpl) {
            float v;
            if (ch < xyz_ch) {
                v = rb[(size_t)id * 3 + ch] - (ch == 0 ? centre[0] : ch == 1 ? centre[1] : centre[2]);
            } else {
                v = features[((size_t)b * c + (ch - xyz_ch)) * n + id];
            }
            out[(((size_t)b * ct + ch) * m + p) * nsample + slot] = v;
        }
    }
}

// K7  interpolate_gpu.cu:9-52.  Lane per unknown point; known points through an LDS tile.
constexpr int NN_TILE = 1024;
__global__ __launch_bounds__(BLK) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int *__restrict__ idx) {
    __shared__ float tile[NN_TILE * 3];
    const int b = blockIdx.y;
    const int p = blockIdx.x * BLK + threadIdx.x;
    const bool live = p < n;
    const float *u = unknown + ((size_t)b * n + (live ? p : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *kb = known + (size_t)b * m * 3;
    // the reference keeps doubles initialised to 1e40 that only ever hold floats: (float)1e40 = +inf,
    // and "d < 1e40" == "d < +inf" for every float d, so float accumulators are exact.
    float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
    int besti1 = 0, besti2 = 0, besti3 = 0;
    for (int base = 0; base < m; base += NN_TILE) {
        const int len = min(NN_TILE, m - base);
        __syncthreads();
        for (int i = threadIdx.x; i < len * 3; i += BLK) tile[i] = kb[(size_t)base * 3 + i];
        __syncthreads();
        for (int k = 0; k < len; ++k) {
            const float d = mcp_sqdist3(ux, uy, uz, tile[k * 3 + 0], tile[k * 3 + 1], tile[k * 3 + 2]);
            const int kk = base + k;
            if (d < best1) {
                best3 = best2; besti3 = besti2;
                best2 = best1; besti2 = besti1;
                best1 = d; besti1 = kk;
            } else if (d < best2) {
                best3 = best2; besti3 = besti2;
                best2 = d; besti2 = kk;
            } else if (d < best3) {
                best3 = d; besti3 = kk;
            }
        }
    }
    if (live) {
        float *od = dist2 + ((size_t)b * n + p) * 3;
        int *oi = idx + ((size_t)b * n + p) * 3;
        od[0] = best1; od[1] = best2; od[2] = best3;
        oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

// K8  interpolate_gpu.cu:77-97   out[b,c,p] = w0*P[i0] + w1*P[i1] + w2*P[i2]  (canon: two fmas)
__global__ __launch_bounds__(BLK) void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points,
                                                                const int *__restrict__ idx, const float *__restrict__ weight,
                                                                float *__restrict__ out) {
    const int b = blockIdx.z;
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    const int c0 = blockIdx.y * CCH;
    const int c1 = min(c, c0 + CCH);
    const int *id = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ci = c0; ci < c1; ++ci) {
        const float *pt = points + ((size_t)b * c + ci) * m;
        out[((size_t)b * c + ci) * n + p] = __builtin_fmaf(w2, pt[i2], __builtin_fmaf(w1, pt[i1], w0 * pt[i0]));
    }
}

// K9  interpolate_gpu.cu:120-142
__global__ __launch_bounds__(BLK) void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx, const float *__restrict__ weight,
                                                                     float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    const int c0 = blockIdx.y * CCH;
    const int c1 = min(c, c0 + CCH);
    const int *id = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ci = c0; ci < c1; ++ci) {
        const float g = grad_out[((size_t)b * c + ci) * n + p];
        float *gp = grad_points + ((size_t)b * c + ci) * m;
        atomicAdd(gp + i0, g * w0);
        atomicAdd(gp + i1, g * w1);
        atomicAdd(gp + i2, g * w2);
    }
}

}  // namespace

MCP_EXPORT int mcp_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out,
                                 mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && npoints > 0 && points && idx && out);
    hipLaunchKernelGGL(gather_points_kernel, dim3(mcp_divup(npoints, BLK), b), dim3(BLK), 0, (hipStream_t)stream, c, n, npoints,
                       points, idx, out);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx, float *grad_points,
                                      mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && npoints > 0 && grad_out && idx && grad_points);
    hipLaunchKernelGGL(gather_points_grad_kernel, dim3(mcp_divup(npoints, BLK), b), dim3(BLK), 0, (hipStream_t)stream, c, n,
                       npoints, grad_out, idx, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                                mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0 && points && idx && out);
    const int sk = npoints * nsample;
    hipLaunchKernelGGL(group_points_kernel, dim3(mcp_divup(sk, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0, (hipStream_t)stream, c, n,
                       sk, points, idx, out);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                     float *grad_points, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0 && grad_out && idx && grad_points);
    const int sk = npoints * nsample;
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(mcp_divup(sk, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0, (hipStream_t)stream,
                       c, n, sk, grad_out, idx, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_points_grad_sorted(int b, int c, int n, int t, const float *grad_out, const int *order, const int *seg,
                                            float *grad_points, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && t > 0 && grad_out && order && seg && grad_points);
    hipLaunchKernelGGL(scatter_cm_sorted_kernel<false>, dim3(mcp_divup(n, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0, (hipStream_t)stream, c, n, t, t,
                       grad_out, order, seg, nullptr, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_three_interpolate_grad_sorted(int b, int c, int n, int m, const float *grad_out, const int *order, const int *seg,
                                                 const float *weight, float *grad_points, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && n > 0 && m > 0 && grad_out && order && seg && weight && grad_points);
    hipLaunchKernelGGL(scatter_cm_sorted_kernel<true>, dim3(mcp_divup(m, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0, (hipStream_t)stream, c, m, 3 * n, n,
                       grad_out, order, seg, weight, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_rows(int b, int n, int c, int t, const float *points, const int *idx, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && c > 0 && t > 0 && points && idx && out);
    hipStream_t s = (hipStream_t)stream;
    mcp_prof_begin(MCP_KERNEL_GROUP_ROWS, s);
    const bool al16 = ((((uintptr_t)points) | ((uintptr_t)out)) & 15) == 0;
    if (c % 4 == 0 && al16) {
        const int cv = c / 4;
        const long long total = (long long)b * t * cv;
        const unsigned grid = (unsigned)min((long long)mcp_divup((unsigned)min(total, (long long)0x7fffffff), BLK), 8192LL);
        hipLaunchKernelGGL((group_rows_kernel<float4, 4>), dim3(grid), dim3(BLK), 0, s, n, cv, total, t,
                           reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out));
    } else {
        const long long total = (long long)b * t * c;
        const unsigned grid = (unsigned)min((long long)mcp_divup((unsigned)min(total, (long long)0x7fffffff), BLK), 8192LL);
        hipLaunchKernelGGL((group_rows_kernel<float, 1>), dim3(grid), dim3(BLK), 0, s, n, c, total, t, points, idx, out);
    }
    mcp_prof_end(MCP_KERNEL_GROUP_ROWS, s);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_rows_add_leaky(int b, int n, int c, int s, int k, float slope, const float *points, const int *idx,
                                        const float *centre, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && c > 0 && s > 0 && k > 0 && points && idx && centre && out);
    if (c % 4 != 0 || ((((uintptr_t)points) | ((uintptr_t)centre) | ((uintptr_t)out)) & 15)) return MCP_ERR_BAD_ARG;
    const int cv = c / 4;
    const long long total = (long long)b * s * k * cv;
    const unsigned grid = (unsigned)min((total + BLK - 1) / BLK, 16384LL);
    hipLaunchKernelGGL(group_rows_add_leaky_kernel, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, cv, total, s * k, k, slope,
                       reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<const float4 *>(centre),
                       reinterpret_cast<float4 *>(out));
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_rows_grad(int b, int n, int c, int t, const float *grad_out, const int *idx, float *grad_points,
                                   mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && c > 0 && t > 0 && grad_out && idx && grad_points);
    const long long total = (long long)b * t * c;
    const unsigned grid = (unsigned)min((long long)mcp_divup((unsigned)min(total, (long long)0x7fffffff), BLK), 8192LL);
    hipLaunchKernelGGL(group_rows_grad_kernel, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, c, total, t, grad_out, idx, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_group_rows_grad_sorted(int b, int n, int c, int t, const float *grad_out, const int *order, const int *seg,
                                          float *grad_points, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && c > 0 && t > 0 && grad_out && order && seg && grad_points);
    if (c <= 4) {
        const long long rows = (long long)b * n;
        const unsigned grid = (unsigned)min((rows * NARROW_SUB + BLK - 1) / BLK, 16384LL);
        hipLaunchKernelGGL(group_rows_grad_sorted_narrow_kernel, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, c, t, rows, grad_out, order, seg,
                           grad_points);
        return mcp_launch_status();
    }
    const bool vec4 = (c % 4 == 0) && !((((uintptr_t)grad_out) | ((uintptr_t)grad_points)) & 15);
    const long long total = (long long)b * n * (vec4 ? c / 4 : c);
    const unsigned grid = (unsigned)min((long long)mcp_divup((unsigned)min(total, (long long)0x7fffffff), BLK), 16384LL);
    if (vec4)
        hipLaunchKernelGGL(group_rows_grad_sorted_kernel<true>, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, c, t, total, grad_out, order,
                           seg, grad_points);
    else
        hipLaunchKernelGGL(group_rows_grad_sorted_kernel<false>, dim3(grid), dim3(BLK), 0, (hipStream_t)stream, n, c, t, total, grad_out, order,
                           seg, grad_points);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                              mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && m > 0 && nsample > 0 && new_xyz && xyz && idx);
    const float radius2 = radius * radius;  // ball_query_gpu.cu:20
    hipLaunchKernelGGL(ball_query_kernel, dim3(mcp_divup(m, BQ_WAVES), b), dim3(64 * BQ_WAVES), 0, (hipStream_t)stream, n, m, radius2,
                       nsample, new_xyz, xyz, idx);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_query_and_group(int b, int n, int m, int c, float radius, int nsample, int use_xyz, const float *xyz, const float *new_xyz,
                                   const float *features, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && m > 0 && nsample > 0 && xyz && new_xyz && out && (features ? c > 0 : use_xyz != 0));
    if (nsample > 64) return MCP_ERR_UNSUPPORTED;
    const float radius2 = radius * radius;  // ball_query_gpu.cu:20
    hipLaunchKernelGGL(query_and_group_kernel, dim3(mcp_divup(m, BQ_WAVES), b), dim3(64 * BQ_WAVES), 0, (hipStream_t)stream, n, m, c, radius2,
                       nsample, (use_xyz || !features) ? 3 : 0, xyz, new_xyz, features, out);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                            mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && m > 0 && unknown && known && dist2 && idx);
    hipLaunchKernelGGL(three_nn_kernel, dim3(mcp_divup(n, BLK), b), dim3(BLK), 0, (hipStream_t)stream, n, m, unknown, known, dist2,
                       idx);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                     float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && m > 0 && n > 0 && points && idx && weight && out);
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(mcp_divup(n, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0, (hipStream_t)stream, c,
                       m, n, points, idx, weight, out);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                                          float *grad_points, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && c > 0 && m > 0 && n > 0 && grad_out && idx && weight && grad_points);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(mcp_divup(n, BLK), mcp_divup(c, CCH), b), dim3(BLK), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return mcp_launch_status();
}
