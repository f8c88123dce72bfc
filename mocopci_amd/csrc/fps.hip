// fps.hip -- furthest point sampling for gfx950 (replaces pointnet2/src/sampling_gpu.cu:93-253).
//
// The reference runs one 1024-thread block per batch element that re-streams xyz and temp
// from global memory on each of the M-1 dependent iterations (20 B/point/iteration) and
// reduces through a 10-level LDS tree with 10 barriers.  Here one workgroup per batch
// element keeps its points AND their running min-distances in VGPRs for the whole call
// (N=8192: 8 points x 4 floats per lane), stages xyz once in LDS so the next centre is an
// LDS broadcast read, and reduces with DPP row ops + one LDS hop + ONE barrier per
// iteration (double-buffered slots).  HBM traffic is the compulsory 12N+4N+4M bytes.
//
// Index parity with the reference, including ties.  The reference's per-thread scan uses a
// strict '>' over k = tid, tid+bs, ... and its tree keeps the lower slot on equal values,
// so among equal distances the winner minimises (bitrev_L(k mod bs), k div bs) with
// bs = 2^L the reference block size (cuda_utils.h:10-14).  That is a total order on points,
// so any reduction shape gives the same answer: we reduce the pair
//   (ord(d2), ~sec(k)),  sec(k) = bitrev_L(k mod bs) << (32-L) | (k >> L)
// by max.  Distances use the canon of common.h (= oracle/pointset_oracle.c).
#include <math.h>

#include "common.h"

namespace {

__device__ __forceinline__ uint32_t fps_sec(uint32_t k, int L) {
    if (L == 0) return k;
    uint32_t vt = k & ((1u << L) - 1u);
    return (__brev(vt) & ~((1u << (32 - L)) - 1u)) | (k >> L);  // brev puts the L bits at the top
}
__device__ __forceinline__ uint32_t fps_unsec(uint32_t sec, int L) {
    if (L == 0) return sec;
    uint32_t vt = __brev(sec & ~((1u << (32 - L)) - 1u));
    return vt | ((sec & ((1u << (32 - L)) - 1u)) << L);
}

// T threads, P points per thread (k = tid + T*j).  FAST: T == reference block size, so a
// physical thread is exactly one reference thread and the in-thread scan is the reference's
// strict-'>' float scan.  Otherwise the in-thread scan also uses the (ord, ~sec) pair.
template <int T, int P, bool FAST, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_resident_kernel(int n, int m, int L, const float *__restrict__ xyz,
                                                         float *__restrict__ temp, int *__restrict__ idxs) {
    constexpr int W = T / 64;
    extern __shared__ float4 smem_f4[];
    uint2 *slots = reinterpret_cast<uint2 *>(smem_f4);        // [2][16]
    float *sxyz = reinterpret_cast<float *>(smem_f4) + 2 * 16 * 2;  // [n*3] when LDS_XYZ

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;

    float px[P], py[P], pz[P], pt[P];
    uint32_t nsec[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        int k = tid + T * j;
        bool ok = k < n;
        int kk = ok ? k : 0;
        px[j] = xyz[kk * 3 + 0];
        py[j] = xyz[kk * 3 + 1];
        pz[j] = xyz[kk * 3 + 2];
        pt[j] = ok ? temp[kk] : -INFINITY;  // never selected, never stored
        nsec[j] = ~fps_sec((uint32_t)kk, L);
    }
    if (LDS_XYZ) {
        for (int i = tid; i < n * 3; i += T) sxyz[i] = xyz[i];
    }
    if (tid < 32) slots[tid] = make_uint2(0u, 0u);
    if (tid == 0) idxs[0] = 0;
    __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if (LDS_XYZ) {
            x1 = sxyz[old * 3 + 0]; y1 = sxyz[old * 3 + 1]; z1 = sxyz[old * 3 + 2];
        } else {
            x1 = xyz[old * 3 + 0]; y1 = xyz[old * 3 + 1]; z1 = xyz[old * 3 + 2];
        }
        uint32_t hi, lo;
        if (FAST) {
            float best = -1.0f;
            uint32_t bsec = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float d = mcp_sqdist3(px[p], py[p], pz[p], x1, y1, z1);
                float d2 = fminf(d, pt[p]);
                pt[p] = d2;
                bool gt = d2 > best;
                bsec = gt ? nsec[p] : bsec;
                best = gt ? d2 : best;
            }
            hi = mcp_ord(best);
            lo = bsec;
        } else {
            hi = 0; lo = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float d = mcp_sqdist3(px[p], py[p], pz[p], x1, y1, z1);
                float d2 = fminf(d, pt[p]);
                pt[p] = d2;
                uint32_t h = mcp_ord(d2);
                bool gt = (h > hi) || (h == hi && nsec[p] > lo);
                lo = gt ? nsec[p] : lo;
                hi = gt ? h : hi;
            }
        }
        // wave reduce: max hi, then max lo among lanes holding that hi
        uint32_t whi = mcp_wave_max_u32(hi);
        uint32_t wlo = mcp_wave_max_u32(hi == whi ? lo : 0u);
        if (W > 1) {
            uint2 *sl = slots + (j & 1) * 16;
            if (lane == 0) sl[wave] = make_uint2(wlo, whi);
            __syncthreads();
            uint2 e = sl[lane & 15];
            uint32_t ghi = mcp_row_max_u32(e.y);
            uint32_t glo = mcp_row_max_u32(e.y == ghi ? e.x : 0u);
            whi = __builtin_amdgcn_readfirstlane((int)ghi);
            wlo = __builtin_amdgcn_readfirstlane((int)glo);
        }
        old = (int)fps_unsec(~wlo, L);
        if (tid == 0) idxs[j] = old;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int k = tid + T * p;
        if (k < n) temp[k] = pt[p];
    }
}

// Large-N fallback (N > 16 points per lane at 1024 threads): temp stays in global memory,
// xyz is re-read from L2 each iteration, same (ord, ~sec) reduction.  Correct for any N.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, int L, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ uint2 slots[2][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;
    if (tid < 32) (&slots[0][0])[tid] = make_uint2(0u, 0u);
    if (tid == 0) idxs[0] = 0;
    __syncthreads();
    int old = 0;
    for (int j = 1; j < m; ++j) {
        float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        uint32_t hi = 0, lo = 0;
        for (int k = tid; k < n; k += 1024) {
            float d = mcp_sqdist3(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
            float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            uint32_t h = mcp_ord(d2), s = ~fps_sec((uint32_t)k, L);
            bool gt = (h > hi) || (h == hi && s > lo);
            lo = gt ? s : lo;
            hi = gt ? h : hi;
        }
        uint32_t whi = mcp_wave_max_u32(hi);
        uint32_t wlo = mcp_wave_max_u32(hi == whi ? lo : 0u);
        uint2 *sl = slots[j & 1];
        if (lane == 0) sl[wave] = make_uint2(wlo, whi);
        __syncthreads();
        uint2 e = sl[lane & 15];
        uint32_t ghi = mcp_row_max_u32(e.y);
        uint32_t glo = mcp_row_max_u32(e.y == ghi ? e.x : 0u);
        wlo = __builtin_amdgcn_readfirstlane((int)glo);
        old = (int)fps_unsec(~wlo, L);
        if (tid == 0) idxs[j] = old;
    }
}

int ref_block_log2(int n) {
    // cuda_utils.h:10-14, same double arithmetic
    int pow_2 = (int)(log((double)n) / log(2.0));
    if (pow_2 > 10) pow_2 = 10;
    if (pow_2 < 0) pow_2 = 0;
    return pow_2;
}

template <int T, int P, bool FAST>
int launch_resident(int b, int n, int m, int L, const float *xyz, float *temp, int *idx, hipStream_t s) {
    const size_t slot_bytes = 2 * 16 * sizeof(uint2);
    const size_t xyz_bytes = (size_t)n * 3 * sizeof(float);
    if (xyz_bytes + slot_bytes <= 150 * 1024) {
        auto kern = fps_resident_kernel<T, P, FAST, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(b), dim3(T), slot_bytes + xyz_bytes, s, n, m, L, xyz, temp, idx);
    } else {
        hipLaunchKernelGGL((fps_resident_kernel<T, P, FAST, false>), dim3(b), dim3(T), slot_bytes, s, n, m, L, xyz, temp, idx);
    }
    return mcp_launch_status();
}

}  // namespace

MCP_EXPORT int mcp_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && temp && idx);
    if (m <= 0) return MCP_OK;  // sampling_gpu.cu:100
    hipStream_t s = (hipStream_t)stream;
    const int L = ref_block_log2(n);
    const int bs = 1 << L;
    int rc;
    mcp_prof_begin(MCP_KERNEL_FPS, s);
    if (bs >= 64) {
        const int P = (n + bs - 1) / bs;
        if (bs == 1024) {
            if (P <= 1) rc = launch_resident<1024, 1, true>(b, n, m, L, xyz, temp, idx, s);
            else if (P <= 2) rc = launch_resident<1024, 2, true>(b, n, m, L, xyz, temp, idx, s);
            else if (P <= 4) rc = launch_resident<1024, 4, true>(b, n, m, L, xyz, temp, idx, s);
            else if (P <= 8) rc = launch_resident<1024, 8, true>(b, n, m, L, xyz, temp, idx, s);
            else if (P <= 16) rc = launch_resident<1024, 16, true>(b, n, m, L, xyz, temp, idx, s);
            else {
                hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, s, n, m, L, xyz, temp, idx);
                rc = mcp_launch_status();
            }
        } else if (bs == 512) rc = launch_resident<512, 2, true>(b, n, m, L, xyz, temp, idx, s);
        else if (bs == 256) rc = launch_resident<256, 2, true>(b, n, m, L, xyz, temp, idx, s);
        else if (bs == 128) rc = launch_resident<128, 2, true>(b, n, m, L, xyz, temp, idx, s);
        else rc = launch_resident<64, 2, true>(b, n, m, L, xyz, temp, idx, s);
    } else {
        rc = launch_resident<64, 1, false>(b, n, m, L, xyz, temp, idx, s);  // n < 64
    }
    mcp_prof_end(MCP_KERNEL_FPS, s);
    return rc;
}
