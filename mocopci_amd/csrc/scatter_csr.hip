// scatter_csr.hip -- CSR form of a gather list for the deterministic scatter kernels (gfx950; training).
//
// The backward of every gather (group_points_gpu.cu:8-44, sampling_gpu.cu:46-83, interpolate_gpu.cu:120-161: atomicAdd in the
// reference) runs here as a segmented reduction: gather positions sorted by destination row, each row's addends summed in
// ascending position order (mcp_group_rows_grad_sorted & co).  Round 4 built that order with a stable key-value sort of the whole
// list (torch.sort + searchsorted: ~10 launches, 6.5 ms per training step in merge-sort kernels for lists of up to 24 x 524288
// positions whose keys are < 8192).  The keys ARE small, so this is a counting sort:
//   count   CHUNKS workgroups per batch element histogram their share of the positions in LDS (LDS atomics: a destination that
//           thousands of positions point at -- the refined cloud of an untrained network -- costs nothing extra) and store the
//           histograms;
//   scan    exclusive prefix sum over the destinations of one batch element -> seg, and for every (chunk, destination) the first
//           slot of that chunk's share of the row (chunks in order);
//   fill    a workgroup hands out the slots of its chunk from cursors in LDS (atomic with return: the order inside a chunk's share
//           of a row is whatever the hardware served) ...
//   rank    ... and one wave per destination puts its row in ascending position order: all-pairs ranks for short rows, a 64-lane
//           bitonic network (1 or 4 registers per lane) up to 256 entries; longer rows are queued and sorted share by share
//           (csr_long_kernel).
// The result is the stable sort's, bit for bit, whatever the atomics did.
#include "common.h"

namespace {

constexpr int CHUNKS = 16, CT = 1024;   // workgroups per batch element in count / fill, threads in each

// chunk g of batch element b: positions [g * per, (g + 1) * per); hist (B, CHUNKS, n)
__global__ __launch_bounds__(CT) void csr_count_kernel(int t, int n, int per, const int *__restrict__ idx, int *__restrict__ hist) {
    extern __shared__ int lds_hist[];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int *ib = idx + (size_t)b * t;
    for (int d = tid; d < n; d += CT) lds_hist[d] = 0;
    __syncthreads();
    const int p1 = min(t, (g + 1) * per);
    for (int p = g * per + tid; p < p1; p += CT) {
        const int d = ib[p];
        if ((unsigned)d < (unsigned)n) atomicAdd(&lds_hist[d], 1);
    }
    __syncthreads();
    int *hb = hist + ((size_t)b * CHUNKS + g) * n;
    for (int d = tid; d < n; d += CT) hb[d] = lds_hist[d];
}

// one workgroup per batch element: seg[d] = number of positions with a destination below d; hist[g][d] becomes the first slot of
// chunk g's share of row d (chunks in order)
__global__ __launch_bounds__(1024) void csr_scan_kernel(int n, int *__restrict__ hist, int *__restrict__ seg) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *hb = hist + (size_t)b * CHUNKS * n, *sb = seg + (size_t)b * (n + 1);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int d = base + tid;
        int cnt[CHUNKS];
        int v = 0;
#pragma unroll
        for (int g = 0; g < CHUNKS; ++g) {
            cnt[g] = d < n ? hb[(size_t)g * n + d] : 0;
            v += cnt[g];
        }
        int x = v;   // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int excl = before + x - v;
        if (d < n) {
            sb[d] = excl;
            int at = excl;
#pragma unroll
            for (int g = 0; g < CHUNKS; ++g) {
                hb[(size_t)g * n + d] = at;
                at += cnt[g];
            }
        }
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (tid == 0) sb[n] = carry_s;
}

__global__ __launch_bounds__(CT) void csr_fill_kernel(int t, int n, int per, const int *__restrict__ idx, const int *__restrict__ hist, int *__restrict__ slots) {
    extern __shared__ int lds_cur[];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int *ib = idx + (size_t)b * t;
    const int *hb = hist + ((size_t)b * CHUNKS + g) * n;
    int *sl = slots + (size_t)b * t;
    for (int d = tid; d < n; d += CT) lds_cur[d] = hb[d];
    __syncthreads();
    const int p1 = min(t, (g + 1) * per);
    for (int p = g * per + tid; p < p1; p += CT) {
        const int d = ib[p];
        if ((unsigned)d < (unsigned)n) sl[atomicAdd(&lds_cur[d], 1)] = p;
    }
}

// ascending bitonic network over NR registers x 64 lanes (element e = r * 64 + lane); padding (INT_MAX) sorts last
template <int NR>
__device__ __forceinline__ void wave_bitonic(int (&v)[NR], int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64 * NR; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            if (j >= 64) {   // partner in another register of the same lane
                const int rj = j >> 6;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (r & rj) continue;
                    const bool up = (((r << 6) | lane) & k2) == 0;
                    const int lo = min(v[r], v[r | rj]), hi = max(v[r], v[r | rj]);
                    v[r] = up ? lo : hi;
                    v[r | rj] = up ? hi : lo;
                }
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int other = __shfl_xor(v[r], j);
                    const bool up = (((r << 6) | lane) & k2) == 0, low = (lane & j) == 0;
                    v[r] = (low == up) ? min(v[r], other) : max(v[r], other);
                }
            }
        }
    }
}

constexpr int LONG_ROW = 256;       // rows longer than this go to csr_long_kernel (a workgroup and LDS per row)
constexpr int LONG_LDS = 16384;     // ... which sorts up to this many positions in LDS

// one wave per destination row: its positions in ascending order.  Rows longer than LONG_ROW are queued: long_list[0] = their
// number, then (batch element, destination) pairs -- queue order does not matter, every row is written to its own slice.
__global__ __launch_bounds__(256) void csr_rank_kernel(int t, int n, const int *__restrict__ seg, const int *__restrict__ slots, int *__restrict__ order,
                                                     int *__restrict__ long_list, int long_cap) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= n) return;
    const int *sb = seg + (size_t)b * (n + 1);
    const int base = sb[d], len = sb[d + 1] - base;
    const int *src = slots + (size_t)b * t + base;
    int *dst = order + (size_t)b * t + base;
    if (len <= 0) return;
    if (len <= 24) {   // all-pairs ranks: len broadcasts
        const int mine = lane < len ? src[lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int j = 0; j < len; ++j) rank += __shfl(mine, j) < mine;
        if (lane < len) dst[rank] = mine;
    } else if (len <= 64) {
        int v[1] = {lane < len ? src[lane] : 0x7FFFFFFF};
        wave_bitonic<1>(v, lane);
        if (lane < len) dst[lane] = v[0];
    } else if (len <= LONG_ROW) {
        int v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = r * 64 + lane < len ? src[r * 64 + lane] : 0x7FFFFFFF;
        wave_bitonic<4>(v, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r * 64 + lane < len) dst[r * 64 + lane] = v[r];
    } else if (lane == 0) {
        const int at = atomicAdd(&long_list[0], 1);
        if (at < long_cap) { long_list[1 + 2 * at] = b; long_list[2 + 2 * at] = d; }
    }
}

// the queued long rows (a point that is everybody's neighbour: the refined cloud of an untrained network is a blob whose surface
// points sit in thousands of lists).  A row's slots were handed out chunk by chunk (hist[g][d] = first slot of chunk g's share, the
// scan's output, which fill only copied), and every position of chunk g is below every position of chunk g + 1: the CHUNKS shares
// are sorted independently -- one workgroup per (row, chunk) task, bitonic sort in LDS; a share beyond LONG_LDS entries (one row
// holding a whole chunk of a very long list) falls back to all-pairs ranks from memory.  (Sorting a 5000-entry row whole took one
// workgroup 91 passes over 8192 LDS slots; its 16 shares of ~300 take 45 passes over 512 each, side by side.)
__global__ __launch_bounds__(256) void csr_long_kernel(int t, int n, const int *__restrict__ seg, const int *__restrict__ hist,
                                                     const int *__restrict__ slots, int *__restrict__ order, const int *__restrict__ long_list,
                                                     int long_cap) {
    __shared__ int buf[LONG_LDS];
    const int tid = threadIdx.x;
    const int tasks = min(long_list[0], long_cap) * CHUNKS;
    for (int q = blockIdx.x; q < tasks; q += gridDim.x) {
        const int b = long_list[1 + 2 * (q / CHUNKS)], d = long_list[2 + 2 * (q / CHUNKS)], g = q % CHUNKS;
        const int *hb = hist + (size_t)b * CHUNKS * n + d;
        const int base = hb[(size_t)g * n];
        const int len = (g + 1 < CHUNKS ? hb[(size_t)(g + 1) * n] : seg[(size_t)b * (n + 1) + d + 1]) - base;
        const int *src = slots + (size_t)b * t + base;
        int *dst = order + (size_t)b * t + base;
        if (len <= 0) continue;   // workgroup-uniform
        if (len <= LONG_LDS) {
            int p2 = 64;
            while (p2 < len) p2 <<= 1;
            for (int e = tid; e < p2; e += 256) buf[e] = e < len ? src[e] : 0x7FFFFFFF;
            __syncthreads();
            for (int k2 = 2; k2 <= p2; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int e = tid; e < p2; e += 256) {
                        const int partner = e ^ j;
                        if (partner > e) {
                            const int x = buf[e], y = buf[partner];
                            const bool up = (e & k2) == 0;
                            if ((x > y) == up) { buf[e] = y; buf[partner] = x; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int e = tid; e < len; e += 256) dst[e] = buf[e];
            __syncthreads();
        } else {
            for (int e = tid; e < len; e += 256) {
                const int mine = src[e];
                int rank = 0;
                for (int j = 0; j < len; ++j) rank += src[j] < mine;
                dst[rank] = mine;
            }
        }
    }
}

// at most t / LONG_ROW rows of one batch element can be longer than LONG_ROW
int long_rows_cap(int b, int t) { return b * (t / LONG_ROW + 1); }

}  // namespace

MCP_EXPORT size_t mcp_scatter_segments_workspace_bytes(int b, int t, int n) {
    if (b <= 0 || t <= 0 || n <= 0) return 0;
    if ((size_t)n * sizeof(int) > 160 * 1024) return 0;   // the chunk histograms live in LDS
    return ((size_t)b * CHUNKS * n + (size_t)b * t + 2 + 2 * (size_t)long_rows_cap(b, t)) * sizeof(int);
}

MCP_EXPORT int mcp_scatter_segments(int b, int t, int n, const int *idx, int *order, int *seg, void *workspace, size_t workspace_bytes,
                                    mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && t > 0 && n > 0 && idx && order && seg && workspace);
    if (workspace_bytes < mcp_scatter_segments_workspace_bytes(b, t, n)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)n * sizeof(int) > 160 * 1024) return MCP_ERR_UNSUPPORTED;
    int *hist = reinterpret_cast<int *>(workspace), *slots = hist + (size_t)b * CHUNKS * n, *long_list = slots + (size_t)b * t;
    const int cap = long_rows_cap(b, t);
    const hipError_t e = hipMemsetAsync(long_list, 0, sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t a_ = hipFuncSetAttribute(reinterpret_cast<const void *>(csr_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (a_ == hipSuccess) a_ = hipFuncSetAttribute(reinterpret_cast<const void *>(csr_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (a_ != hipSuccess) return (int)a_;
        attr_once.done();
    }
    const int per = (t + CHUNKS - 1) / CHUNKS;
    const size_t lds = (size_t)n * sizeof(int);
    hipLaunchKernelGGL(csr_count_kernel, dim3(CHUNKS, b), dim3(CT), lds, s, t, n, per, idx, hist);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(b), dim3(1024), 0, s, n, hist, seg);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(CHUNKS, b), dim3(CT), lds, s, t, n, per, idx, hist, slots);
    hipLaunchKernelGGL(csr_rank_kernel, dim3((n + 3) / 4, b), dim3(256), 0, s, t, n, seg, slots, order, long_list, cap);
    hipLaunchKernelGGL(csr_long_kernel, dim3(1024), dim3(256), 0, s, t, n, seg, hist, slots, order, long_list, cap);
    return mcp_launch_status();
}
