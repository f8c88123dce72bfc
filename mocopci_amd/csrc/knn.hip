// knn.hip -- fused brute-force K-nearest-neighbour search for gfx950.
//
// Replaces  square_distance + torch.topk  (mocopci.py:1130-1169 = pointconv_util.py:67-140)
// and pytorch3d.ops.knn_points (pointconv_util.py:910).  The reference materialises a
// (B,Q,N) fp32 distance matrix (24 B of HBM traffic per pair); this kernel never does:
//
//   * lane = one query (coordinates + |q|^2 in VGPRs), wave = 64 queries;
//   * reference points stream through a per-wave, double-buffered LDS tile of float4
//     (x,y,z,|r|^2); every lane reads the same float4 (LDS broadcast, conflict-free);
//   * selection: a per-lane threshold tau (the current K-th distance) filters candidates into
//     a per-lane LDS queue (column layout [slot][lane], conflict-free); when any lane's queue
//     is nearly full the wave sorts its queues with a register bitonic network and merges
//     them into the per-lane sorted K-list, tightening tau;
//   * small Q: SPLIT waves of a workgroup scan disjoint slices of the reference set for the
//     same 64 queries and merge their K-lists through LDS, so B*Q/64 < #SIMDs still fills
//     the chip.
//
// Result definition (same as oracle/pointset_oracle.c:orc_knn): the K smallest under the
// lexicographic order (distance, index), ascending; distances in the canon of common.h.
// Keys are the 64-bit (distance, index) keys of topk.h: a compare-exchange is one v_min_f64 / v_max_f64 pair.
#include "common.h"
#include "topk.h"

namespace {

typedef mcp_key u64;  // 64-bit (distance, index) key, see topk.h
#define KEY_INF MCP_KEY_INF
constexpr int TILE = 64;   // reference points per LDS tile (per wave); small, so that the N = 256 .. 1024 searches can be split over 2-4 waves

template <int MODE>
__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float qn, const float4 r) {
    if (MODE == MCP_DIST_EXPANSION) return mcp_expdist(qx, qy, qz, qn, r.x, r.y, r.z, r.w);
    return mcp_sqdist3(qx, qy, qz, r.x, r.y, r.z);
}

template <int MODE>
__device__ __forceinline__ float4 load_ref(const float *__restrict__ ref, int i, int n) {
    if (i < n) {
        const float x = ref[(size_t)i * 3 + 0], y = ref[(size_t)i * 3 + 1], z = ref[(size_t)i * 3 + 2];
        return make_float4(x, y, z, mcp_sqnorm3(x, y, z));
    }
    // padding: distance evaluates to +inf in both forms, so it never passes "d < tau"
    return MODE == MCP_DIST_EXPANSION ? make_float4(0.f, 0.f, 0.f, INFINITY) : make_float4(INFINITY, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// K <= 4: sorted 4-list in registers, insertion guarded by a wave-uniform branch.
// ---------------------------------------------------------------------------------------------
template <int MODE, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void knn_small_kernel(int q, int n, int kout, const float *__restrict__ query,
                                                               const float *__restrict__ ref, int *__restrict__ idx,
                                                               float *__restrict__ dist) {
    __shared__ float4 tiles[SPLIT][2][TILE];
    __shared__ u64 mrg[SPLIT][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 64 + lane;
    const bool live = qi < q;
    const float *qp = query + ((size_t)b * q + (live ? qi : 0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qn = mcp_sqnorm3(qx, qy, qz);
    ref += (size_t)b * n * 3;

    // slice of the reference set for this wave, in whole tiles
    const int ntiles = (n + TILE - 1) / TILE;
    const int t0 = (int)((long long)ntiles * wave / SPLIT), t1 = (int)((long long)ntiles * (wave + 1) / SPLIT);

    u64 a[4] = {KEY_INF, KEY_INF, KEY_INF, KEY_INF};
    float tau = INFINITY;
    float4(*tile)[TILE] = tiles[wave];

    if (t0 < t1) {
#pragma unroll
        for (int u = 0; u < TILE / 64; ++u) tile[0][lane + 64 * u] = load_ref<MODE>(ref, t0 * TILE + lane + 64 * u, n);
    }
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        float4 nxt[TILE / 64];
        if (t + 1 < t1) {
#pragma unroll
            for (int u = 0; u < TILE / 64; ++u) nxt[u] = load_ref<MODE>(ref, (t + 1) * TILE + lane + 64 * u, n);
        }
        __builtin_amdgcn_wave_barrier();
        const int base = t * TILE;
        constexpr int G = 8;  // references per group: loads of group g+1 are in flight under the math of group g
        float4 rc[G];
#pragma unroll
        for (int u = 0; u < G; ++u) rc[u] = tile[cur][u];
        for (int r0 = 0; r0 < TILE; r0 += G) {
            float4 rn[G];
            const int rnext = r0 + G < TILE ? r0 + G : r0;
#pragma unroll
            for (int u = 0; u < G; ++u) rn[u] = tile[cur][rnext + u];
            float d[G];
            bool any = false;
#pragma unroll
            for (int u = 0; u < G; ++u) {
                d[u] = pair_dist<MODE>(qx, qy, qz, qn, rc[u]);
                any |= d[u] < tau;
            }
            if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    if (__builtin_amdgcn_ballot_w64(d[u] < tau)) {
                        u64 key = mcp_make_key(d[u], (uint32_t)(base + r0 + u));
                        key = d[u] < tau ? key : KEY_INF;
                        a[3] = mcp_key_min(key, a[3]);
                        mcp_ce_asc(a[2], a[3]);
                        mcp_ce_asc(a[1], a[2]);
                        mcp_ce_asc(a[0], a[1]);
                        tau = mcp_tau_of(a[3]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < G; ++u) rc[u] = rn[u];
        }
        if (t + 1 < t1) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < TILE / 64; ++u) tile[cur ^ 1][lane + 64 * u] = nxt[u];
        }
    }
    if (SPLIT > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mrg[wave][j][lane] = a[j];
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < SPLIT; ++w) {
            u64 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = mrg[w][j][lane];
            mcp_merge_sorted<4, 4>(a, o);
        }
    }
    if (!live) return;
    int *oi = idx + ((size_t)b * q + qi) * kout;
    float *od = dist ? dist + ((size_t)b * q + qi) * kout : nullptr;
    mcp_store_list<4>(a, kout, oi, od);
}

// ---------------------------------------------------------------------------------------------
// K in {16, 32}: threshold-filtered LDS queues + register bitonic merges.
// ---------------------------------------------------------------------------------------------
template <int K>
struct KnnLds {
    static constexpr int QS = 16;                       // queue slots per lane
    static constexpr int TILE_BYTES = 2 * TILE * 16;    // double-buffered float4 tile
    static constexpr int QUEUE_BYTES = QS * 64 * 8;     // [QS][64] (d, idx) pairs
    static constexpr int MERGE_BYTES = K * 64 * 8;      // [K][64] keys, aliases tile+queue after the scan
    static constexpr int SCAN_BYTES = TILE_BYTES + QUEUE_BYTES;
    static constexpr int WAVE_BYTES = SCAN_BYTES > MERGE_BYTES ? SCAN_BYTES : MERGE_BYTES;
};

template <int K, int MODE, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void knn_queue_kernel(int q, int n, int kout, const float *__restrict__ query,
                                                               const float *__restrict__ ref, int *__restrict__ idx,
                                                               float *__restrict__ dist) {
    using L = KnnLds<K>;
    constexpr int QS = L::QS;
    constexpr int CHK = 4;  // refs between queue-full checks
    extern __shared__ float4 smem_f4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *wbase = reinterpret_cast<char *>(smem_f4) + (size_t)wave * L::WAVE_BYTES;
    float4(*tile)[TILE] = reinterpret_cast<float4(*)[TILE]>(wbase);
    uint2(*queue)[64] = reinterpret_cast<uint2(*)[64]>(wbase + L::TILE_BYTES);

    const int b = blockIdx.y;
    const int qi = blockIdx.x * 64 + lane;
    const bool live = qi < q;
    const float *qp = query + ((size_t)b * q + (live ? qi : 0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qn = mcp_sqnorm3(qx, qy, qz);
    ref += (size_t)b * n * 3;

    const int ntiles = (n + TILE - 1) / TILE;
    const int t0 = (int)((long long)ntiles * wave / SPLIT), t1 = (int)((long long)ntiles * (wave + 1) / SPLIT);

    u64 a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = KEY_INF;
    float tau = INFINITY;
    int cnt = 0;

    auto flush = [&]() {
        mcp_flush_queue<K, QS>(a, queue, lane, cnt);
        tau = mcp_tau_of(a[K - 1]);
        cnt = 0;
    };

    if (t0 < t1) {
#pragma unroll
        for (int u = 0; u < TILE / 64; ++u) tile[0][lane + 64 * u] = load_ref<MODE>(ref, t0 * TILE + lane + 64 * u, n);
    }
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        float4 nxt[TILE / 64];
        if (t + 1 < t1) {
#pragma unroll
            for (int u = 0; u < TILE / 64; ++u) nxt[u] = load_ref<MODE>(ref, (t + 1) * TILE + lane + 64 * u, n);
        }
        __builtin_amdgcn_wave_barrier();
        const int base = t * TILE;
        // software pipeline over groups of CHK references: the next group's float4s are in flight while this
        // group's distances and queue pushes run (the compiler will not hoist LDS reads above the queue writes)
        float4 rc[CHK];
#pragma unroll
        for (int u = 0; u < CHK; ++u) rc[u] = tile[cur][u];
        for (int r0 = 0; r0 < TILE; r0 += CHK) {
            float4 rn[CHK];
            const int rnext = r0 + CHK < TILE ? r0 + CHK : r0;  // last group re-reads itself (harmless)
#pragma unroll
            for (int u = 0; u < CHK; ++u) rn[u] = tile[cur][rnext + u];
            float d[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u) d[u] = pair_dist<MODE>(qx, qy, qz, qn, rc[u]);
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                if (d[u] < tau) {
                    queue[cnt][lane] = make_uint2(__float_as_uint(d[u]), (uint32_t)(base + r0 + u));
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QS - CHK)) flush();
#pragma unroll
            for (int u = 0; u < CHK; ++u) rc[u] = rn[u];
        }
        if (t + 1 < t1) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < TILE / 64; ++u) tile[cur ^ 1][lane + 64 * u] = nxt[u];
        }
    }
    flush();

    if (SPLIT > 1) {
        __syncthreads();  // every wave is done with its tile/queue region before it is reused for keys
        u64(*mrg)[64] = reinterpret_cast<u64(*)[64]>(wbase);
#pragma unroll
        for (int j = 0; j < K; ++j) mrg[j][lane] = a[j];
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < SPLIT; ++w) {
            u64(*om)[64] = reinterpret_cast<u64(*)[64]>(reinterpret_cast<char *>(smem_f4) + (size_t)w * L::WAVE_BYTES);
            u64 o[K];
#pragma unroll
            for (int j = 0; j < K; ++j) o[j] = om[j][lane];
            mcp_merge_sorted<K, K>(a, o);
        }
    }
    if (!live) return;
    int *oi = idx + ((size_t)b * q + qi) * kout;
    float *od = dist ? dist + ((size_t)b * q + qi) * kout : nullptr;
    mcp_store_list<K>(a, kout, oi, od);
}

// Chamfer helper: nearest squared distance from every x to the set y (direct form).
__global__ __launch_bounds__(256) void nn1_kernel(int n, int m, const float *__restrict__ x, const float *__restrict__ y,
                                                  float *__restrict__ out) {
    __shared__ float tile[1024 * 3];
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < n;
    const float *u = x + ((size_t)b * n + (live ? p : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *yb = y + (size_t)b * m * 3;
    float best = INFINITY;
    for (int base = 0; base < m; base += 1024) {
        const int len = min(1024, m - base);
        __syncthreads();
        for (int i = threadIdx.x; i < len * 3; i += 256) tile[i] = yb[(size_t)base * 3 + i];
        __syncthreads();
        for (int k = 0; k < len; ++k) best = fminf(best, mcp_sqdist3(ux, uy, uz, tile[k * 3], tile[k * 3 + 1], tile[k * 3 + 2]));
    }
    if (live) out[(size_t)b * n + p] = best;
}

int pick_split(int b, int q, int n) {
    // aim for >= 2 waves per SIMD (1024 SIMDs) while keeping at least two 64-reference tiles per wave.  (The rule used to keep
    // two tiles per wave: the N = 512 / 1024 searches of the lower pyramid levels then ran 384-768 waves of 512 sequential
    // references each on 1024 SIMDs, ~40 us launches on the critical path; splitting them took 0.3 ms off the step.)
    const long long waves = (long long)b * ((q + 63) / 64);
    const int ntiles = (n + TILE - 1) / TILE;
    int split = 1;
    while (split < 8 && waves * split < 2048 && ntiles >= 2 * split) split *= 2;
    return split;
}

template <int MODE, int SPLIT>
int launch_small(int b, int q, int n, int k, const float *query, const float *ref, int *idx, float *dist, hipStream_t s) {
    hipLaunchKernelGGL((knn_small_kernel<MODE, SPLIT>), dim3(mcp_divup(q, 64), b), dim3(64 * SPLIT), 0, s, q, n, k, query, ref, idx,
                       dist);
    return mcp_launch_status();
}
template <int K, int MODE, int SPLIT>
int launch_queue(int b, int q, int n, int k, const float *query, const float *ref, int *idx, float *dist, hipStream_t s) {
    const size_t lds = (size_t)KnnLds<K>::WAVE_BYTES * SPLIT;
    auto kern = knn_queue_kernel<K, MODE, SPLIT>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 64), b), dim3(64 * SPLIT), lds, s, q, n, k, query, ref, idx, dist);
    return mcp_launch_status();
}
template <int MODE, int SPLIT>
int dispatch_k(int b, int q, int n, int k, const float *query, const float *ref, int *idx, float *dist, hipStream_t s) {
    if (k <= 4) return launch_small<MODE, SPLIT>(b, q, n, k, query, ref, idx, dist, s);
    if (k <= 16) return launch_queue<16, MODE, SPLIT>(b, q, n, k, query, ref, idx, dist, s);
    return launch_queue<32, MODE, SPLIT>(b, q, n, k, query, ref, idx, dist, s);
}
template <int MODE>
int dispatch_split(int split, int b, int q, int n, int k, const float *query, const float *ref, int *idx, float *dist,
                   hipStream_t s) {
    if (split == 1) return dispatch_k<MODE, 1>(b, q, n, k, query, ref, idx, dist, s);
    if (split == 2) return dispatch_k<MODE, 2>(b, q, n, k, query, ref, idx, dist, s);
    if (split == 4) return dispatch_k<MODE, 4>(b, q, n, k, query, ref, idx, dist, s);
    return dispatch_k<MODE, 8>(b, q, n, k, query, ref, idx, dist, s);
}

}  // namespace

MCP_EXPORT int mcp_knn(int b, int q, int n, int k, int dist_form, const float *query, const float *ref, int *idx, float *dist,
                       mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && q > 0 && n > 0 && k > 0 && query && ref && idx);
    MCP_CHECK_ARGS(dist_form == MCP_DIST_EXPANSION || dist_form == MCP_DIST_DIRECT);
    if (k > 32) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int split = pick_split(b, q, n);
    mcp_prof_begin(MCP_KERNEL_KNN, s);
    const int rc = dist_form == MCP_DIST_EXPANSION ? dispatch_split<MCP_DIST_EXPANSION>(split, b, q, n, k, query, ref, idx, dist, s)
                                                   : dispatch_split<MCP_DIST_DIRECT>(split, b, q, n, k, query, ref, idx, dist, s);
    mcp_prof_end(MCP_KERNEL_KNN, s);
    return rc;
}

MCP_EXPORT int mcp_chamfer_nn(int b, int n, int m, const float *x, const float *y, float *dxy, float *dyx, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && m > 0 && x && y && dxy && dyx);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nn1_kernel, dim3(mcp_divup(n, 256), b), dim3(256), 0, s, n, m, x, y, dxy);
    hipLaunchKernelGGL(nn1_kernel, dim3(mcp_divup(m, 256), b), dim3(256), 0, s, m, n, y, x, dyx);
    return mcp_launch_status();
}
