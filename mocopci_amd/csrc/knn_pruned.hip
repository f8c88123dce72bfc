// knn_pruned.hip -- exact KNN with spatial pruning for large reference sets (gfx950).
//
// Same result definition as knn.hip / the oracle (K smallest under (distance, index), ascending,
// distances in the shared fp32 canon), but the reference cloud is visited selectively:
//   * queries and references are Morton-sorted once per cloud (mcp_build_cloud: one launch per cloud up to 16384 points;
//     mcp_morton_codes / mcp_tile_boxes for larger ones, with the sort left to the caller), with one axis-aligned box per
//     tile of PT consecutive sorted references;
//   * a wave owns 16 consecutive SORTED queries (a compact region) x 4 lanes each, computes the lower bound of the
//     squared distance from its query box to every tile box, and visits tiles in ascending bound order; it stops as soon
//     as the smallest unvisited bound exceeds the largest per-lane K-th distance plus a slack that covers the rounding of
//     the distance expression, and skips a tile no query of the wave can use (point-to-box test) -- every skipped
//     reference would have failed the per-lane "d <= tau" test anyway, so the output is bit-identical to the exhaustive
//     scan;
//   * within a visited tile the scan / threshold queue / register bitonic merge are those of knn.hip; candidates carry
//     the ORIGINAL reference index (tie order is defined on it), and "d <= tau" (not "<") is used because tiles are no
//     longer visited in index order;
//   * results are written to the query's original row (qperm).
// At N=8192 a wave scans ~900 of the 8192 references (64-point tiles measured 5-7 % faster than 128/256).
#include <stdlib.h>

#include <rocprim/block/block_radix_sort.hpp>

#include "common.h"
#include "topk.h"

namespace {

typedef mcp_key u64;  // 64-bit (distance, index) key, see topk.h
#define KEY_INF MCP_KEY_INF
#ifndef MCP_PRUNED_PT
#define MCP_PRUNED_PT 64
#endif
constexpr int PT = MCP_PRUNED_PT;  // references per tile
constexpr int MAX_TPL = 16;  // tiles per lane -> up to 1024 tiles (N <= 65536 at 64 references per tile)

__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- preprocessing kernels ------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t spread2_10(uint32_t v) {  // 10 bits -> every second bit
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// Which axes the space-filling order runs over (round 5).  The cells are isotropic (one scale for the three axes: per-axis scaling
// would slice a flat LiDAR cloud into thin slabs with a huge footprint), but an isotropic 3-D code still spends every third bit on the
// THIN axis at the scale of a tile: an 80 x 80 x 6 scan of 8192 points has tiles of ~300 unit^3, which the Z-curve cuts into two
// layers of ~10 x 10 x 3 instead of one column of 7 x 7 x 6 -- and a query's neighbourhood (a ball wider than the slab is thick) then
// meets twice as many tiles.  So: let s = the side of the footprint a PT-point tile would have if the thinnest axis were ignored,
// sqrt(PT * (product of the other two extents) / n); when the cloud is no thicker than that, the code runs over the other two axes
// only (10 bits each).  The order only steers the pruning -- any order gives the same neighbours, bit for bit.
struct CloudCode {
    int thin;      // axis left out of the code, -1: none (3-D code)
    float inv;     // cells per unit length
};
__device__ __forceinline__ CloudCode cloud_code(const float *bbox, int n, int bits3, int bits2) {
    const float e[3] = {bbox[3] - bbox[0], bbox[4] - bbox[1], bbox[5] - bbox[2]};
    const float ext = fmaxf(fmaxf(e[0], e[1]), e[2]);
    const int t = e[0] <= e[1] ? (e[0] <= e[2] ? 0 : 2) : (e[1] <= e[2] ? 1 : 2);
    const float area = e[(t + 1) % 3] * e[(t + 2) % 3];
    CloudCode c;
    c.thin = (area > 0.f && e[t] * e[t] * (float)n <= (float)PT * area) ? t : -1;
    c.inv = ext > 0.f ? (float)(1 << (c.thin < 0 ? bits3 : bits2)) / ext : 0.f;
    return c;
}
template <int BITS2>
__device__ __forceinline__ uint32_t cloud_key(const float *p, const float *bbox, const CloudCode cc, int bits3) {
    uint32_t c[3];
    const float top = (float)((1 << (cc.thin < 0 ? bits3 : BITS2)) - 1);
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = (uint32_t)fminf(fmaxf((p[a] - bbox[a]) * cc.inv, 0.f), top);
    if (cc.thin < 0) {
#ifdef MCP_CLOUD_MORTON3D
        return spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
#else
        // three axes: the Hilbert curve here too (Skilling's transposition, "Programming the Hilbert curve", 2004: undo the excess
        // rotations level by level, then Gray-encode), for the same reason as in the plane -- a run of PT consecutive points is one
        // connected blob instead of the Z-curve's two or three
        uint32_t x[3] = {c[0], c[1], c[2]};
        for (uint32_t q = 1u << (bits3 - 1); q > 1; q >>= 1) {
            const uint32_t pm = q - 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (x[i] & q) x[0] ^= pm;
                else { const uint32_t t = (x[0] ^ x[i]) & pm; x[0] ^= t; x[i] ^= t; }
            }
        }
        x[1] ^= x[0];
        x[2] ^= x[1];
        uint32_t t = 0;
        for (uint32_t q = 1u << (bits3 - 1); q > 1; q >>= 1)
            if (x[2] & q) t ^= q - 1;
        x[0] ^= t; x[1] ^= t; x[2] ^= t;
        return (spread10(x[0]) << 2) | (spread10(x[1]) << 1) | spread10(x[2]);
#endif
    }
    uint32_t u = cc.thin == 0 ? c[1] : c[0], v = cc.thin == 2 ? c[1] : c[2];
#ifdef MCP_CLOUD_Z2D
    return spread2_10(u) | (spread2_10(v) << 1);
#else
    // two axes: the Hilbert curve's index of cell (u, v) on the 2^BITS2 x 2^BITS2 grid.  A run of PT consecutive points is then a
    // CONNECTED piece of the plane (the Z-curve jumps: a run is often two or three separate blocks inside one large box), so the
    // tile boxes are tighter and a walk meets fewer of them.
    uint32_t d = 0;
    constexpr uint32_t TOP = (1u << BITS2) - 1u;
#pragma unroll
    for (uint32_t sft = BITS2; sft-- > 0;) {
        const uint32_t rx = (u >> sft) & 1u, ry = (v >> sft) & 1u;
        d = (d << 2) | ((3u * rx) ^ ry);
        if (ry == 0) {   // rotate / reflect the quadrant so that the sub-curve enters and leaves where its neighbours expect it
            if (rx) { u = TOP - u; v = TOP - v; }
            const uint32_t t = u; u = v; v = t;
        }
    }
    return d;
#endif
}
// Morton code (30 bits over three axes, or 20 over two: cloud_code) on the per-batch box [lo, hi] (box (B,6): lo xyz, hi xyz)
__global__ __launch_bounds__(256) void morton_kernel(int n, const float *__restrict__ xyz, const float *__restrict__ box,
                                                     int *__restrict__ codes) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *bx = box + b * 6;
    codes[(size_t)b * n + i] = (int)cloud_key<10>(xyz + ((size_t)b * n + i) * 3, bx, cloud_code(bx, n, 10, 10), 10);
}
// one wave per tile of PT sorted points: (lo xyz, hi xyz)
__global__ __launch_bounds__(64) void tile_box_kernel(int n, int tiles, const float *__restrict__ sorted_xyz, float *__restrict__ boxes) {
    const int b = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = t * PT + lane; i < min(n, (t + 1) * PT); i += 64) {
        const float *p = sorted_xyz + ((size_t)b * n + i) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], p[a]);
            hi[a] = fmaxf(hi[a], p[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_minf(lo[a]);
        hi[a] = wave_maxf(hi[a]);
    }
    if (lane == 0) {
        float *o = boxes + ((size_t)b * tiles + t) * 6;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
    }
}

// ---- fused cloud builder: bbox -> Morton keys -> block radix sort -> gather -> tile boxes -----------------------
// One workgroup per batch element (N <= 16384).  The sort is rocPRIM's block_radix_sort on (30-bit code, index) pairs
// held in registers (IPT per thread): stable, so points of one cell stay in index order -- the same permutation as
// sorting (code << 32 | index).
constexpr int BT = 1024;
// Code width of the fused builder: 15 bits either way -- 5 per axis over three axes (32768 cells) or 7 + 7 over two (16384 cells, Hilbert
// order) -- so that with the padding bit above them the keys sort in TWO 8-bit radix passes (rocPRIM's match ranking; round 4's 21-bit
// codes took three).  The order only steers the pruning: <= 16384 points have no use for finer cells than a tile is wide.
#ifndef MCP_CLOUD_CELL_BITS
#define MCP_CLOUD_CELL_BITS 5
#endif
constexpr int CELL_BITS = MCP_CLOUD_CELL_BITS, CELL_BITS_2D = 7, CODE_BITS = 15;
static_assert(3 * CELL_BITS <= CODE_BITS && 2 * CELL_BITS_2D <= CODE_BITS, "codes stay below the padding bit");
template <int IPT>
struct CloudSort {
    using Sort = rocprim::block_radix_sort<uint32_t, BT, IPT, uint32_t>;
    union Lds {
        typename Sort::storage_type sort;
    };
};

template <int IPT>
__global__ __launch_bounds__(BT) void build_cloud_kernel(int n, int tiles, const float *__restrict__ xyz,
                                                         float *__restrict__ sorted_xyz, int *__restrict__ perm,
                                                         float *__restrict__ boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long dyn[];
    // dynamic LDS: [0,512) reduction scratch + bbox, then the sort storage
    float(*red)[BT / 64] = reinterpret_cast<float(*)[BT / 64]>(dyn);     // [6][16] floats = 384 B
    float *bbox = reinterpret_cast<float *>(dyn) + 6 * (BT / 64);          // [6]
    typename CloudSort<IPT>::Lds &lds = *reinterpret_cast<typename CloudSort<IPT>::Lds *>(dyn + 64);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)b * n * 3;
    sorted_xyz += (size_t)b * n * 3;
    perm += (size_t)b * n;
    boxes += (size_t)b * tiles * 6;
    // 1. bounding box of the cloud; thread t holds points t*IPT .. t*IPT+IPT-1 (kept in registers for the keys)
    float px[IPT], py[IPT], pz[IPT];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = min(tid * IPT + u, n - 1);   // padding repeats the last point: no effect on the box
        px[u] = xyz[(size_t)i * 3 + 0]; py[u] = xyz[(size_t)i * 3 + 1]; pz[u] = xyz[(size_t)i * 3 + 2];
        lo[0] = fminf(lo[0], px[u]); hi[0] = fmaxf(hi[0], px[u]);
        lo[1] = fminf(lo[1], py[u]); hi[1] = fmaxf(hi[1], py[u]);
        lo[2] = fminf(lo[2], pz[u]); hi[2] = fmaxf(hi[2], pz[u]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_minf(lo[a]);
        hi[a] = wave_maxf(hi[a]);
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < BT / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        bbox[tid] = v;
    }
    __syncthreads();
    // 2. keys: isotropic cells over three axes or, for a cloud flatter than a tile is wide, over two (cloud_code); padding sorts
    //    last (the bit above the codes)
    const CloudCode cc = cloud_code(bbox, n, CELL_BITS, CELL_BITS_2D);
    uint32_t keys[IPT], vals[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = tid * IPT + u;
        const float p[3] = {px[u], py[u], pz[u]};
        keys[u] = i < n ? cloud_key<CELL_BITS_2D>(p, bbox, cc, CELL_BITS) : 1u << CODE_BITS;
        vals[u] = (uint32_t)i;
    }
    // 3. stable radix sort of the (code, index) pairs over the code's bits and the padding bit: points of one cell stay in index order
    typename CloudSort<IPT>::Sort().sort(keys, vals, lds.sort, 0, CODE_BITS + 1);
    // 4. permutation + sorted coordinates (sorted position s = tid*IPT + u), and the tile boxes from the same registers: a tile of PT
    //    sorted points is held by PT / IPT consecutive threads
    float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int s = tid * IPT + u;
        if (s < n) {
            const int src = (int)vals[u];
            const float x = xyz[(size_t)src * 3 + 0], y = xyz[(size_t)src * 3 + 1], z = xyz[(size_t)src * 3 + 2];
            perm[s] = src;
            sorted_xyz[(size_t)s * 3 + 0] = x;
            sorted_xyz[(size_t)s * 3 + 1] = y;
            sorted_xyz[(size_t)s * 3 + 2] = z;
            tl[0] = fminf(tl[0], x); th[0] = fmaxf(th[0], x);
            tl[1] = fminf(tl[1], y); th[1] = fmaxf(th[1], y);
            tl[2] = fminf(tl[2], z); th[2] = fmaxf(th[2], z);
        }
    }
    constexpr int TPT = PT / IPT;   // threads per tile: 8 (IPT = 8) .. 64 (IPT = 1); IPT = 16 (PT / IPT = 4) likewise
    static_assert(PT % IPT == 0 && TPT >= 1 && TPT <= 64 && (TPT & (TPT - 1)) == 0, "a tile is held by a power-of-two run of lanes");
#pragma unroll
    for (int o = TPT / 2; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            tl[a] = fminf(tl[a], __shfl_xor(tl[a], o));
            th[a] = fmaxf(th[a], __shfl_xor(th[a], o));
        }
    }
    const int t = tid / TPT;
    if ((tid & (TPT - 1)) == 0 && t < tiles) {
        float *o = boxes + (size_t)t * 6;
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = th[0]; o[4] = th[1]; o[5] = th[2];
    }
}

template <int IPT>
int launch_build_cloud(int b, int n, int tiles, const float *xyz, float *sorted_xyz, int *perm, float *boxes, hipStream_t s) {
    auto kern = build_cloud_kernel<IPT>;
    const size_t lds = 512 + sizeof(typename CloudSort<IPT>::Lds);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(BT), lds, s, n, tiles, xyz, sorted_xyz, perm, boxes);
    return mcp_launch_status();
}

// ---- the search ------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float qn, const float4 r) {
    if (MODE == MCP_DIST_EXPANSION) return mcp_expdist(qx, qy, qz, qn, r.x, r.y, r.z, r.w);
    return mcp_sqdist3(qx, qy, qz, r.x, r.y, r.z);
}

// ---- the search (round 4): threshold walk, unsorted candidate lists, ONE exact sort per query -------------------------------
// Where round 3's kernel spent its cycles: every one of its four lanes per query kept a sorted K-list of 64-bit keys in
// registers (64 VGPRs) and paid a 176-compare-exchange register network -- two v_min/v_max_f64 each -- per queue flush, plus
// two more K + K merges at the end.  Here the 64-bit keys are sorted ONCE per query:
//   * a lane keeps only the KS = K/4 smallest DISTANCES it has seen (fp32, sorted; KS registers).  tau = the largest of the
//     four lanes' KS-th distances is an upper bound of the query's K-th distance (4 * KS references lie at or below it);
//   * a scanned reference with d <= tau is appended, unsorted, to the lane's own column of an LDS list (CL slots): one
//     unconditional ds_write at the write position, which advances only when the test passes -- no branch, no exec mask;
//   * when some lane has NQ new entries their distances are merged into its KS-list (fp32 network: one instruction per
//     min / max, against two double-rate ones for a key) and tau tightens; when some list is nearly full its entries above
//     tau are dropped in place ("soft" compaction: no sorting); should a list stay full (skewed or tied distances) the four
//     lanes sort their 4 * CL keys together and keep exactly the K smallest, KS per lane ("hard" compaction: always frees room);
//   * at the end the same quad network -- bitonic with mirror steps so that every compare-exchange is ascending; the two stages
//     that cross lanes go through DPP quad permutes -- sorts the survivors once, and lanes 0 / 1 of the quad write the row.
// Everything dropped on the way had d > tau >= the K-th distance, so the result is the exhaustive scan's, bit for bit; the
// walk over the tiles (ascending box bound, per-query point-to-box filter, rounding slack) is round 3's.
template <int K>
struct WalkCfg {
    static constexpr int KS = (K + 3) / 4;          // depth of a lane's distance list: 4 * KS >= K
    static constexpr int NET = K >= 16 ? 16 : 8;    // slots per lane that go through the quad network (4 * NET keys)
#ifndef MCP_KNN_CL32
#define MCP_KNN_CL32 32
#endif
    static constexpr int CL = K > 16 ? MCP_KNN_CL32 : 16;   // list slots per lane (K = 32: rule-b survivors average 11 per lane; 16 slots left no room)
    static constexpr int CHK = 4;                   // rows per scan call
    static constexpr int GAP = 8;                   // rows scanned between two capacity checks: a lane enters them with at most CL - GAP entries
    static constexpr int NQ = 8;                    // entries merged into a lane's KS-list per pass of a threshold update
#ifndef MCP_KNN_UPD_MIN
#define MCP_KNN_UPD_MIN 8
#endif
#ifndef MCP_KNN_UPD_END
#define MCP_KNN_UPD_END 6
#endif
    static constexpr int UPD_MIN = MCP_KNN_UPD_MIN; // at a capacity check: update the thresholds when some lane has this many new entries
    static constexpr int UPD_END = MCP_KNN_UPD_END; // at the end of a tile: likewise (tau steers the choice of the next tile)
    static constexpr int PRUNE_AFTER = 3;           // tiles in a row the per-query filter rejects before every remaining tile is filtered at once
    static constexpr int TILE_BYTES = PT * 16;      // float4 (x,y,z,|r|^2)
    static constexpr int PERM_BYTES = PT * 4;       // original indices, stored [sub][PT/4]
    static constexpr int SLOT = 384;                // per slot: the 64 lanes' distances (4 bytes each), then their original indices (2 bytes: n <= 65536)
    static constexpr int LIST_BYTES = CL * SLOT;
    static constexpr int WAVE_BYTES = TILE_BYTES + PERM_BYTES + LIST_BYTES;
    static_assert(KS < NET && NET <= CL && NET % KS == 0 && GAP <= CL - KS && GAP % CHK == 0 && NQ <= CL,
                  "a hard compaction shortens every list longer than KS (kept KS + what sat beyond slot NET): repeated, it makes room for GAP rows");
};

template <int CTRL>
__device__ __forceinline__ mcp_key dpp_key(mcp_key k) {
    return mcp_key_words(mcp_dpp<CTRL>(mcp_key_hi(k)), mcp_dpp<CTRL>(mcp_key_lo(k)));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(mcp_dpp<CTRL>(__float_as_uint(v))); }

// ascending compare-exchange on fp32 values (never NaN: list entries passed a `d <= tau` test)
__device__ __forceinline__ void ce_f(float &a, float &b) {
    const float lo = mcp_min_raw(a, b), hi = mcp_max_raw(a, b);
    a = lo;
    b = hi;
}
// in-lane part of the bitonic network on N values: blocks of k = 2 .. N, mirror step then xor steps, every exchange ascending
template <int N, typename T, typename CE>
__device__ __forceinline__ void sort_in_lane(T (&v)[N], CE ce) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int l = r ^ (k - 1);
            if (l > r) ce(v[r], v[l]);
        }
#pragma unroll
        for (int j = k >> 2; j > 0; j >>= 1)
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const int l = r ^ j;
                if (l > r) ce(v[r], v[l]);
            }
    }
}
// v bitonic (one mirror step already applied across a block of 2N) -> ascending, in-lane xor steps only
template <int N, typename T, typename CE>
__device__ __forceinline__ void merge_in_lane(T (&v)[N], CE ce) {
#pragma unroll
    for (int j = N >> 1; j > 0; j >>= 1)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int l = r ^ j;
            if (l > r) ce(v[r], v[l]);
        }
}

// The 4 * CL keys of a quad (element e = CL * sub + r), sorted as far as the result needs: on return lane 0 of the quad holds
// ranks 0 .. CL-1 and lane 1 ranks CL .. 2CL-1, ascending; lanes 2 and 3 hold nothing useful.
template <int CL>
__device__ __forceinline__ void quad_sort_low(mcp_key (&v)[CL], int sub) {
    auto ce = [](mcp_key &a, mcp_key &b) { mcp_ce_asc(a, b); };
    sort_in_lane<CL>(v, ce);
    const bool odd = (sub & 1) != 0;
    mcp_key w[CL];
    // blocks of 2 CL: element e against its mirror 2CL-1-e = (lane ^ 1, register CL-1-r); the even lane keeps the smaller key
#pragma unroll
    for (int r = 0; r < CL; ++r) {
        const mcp_key o = dpp_key<0xB1>(v[CL - 1 - r]);
        const mcp_key lo = mcp_key_min(v[r], o), hi = mcp_key_max(v[r], o);
        w[r] = odd ? hi : lo;
    }
    merge_in_lane<CL>(w, ce);
    // the block of 4 CL: mirror = (3 - lane, CL-1-r); only the lower half (lanes 0, 1) is wanted, so every lane takes the minimum
#pragma unroll
    for (int r = 0; r < CL; ++r) v[r] = mcp_key_min(w[r], dpp_key<0x1B>(w[CL - 1 - r]));
    // ... which is bitonic over lanes 0, 1: xor step CL = the same register of lane ^ 1
#pragma unroll
    for (int r = 0; r < CL; ++r) {
        const mcp_key o = dpp_key<0xB1>(v[r]);
        const mcp_key lo = mcp_key_min(v[r], o), hi = mcp_key_max(v[r], o);
        w[r] = odd ? hi : lo;
    }
    merge_in_lane<CL>(w, ce);
#pragma unroll
    for (int r = 0; r < CL; ++r) v[r] = w[r];
}

// ts (KS ascending) <- the KS smallest of ts U nv (NQ values, any order; +inf = absent)
template <int KS, int NQ>
__device__ __forceinline__ void merge_new(float (&ts)[KS], float (&nv)[NQ]) {
    auto ce = [](float &a, float &b) { ce_f(a, b); };
    if constexpr (KS == 1) {
        float m = nv[0];
#pragma unroll
        for (int j = 1; j < NQ; ++j) m = mcp_min_raw(m, nv[j]);
        ts[0] = mcp_min_raw(ts[0], m);
    } else {
        static_assert(KS <= NQ, "the mirror step below pairs ts[i] with nv[KS-1-i]");
        sort_in_lane<NQ>(nv, ce);
#pragma unroll
        for (int i = 0; i < KS; ++i) ts[i] = mcp_min_raw(ts[i], nv[KS - 1 - i]);   // mirror step of the 2 KS block: bitonic
        merge_in_lane<KS>(ts, ce);
    }
}

#ifdef MCP_KNN_DIAG
// diagnostic build only (never in the product library).  Counters: [0] waves [1] tiles scanned [2] tiles examined [3] threshold
// updates [4] soft compactions [5] hard compactions [6] appended entries [7] survivors at the end;  [8..15] shader cycles per
// phase (s_memtime deltas summed over waves): 8 setup, 9 walk (next tile + filters + fetch issue), 10 staging, 11 scan,
// 12 threshold update, 13 compaction, 14 final sort + store
__device__ unsigned long long g_walk_diag[16];
#define WALK_COUNT(slot, v) do { cn_[slot] += (unsigned)(v); } while (0)
#define WALK_STAMP(slot)                                                               \
    do {                                                                               \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        ph_[slot] += t_ - t_prev_;                                                     \
        t_prev_ = t_;                                                                  \
    } while (0)
#else
#define WALK_COUNT(slot, v)
#define WALK_STAMP(slot)
#endif

template <int K, int MODE, int TPL>
__global__ __launch_bounds__(64) void knn_walk_kernel(int q, int n, int tiles, int kout, const float *__restrict__ query,
                                                      const int *__restrict__ qperm, const float *__restrict__ ref,
                                                      const int *__restrict__ rperm, const float *__restrict__ boxes,
                                                      int *__restrict__ idx, float *__restrict__ dist, int vec_rows) {
    using C = WalkCfg<K>;
    constexpr int SUB = 4, QPW = 64 / SUB, RPL = PT / SUB, KS = C::KS, CL = C::CL, NET = C::NET, CHK = C::CHK, GAP = C::GAP, NQ = C::NQ;
    extern __shared__ float4 smem_f4[];
    const int lane = threadIdx.x;
    char *wbase = reinterpret_cast<char *>(smem_f4);
    float4 *tile = reinterpret_cast<float4 *>(wbase);
    int *tperm = reinterpret_cast<int *>(wbase + C::TILE_BYTES);
    // candidate list: slot s = 384 bytes, [distance of the 64 lanes, 4 bytes each | original index of the 64 lanes, 2 bytes each (the
    // walk covers n <= 65536)]; a lane's distance of slot s sits at byte s * 384 + lane * 4, its index in the 128 bytes behind the
    // distances: conflict-free whatever slots the lanes are at.  6 bytes per entry instead of 8 is what lets 32 slots per lane fit
    // beside 12 waves per CU.
    char *lbase = wbase + C::TILE_BYTES + C::PERM_BYTES;
    const int lane4 = lane * 4;
    auto l_dist = [&](int byte_off) -> float & { return *reinterpret_cast<float *>(lbase + byte_off); };
    // index halfword of the same slot: lanes 0..31 take the low halves of 32 words, lanes 32..63 the high halves (a 2-byte access is
    // served in two groups of 32 lanes: consecutive halfwords, two lanes per word, were a 2-way bank conflict on every access)
    const int idx_off = 256 + (lane & 31) * 4 + (lane >> 5) * 2 - lane * 4;
    auto l_index = [&](int byte_off) -> unsigned short & { return *reinterpret_cast<unsigned short *>(lbase + byte_off + idx_off); };
#ifdef MCP_KNN_DIAG
    unsigned napp_ = 0, cn_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_)::"memory");
#endif

    const int b = blockIdx.y;
    const int sub = lane % SUB;
    const int q0 = blockIdx.x * QPW;            // first query of the wave: always < q (grid is sized from q)
    const int qi = q0 + lane / SUB;
    const bool live = qi < q;
    // dead lanes replicate the wave's first query so they do not inflate the query box
    const float *qp = query + ((size_t)b * q + (live ? qi : q0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qn = mcp_sqnorm3(qx, qy, qz);
    ref += (size_t)b * n * 3;
    rperm += (size_t)b * n;
    boxes += (size_t)b * tiles * 6;

    const float bl0 = wave_minf(qx), bl1 = wave_minf(qy), bl2 = wave_minf(qz);
    const float bh0 = wave_maxf(qx), bh1 = wave_maxf(qy), bh2 = wave_maxf(qz);

    // lower bound of the squared distance to each tile box; lane l owns tiles l, l+64, ... (TPL per lane)
    float lb[TPL];
    float m2 = fmaxf(fmaxf(fabsf(bl0), fabsf(bh0)), fmaxf(fmaxf(fabsf(bl1), fabsf(bh1)), fmaxf(fabsf(bl2), fabsf(bh2))));
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
        const int t = lane + 64 * u;
        lb[u] = INFINITY;
        if (t < tiles) {
            const float *bx = boxes + t * 6;
            const float g0 = fmaxf(0.f, fmaxf(bx[0] - bh0, bl0 - bx[3]));
            const float g1 = fmaxf(0.f, fmaxf(bx[1] - bh1, bl1 - bx[4]));
            const float g2 = fmaxf(0.f, fmaxf(bx[2] - bh2, bl2 - bx[5]));
            lb[u] = g0 * g0 + g1 * g1 + g2 * g2;
            m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(bx[0]), fabsf(bx[3])), fmaxf(fmaxf(fabsf(bx[1]), fabsf(bx[4])), fmaxf(fabsf(bx[2]), fabsf(bx[5])))));
        }
    }
    m2 = wave_maxf(m2);
    // |computed - exact| of either distance form is below ~40 * 2^-24 * M^2 = 2.4e-6 M^2 (M = largest |coordinate|);
    // the bound arithmetic errs by a few ulp of the bound: 3e-5*M^2 absolute + 1e-6 relative covers both.
    const float slack_abs = 3e-5f * m2 * m2, slack_rel = 1e-6f;

    float ts[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) ts[i] = INFINITY;
    float tau = live ? INFINITY : -INFINITY;      // the query's threshold (same in its four lanes); dead lanes never append
    float taumax = INFINITY;                      // largest tau in the wave: what the next tile must beat
    int wp = lane4, dp = lane4;                   // byte offset of the lane's first free slot; of its first entry not yet merged into ts
    constexpr int SLOT = C::SLOT;

    // Threshold update: the (at most NQ) entries appended since the last one go into the lane's KS-list; tau = the largest
    // KS-th distance of the query's four lanes.
    auto update_tau = [&]() {
        WALK_COUNT(3, 1);
        do {  // NQ entries per lane and pass (two passes cover a whole tile's worth)
            float nv[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int o = dp + j * SLOT;
                const float v = l_dist(min(o, lane4 + (CL - 1) * SLOT));
                nv[j] = o < wp ? v : INFINITY;
            }
            merge_new<KS, NQ>(ts, nv);
            dp = min(dp + NQ * SLOT, wp);
        } while (__builtin_amdgcn_ballot_w64(dp < wp));
        float own = live ? ts[KS - 1] : -INFINITY;   // (a compiler-visible instruction between the raw min / max and the DPP reads)
        own = fmaxf(own, dpp_f<0xB1>(own));
        own = fmaxf(own, dpp_f<0x4E>(own));
        tau = own;
        taumax = mcp_unord(mcp_wave_max_u32(mcp_ord(tau)));
    };
    // Soft compaction: entries above tau leave the list (in place: the write position never passes the read position).
    auto compact_soft = [&]() {
        WALK_COUNT(4, 1);
        int w = lane4;
#pragma unroll
        for (int s0 = 0; s0 < CL; s0 += 16) {   // 16 entries at a time: reads of a block come after the writes of the one before (w <= s0)
            float ed[16];
            unsigned short ei[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                ed[s] = l_dist(lane4 + (s0 + s) * SLOT);
                ei[s] = l_index(lane4 + (s0 + s) * SLOT);
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bool keep = lane4 + (s0 + s) * SLOT < wp && ed[s] <= tau;
                l_dist(w) = ed[s];
                l_index(w) = ei[s];
                w += keep ? SLOT : 0;
            }
        }
        wp = w;
        dp = w;
    };
    // Hard compaction: the quad sorts the keys of its first NET slots per lane and keeps exactly the K smallest of them, KS per lane
    // (a key outside the K smallest of a subset is outside the K smallest of the whole); entries beyond slot NET move down behind
    // them.  tau becomes the K-th key's distance.  Every lane ends with at most KS + (CL - NET) entries.
    auto compact_hard = [&]() {
        WALK_COUNT(5, 1);
        mcp_key v[NET];
#pragma unroll
        for (int s = 0; s < NET; ++s)
            v[s] = lane4 + s * SLOT < wp ? mcp_make_key(l_dist(lane4 + s * SLOT), l_index(lane4 + s * SLOT)) : KEY_INF;
        quad_sort_low<NET>(v, sub);
        // rank r lives in lane r / NET, register r % NET; lane t takes ranks t*KS .. t*KS+KS-1
        constexpr int SRC_CTRL = (4 * KS > NET) ? 0x50 : 0x00;   // K = 32: lanes (0,0,1,1); K <= 16: every rank sits in lane 0
        const int off = (sub * KS) % NET;
        mcp_key mine[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) mine[i] = KEY_INF;
#pragma unroll
        for (int r = 0; r < NET; ++r) {
            const mcp_key g = dpp_key<SRC_CTRL>(v[r]);
#pragma unroll
            for (int i = 0; i < KS; ++i)
                if ((r - i) % KS == 0 && r - i >= 0 && off == r - i) mine[i] = g;
        }
        float rd[CL > NET ? CL - NET : 1];
        unsigned short ri[CL > NET ? CL - NET : 1];
        if constexpr (CL > NET) {
#pragma unroll
            for (int s = NET; s < CL; ++s) {
                rd[s - NET] = l_dist(lane4 + s * SLOT);
                ri[s - NET] = l_index(lane4 + s * SLOT);
            }
        }
        int w = lane4;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const bool have = !mcp_key_is_inf(mine[i]);
            l_dist(lane4 + i * SLOT) = mcp_key_dist(mine[i]);
            l_index(lane4 + i * SLOT) = (unsigned short)mcp_key_index(mine[i]);
            ts[i] = have ? mcp_key_dist(mine[i]) : INFINITY;
            w += have ? SLOT : 0;
        }
        if constexpr (CL > NET) {
            const int extra = max(wp - (lane4 + NET * SLOT), 0);   // bytes of entries that sat beyond slot NET
#pragma unroll
            for (int s = NET; s < CL; ++s) {
                l_dist(w + (s - NET) * SLOT) = rd[s - NET];        // w <= lane4 + KS * SLOT: the copies stay inside the list
                l_index(w + (s - NET) * SLOT) = ri[s - NET];
            }
            w += extra;
        }
        wp = w;
        dp = w;
        float own = live ? ts[KS - 1] : -INFINITY;
        own = fmaxf(own, dpp_f<0xB1>(own));
        own = fmaxf(own, dpp_f<0x4E>(own));
        tau = own;
        taumax = mcp_unord(mcp_wave_max_u32(mcp_ord(tau)));
    };
    // every GAP rows: room for the next GAP rows (a lane enters them with at most CL - GAP entries), thresholds from the new entries
    auto housekeeping = [&]() {
        const bool fresh = wp - dp >= C::UPD_MIN * SLOT, full = wp - lane4 > (CL - GAP) * SLOT;
        const unsigned long long any_full = __builtin_amdgcn_ballot_w64(full);
        if (__builtin_amdgcn_ballot_w64(fresh) | any_full) {
            WALK_STAMP(3);
            update_tau();
            WALK_STAMP(4);
            if (any_full) {
                compact_soft();
                while (__builtin_amdgcn_ballot_w64(wp - lane4 > (CL - GAP) * SLOT)) compact_hard();
                WALK_STAMP(5);
            }
        }
    };

    // Next unvisited tile by ascending bound, one wave reduction: the key packs the bound's ordered bits truncated to 22
    // (rounded DOWN, so the decoded value is still a lower bound) above the inverted 10-bit tile id.  Returns -1 when
    // every tile has been visited; marks the returned tile visited.
    constexpr uint32_t LBQ_INF = 0xFF800000u >> 10;
    auto next_tile = [&](float &bound) -> int {
        float mylb = lb[0];
        int myu = 0;
#pragma unroll
        for (int u = 1; u < TPL; ++u) {
            if (lb[u] < mylb) { mylb = lb[u]; myu = u; }
        }
        const uint32_t myt = (uint32_t)(lane + 64 * myu);
        const uint32_t wkey = mcp_wave_max_u32(((~(mcp_ord(mylb) >> 10)) << 10) | (~myt & 0x3FFu));
        const uint32_t lbq = ~(wkey >> 10) & 0x3FFFFFu;
        if (lbq == LBQ_INF) return -1;
        const int t = (int)(~wkey & 0x3FFu);
        bound = mcp_unord(lbq << 10);
        if ((int)myt == t) {
#pragma unroll
            for (int u = 0; u < TPL; ++u)
                if (u == myu) lb[u] = INFINITY;
        }
        return t;
    };
    // a tile's coordinates and original indices travel through registers, so the NEXT tile's global loads are in
    // flight while the current tile is scanned
    struct TileRegs { float x, y, z; int pi; };
    static_assert(PT == 64, "one reference per lane and tile");
    auto fetch = [&](int t, TileRegs &g) {
        const int gi = t * PT + lane;
        const bool ok = gi < n;
        const int gg = ok ? gi : 0;
        const float x = ref[(size_t)gg * 3 + 0], y = ref[(size_t)gg * 3 + 1], z = ref[(size_t)gg * 3 + 2];
        g.pi = ok ? rperm[gg] : 0;
        g.x = x;
        g.y = y;
        g.z = ok ? z : NAN;  // marks padding
    };
    // per-query filter: a tile that passes the box test is scanned only if, for at least one query, the squared distance from the
    // query POINT to the tile box is within that query's own threshold (same rounding slack as the box test)
    auto wanted = [&](int tt) -> bool {
        const float *bx = boxes + tt * 6;  // wave-uniform address
        const float g0 = fmaxf(0.f, fmaxf(bx[0] - qx, qx - bx[3]));
        const float g1 = fmaxf(0.f, fmaxf(bx[1] - qy, qy - bx[4]));
        const float g2 = fmaxf(0.f, fmaxf(bx[2] - qz, qz - bx[5]));
        const float g = g0 * g0 + g1 * g1 + g2 * g2;
        return __builtin_amdgcn_ballot_w64(g <= tau + (tau * slack_rel + slack_abs)) != 0;  // dead lanes: tau = -inf
    };
    // The same filter for every unvisited tile at once, lane = tile owner: a tile no query of the wave can use NOW can never be used
    // later (thresholds only shrink), so it is struck off for good.  576 instructions at 128 tiles -- worth it only when the sequential
    // walk keeps meeting useless tiles: queries far from a compact reference cloud see every tile box at about the same distance,
    // and the box order then says nothing (the refined cloud of the untrained network: 128 tiles examined, 15 scanned).
    auto prune_tiles = [&]() {
        const float lim = tau + (tau * slack_rel + slack_abs);   // dead lanes: -inf
        float bx[TPL][6];
        bool need[TPL];
#pragma unroll
        for (int u = 0; u < TPL; ++u) {
            const int tt = lane + 64 * u;
            const float *p = boxes + (tt < tiles ? tt : 0) * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) bx[u][c] = p[c];
            need[u] = false;
        }
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            auto from_quad = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), i * SUB)); };
            const float ax = from_quad(qx), ay = from_quad(qy), az = from_quad(qz), al = from_quad(lim);
#pragma unroll
            for (int u = 0; u < TPL; ++u) {
                const float g0 = fmaxf(0.f, fmaxf(bx[u][0] - ax, ax - bx[u][3]));
                const float g1 = fmaxf(0.f, fmaxf(bx[u][1] - ay, ay - bx[u][4]));
                const float g2 = fmaxf(0.f, fmaxf(bx[u][2] - az, az - bx[u][5]));
                need[u] = need[u] || (g0 * g0 + g1 * g1 + g2 * g2 <= al);
            }
        }
#pragma unroll
        for (int u = 0; u < TPL; ++u)
            if (!need[u]) lb[u] = INFINITY;
    };
    // next tile in bound order that passes both tests; -1 ends the walk (every later tile has a larger box bound)
    auto next_wanted = [&]() -> int {
        int rejected = 0;
        for (;;) {
            float bd = 0.f;
            const int tt = next_tile(bd);
            WALK_COUNT(2, 1);
            if (tt < 0 || !(bd <= taumax + (taumax * slack_rel + slack_abs))) return -1;
            if (wanted(tt)) return tt;
            if (++rejected == C::PRUNE_AFTER) prune_tiles();
        }
    };

    float bound = 0.f;
    int t = next_tile(bound);  // the first tile is always visited (tau = +inf)
    TileRegs cur, nxt;
    if (t >= 0) fetch(t, cur);
    WALK_COUNT(0, 1);
    WALK_STAMP(0);
    while (t >= 0) {
        WALK_COUNT(1, 1);
        int t2 = next_wanted();  // chosen with the thresholds as they stand BEFORE this tile's scan ...
        if (t2 >= 0) fetch(t2, nxt);
        WALK_STAMP(1);
        // stage the tile: coordinates + squared norm in reference order, original indices grouped per sub-lane.  Padding rows get
        // a NaN distance in both forms: they never pass `d <= tau`.
        __builtin_amdgcn_wave_barrier();
        {
            const bool pad = cur.z != cur.z;
            float4 v = make_float4(pad ? NAN : cur.x, cur.y, pad ? 0.f : cur.z, 0.f);
            v.w = pad ? NAN : mcp_sqnorm3(v.x, v.y, v.z);
            tile[lane] = v;
            tperm[(lane % SUB) * RPL + lane / SUB] = cur.pi;
        }
        __builtin_amdgcn_wave_barrier();
        WALK_STAMP(2);
        // lane scans references sub, sub+SUB, ... ; j-th reference of the lane is tile[j*SUB + sub].  Two register sets take turns
        // (the next chunk's LDS reads are in flight while this one is scanned, and nothing is copied)
        const int4 *myperm = reinterpret_cast<const int4 *>(tperm + sub * RPL);
        auto load4 = [&](int j0, float4 (&rc)[CHK], int4 &pc) {
#pragma unroll
            for (int u = 0; u < CHK; ++u) rc[u] = tile[(j0 + u) * SUB + sub];
            pc = myperm[j0 >> 2];
        };
        auto scan4 = [&](const float4 (&rc)[CHK], const int4 pc) {
            float d[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u) d[u] = pair_dist<MODE>(qx, qy, qz, qn, rc[u]);
            const int pidx[CHK] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                l_dist(wp) = d[u];                       // unconditional: the entry counts only if the write position advances
                l_index(wp) = (unsigned short)pidx[u];
                wp += d[u] <= tau ? SLOT : 0;
#ifdef MCP_KNN_DIAG
                napp_ += d[u] <= tau ? 1 : 0;
#endif
            }
        };
        static_assert(RPL % (2 * CHK) == 0 && GAP == 2 * CHK, "chunks are scanned in pairs, one capacity check per pair");
        float4 ra[CHK], rb[CHK];
        int4 pa, pb;
        load4(0, ra, pa);
#pragma unroll 1
        for (int j0 = 0; j0 < RPL; j0 += 2 * CHK) {
            load4(j0 + CHK, rb, pb);
            scan4(ra, pa);
            load4(j0 + 2 * CHK < RPL ? j0 + 2 * CHK : 0, ra, pa);
            scan4(rb, pb);
            housekeeping();
        }
        WALK_STAMP(3);
        // tighten tau before the next pruning decision when some lane has a few new entries (a stale, larger tau is still a
        // valid bound, it just prunes a little less)
        if (__builtin_amdgcn_ballot_w64(wp - dp >= C::UPD_END * SLOT)) {
            update_tau();
            WALK_STAMP(4);
        }
        // ... and re-examined with the tightened ones: the prefetched tile may have become useless (its loads are then
        // wasted and the next candidate is fetched without overlap)
        if (t2 >= 0 && !wanted(t2)) {
            t2 = next_wanted();
            if (t2 >= 0) fetch(t2, nxt);
        }
        t = t2;
        cur = nxt;
        WALK_STAMP(1);
    }
    // the survivors, sorted once on the full (distance, index) keys: NET slots per lane go through the network, so lists longer
    // than that (rule-b thresholds leave about K * 1.4 survivors per query, unevenly dealt) shed their entries above tau first, and
    // the rare ones that are still longer take hard compactions
    if (CL > NET && __builtin_amdgcn_ballot_w64(wp - lane4 > NET * SLOT)) {
        update_tau();
        compact_soft();
        while (__builtin_amdgcn_ballot_w64(wp - lane4 > NET * SLOT)) compact_hard();
    }
    mcp_key v[NET];
#pragma unroll
    for (int s = 0; s < NET; ++s)
        v[s] = lane4 + s * SLOT < wp ? mcp_make_key(l_dist(lane4 + s * SLOT), l_index(lane4 + s * SLOT)) : KEY_INF;
#ifdef MCP_KNN_DIAG
    atomicAdd(&g_walk_diag[6], (unsigned long long)napp_);
    atomicAdd(&g_walk_diag[7], (unsigned long long)((wp - lane4) / SLOT));
#endif
    quad_sort_low<NET>(v, sub);
    if (n < K) {
        // fewer references than list entries: the missing ranks repeat the last valid one (as mcp_store_list does)
        mcp_key last = __hiloint2double((int)0xFFF00000, 0);   // -inf: below every key
#pragma unroll
        for (int s = 0; s < NET; ++s) last = mcp_key_is_inf(v[s]) ? last : v[s];
        if (sub >= 2) last = __hiloint2double((int)0xFFF00000, 0);
        const mcp_key o = dpp_key<0xB1>(last);
        last = fmax(last, o);
#pragma unroll
        for (int s = 0; s < NET; ++s) v[s] = mcp_key_is_inf(v[s]) ? last : v[s];
    }
    if (live && sub * NET < kout) {
        const int row = qperm ? qperm[(size_t)b * q + qi] : qi;
        int *oi = idx + ((size_t)b * q + row) * kout + sub * NET;
        float *od = dist ? dist + ((size_t)b * q + row) * kout + sub * NET : nullptr;
        if (vec_rows && kout >= (sub + 1) * NET) {   // whole 16-byte pieces (vec_rows: kout % 4 == 0 and 16-byte aligned outputs)
#pragma unroll
            for (int s = 0; s < NET; s += 4) {
                reinterpret_cast<int4 *>(oi)[s >> 2] = make_int4((int)mcp_key_index(v[s]), (int)mcp_key_index(v[s + 1]), (int)mcp_key_index(v[s + 2]), (int)mcp_key_index(v[s + 3]));
                if (od) reinterpret_cast<float4 *>(od)[s >> 2] = make_float4(mcp_key_dist(v[s]), mcp_key_dist(v[s + 1]), mcp_key_dist(v[s + 2]), mcp_key_dist(v[s + 3]));
            }
        } else {
#pragma unroll
            for (int s = 0; s < NET; ++s) {
                if (sub * NET + s < kout) {
                    oi[s] = (int)mcp_key_index(v[s]);
                    if (od) od[s] = mcp_key_dist(v[s]);
                }
            }
        }
    }
    WALK_STAMP(6);
#ifdef MCP_KNN_DIAG
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&g_walk_diag[i], (unsigned long long)cn_[i]);
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&g_walk_diag[8 + i], ph_[i]);
    }
#endif
}

template <int K, int MODE, int TPL>
int launch_walk_tpl(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                    const float *boxes, int *idx, float *dist, hipStream_t s) {
    const size_t lds = (size_t)WalkCfg<K>::WAVE_BYTES;
    auto kern = knn_walk_kernel<K, MODE, TPL>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {  // lets the CU's whole 160 KB LDS count towards residency (default budget: 64 KB)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    const int vec_rows = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0 && (!dist || (reinterpret_cast<uintptr_t>(dist) & 15) == 0);
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 16), b), dim3(64), lds, s, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, vec_rows);
    return mcp_launch_status();
}

template <int K, int MODE>
int launch_walk(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                const float *boxes, int *idx, float *dist, hipStream_t s) {
    // tile bounds held per lane: sized to the cloud so the per-visit argmin stays short
    if (tiles <= 64) return launch_walk_tpl<K, MODE, 1>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 128) return launch_walk_tpl<K, MODE, 2>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 256) return launch_walk_tpl<K, MODE, 4>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    return launch_walk_tpl<K, MODE, MAX_TPL>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

template <int MODE>
int launch_walk_k(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                  const float *boxes, int *idx, float *dist, hipStream_t s) {
#if defined(MCP_AB) || defined(MCP_KNN_DIAG)   // the product sends K <= 16 to round 3's kernel (below)
    if (k <= 4) return launch_walk<4, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (k <= 16) return launch_walk<16, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
#endif
    return launch_walk<32, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

// ---- round 3's search kernel (per-lane sorted K-lists + threshold queue): the searches with K <= 16 ----------------------------
// Measured on MI355X (tools/knn_ab.py): with short lists the register networks are small and this kernel is the faster one
// (48 x 2048^2, K = 16: 71 us against 92; 24 x 8192 -> 2048, K = 3: 64 against 72), at K = 32 the walk kernel above is.
template <int K>
struct PrunedLds {
#ifndef MCP_PRUNED_QS
#define MCP_PRUNED_QS 16
#endif
    static constexpr int QS = MCP_PRUNED_QS < K ? MCP_PRUNED_QS : K;  // the merge network needs QS <= K
    static constexpr int TILE_BYTES = PT * 16;      // float4 (x,y,z,|r|^2)
    static constexpr int PERM_BYTES = PT * 4;       // original indices, stored [sub][PT/SUB]
    static constexpr int QUEUE_BYTES = QS * 64 * 8;
    static constexpr int WAVE_BYTES = TILE_BYTES + PERM_BYTES + QUEUE_BYTES;
};

// min over the SUB adjacent lanes that share a query
template <int SUB>
__device__ __forceinline__ float sub_min(float v) {
    if (SUB >= 2) v = fminf(v, __uint_as_float(mcp_dpp<0xB1>(__float_as_uint(v))));  // quad_perm [1,0,3,2]
    if (SUB >= 4) v = fminf(v, __uint_as_float(mcp_dpp<0x4E>(__float_as_uint(v))));  // quad_perm [2,3,0,1]
    return v;
}

template <int SUB>
__device__ __forceinline__ float sub_max(float v) {
    if (SUB >= 2) v = fmaxf(v, __uint_as_float(mcp_dpp<0xB1>(__float_as_uint(v))));
    if (SUB >= 4) v = fmaxf(v, __uint_as_float(mcp_dpp<0x4E>(__float_as_uint(v))));
    return v;
}

// SUB lanes cooperate on one query (64/SUB queries per wave): lane sub = lane % SUB scans references
// r = sub (mod SUB) of every visited tile into its own K-list; the push threshold is the minimum of the
// SUB K-th distances (any one list already holds K references below it, so nothing above can reach the
// final K); the SUB lists are merged through DPP exchanges at the end.  More, shorter waves with smaller
// query boxes: better pruning, latency and load balance than one lane per query.
#ifdef MCP_KNN_DIAG
// diagnostic build only (never in the product library), round 3's kernel: [0] waves, [1] tiles visited, [2] flushes, [3] pushes (all lanes)
__device__ unsigned long long g_knn_diag[4];
__device__ unsigned long long g_knn_phase[8];  // cycles: 0 setup 1 walk 2 staging 3 scan 4 flush 5 final
#define KNN_COUNT(slot, v) do { ocn_[slot] += (unsigned)(v); } while (0)
#define OLD_STAMP(slot)                                                                \
    do {                                                                               \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        oph_[slot] += t_ - ot_prev_;                                                   \
        ot_prev_ = t_;                                                                 \
    } while (0)
#else
#define KNN_COUNT(slot, v)
#define OLD_STAMP(slot)
#endif

template <int K, int MODE, int SUB, int TPL>
__global__ __launch_bounds__(64) void knn_pruned_kernel(int q, int n, int tiles, int kout, const float *__restrict__ query,
                                                        const int *__restrict__ qperm, const float *__restrict__ ref,
                                                        const int *__restrict__ rperm, const float *__restrict__ boxes,
                                                        int *__restrict__ idx, float *__restrict__ dist) {
    using L = PrunedLds<K>;
    constexpr int QS = L::QS, CHK = 4, QPW = 64 / SUB, RPL = PT / SUB;  // queries per wave, references per lane per tile
    extern __shared__ float4 smem_f4[];
    const int lane = threadIdx.x;
#ifdef MCP_KNN_DIAG
    unsigned ocn_[4] = {0, 0, 0, 0}, opush_ = 0;
    unsigned long long oph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ot_prev_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ot_prev_)::"memory");
#endif
    char *wbase = reinterpret_cast<char *>(smem_f4);
    float4 *tile = reinterpret_cast<float4 *>(wbase);
    int *tperm = reinterpret_cast<int *>(wbase + L::TILE_BYTES);
    uint2(*queue)[64] = reinterpret_cast<uint2(*)[64]>(wbase + L::TILE_BYTES + L::PERM_BYTES);

    const int b = blockIdx.y;
    const int sub = lane % SUB;
    const int q0 = blockIdx.x * QPW;            // first query of the wave: always < q (grid is sized from q)
    const int qi = q0 + lane / SUB;
    const bool live = qi < q;
    // dead lanes replicate the wave's first query so they do not inflate the query box
    const float *qp = query + ((size_t)b * q + (live ? qi : q0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qn = mcp_sqnorm3(qx, qy, qz);
    ref += (size_t)b * n * 3;
    rperm += (size_t)b * n;
    boxes += (size_t)b * tiles * 6;

    const float bl0 = wave_minf(qx), bl1 = wave_minf(qy), bl2 = wave_minf(qz);
    const float bh0 = wave_maxf(qx), bh1 = wave_maxf(qy), bh2 = wave_maxf(qz);

    // lower bound of the squared distance to each tile box; lane l owns tiles l, l+64, ... (TPL per lane)
    float lb[TPL];
    float m2 = fmaxf(fmaxf(fabsf(bl0), fabsf(bh0)), fmaxf(fmaxf(fabsf(bl1), fabsf(bh1)), fmaxf(fabsf(bl2), fabsf(bh2))));
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
        const int t = lane + 64 * u;
        lb[u] = INFINITY;
        if (t < tiles) {
            const float *bx = boxes + t * 6;
            const float g0 = fmaxf(0.f, fmaxf(bx[0] - bh0, bl0 - bx[3]));
            const float g1 = fmaxf(0.f, fmaxf(bx[1] - bh1, bl1 - bx[4]));
            const float g2 = fmaxf(0.f, fmaxf(bx[2] - bh2, bl2 - bx[5]));
            lb[u] = g0 * g0 + g1 * g1 + g2 * g2;
            m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(bx[0]), fabsf(bx[3])), fmaxf(fmaxf(fabsf(bx[1]), fabsf(bx[4])), fmaxf(fabsf(bx[2]), fabsf(bx[5])))));
        }
    }
    m2 = wave_maxf(m2);
    // |computed - exact| of either distance form is below ~40 * 2^-24 * M^2 = 2.4e-6 M^2 (M = largest |coordinate|);
    // the bound arithmetic errs by a few ulp of the bound: 3e-5*M^2 absolute + 1e-6 relative covers both.
    const float slack_abs = 3e-5f * m2 * m2, slack_rel = 1e-6f;

    u64 a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = KEY_INF;
    float tau_own = live ? INFINITY : -INFINITY;  // dead lanes never push and never hold the wave back
    float tau = tau_own;                          // shared push threshold of the query's SUB lanes
    float taumax = INFINITY;                      // largest tau in the wave: what the next tile must beat
    int cnt = 0;
    auto flush = [&]() {
        KNN_COUNT(2, 1);
#ifdef MCP_KNN_DIAG
        opush_ += (unsigned)cnt;
#endif
        OLD_STAMP(3);
        mcp_flush_queue<K, QS>(a, queue, lane, cnt);
        if (live) tau_own = mcp_tau_of(a[K - 1]);
        // Push threshold shared by the query's SUB lanes.  Two valid upper bounds of the K-th distance of the union:
        // (a) any single lane's K-th entry; (b) the largest of the lanes' (K/SUB)-th entries -- SUB * K/SUB = K
        // candidates lie at or below it.  (b) is the tight one: a lane's own K-th entry is about the SUB*K-th overall.
        tau = sub_min<SUB>(tau_own);
        if (SUB > 1) tau = fminf(tau, sub_max<SUB>(live ? mcp_tau_of(a[K / SUB - 1]) : -INFINITY));
        taumax = mcp_unord(mcp_wave_max_u32(mcp_ord(tau)));
        cnt = 0;
        OLD_STAMP(4);
    };

    // Next unvisited tile by ascending bound, one wave reduction: the key packs the bound's ordered bits truncated to 22
    // (rounded DOWN, so the decoded value is still a lower bound) above the inverted 10-bit tile id.  Returns -1 when
    // every tile has been visited; marks the returned tile visited.
    constexpr uint32_t LBQ_INF = 0xFF800000u >> 10;
    auto next_tile = [&](float &bound) -> int {
        float mylb = lb[0];
        int myu = 0;
#pragma unroll
        for (int u = 1; u < TPL; ++u) {
            if (lb[u] < mylb) { mylb = lb[u]; myu = u; }
        }
        const uint32_t myt = (uint32_t)(lane + 64 * myu);
        const uint32_t wkey = mcp_wave_max_u32(((~(mcp_ord(mylb) >> 10)) << 10) | (~myt & 0x3FFu));
        const uint32_t lbq = ~(wkey >> 10) & 0x3FFFFFu;
        if (lbq == LBQ_INF) return -1;
        const int t = (int)(~wkey & 0x3FFu);
        bound = mcp_unord(lbq << 10);
        if ((int)myt == t) {
#pragma unroll
            for (int u = 0; u < TPL; ++u)
                if (u == myu) lb[u] = INFINITY;
        }
        return t;
    };
    // a tile's coordinates and original indices travel through registers, so the NEXT tile's global loads are in
    // flight while the current tile is scanned (PT / 64 references per lane)
    struct TileRegs { float x[PT / 64], y[PT / 64], z[PT / 64]; int pi[PT / 64]; };
    auto fetch = [&](int t, TileRegs &g) {
#pragma unroll
        for (int u = 0; u < PT / 64; ++u) {
            const int gi = t * PT + lane + 64 * u;
            const bool ok = gi < n;
            const int gg = ok ? gi : 0;
            const float x = ref[(size_t)gg * 3 + 0], y = ref[(size_t)gg * 3 + 1], z = ref[(size_t)gg * 3 + 2];
            g.pi[u] = ok ? rperm[gg] : 0;
            // padding: distance evaluates to +inf in both forms
            g.x[u] = ok ? x : (MODE == MCP_DIST_EXPANSION ? 0.f : INFINITY);
            g.y[u] = ok ? y : 0.f;
            g.z[u] = ok ? z : NAN;  // marks padding for the norm below
        }
    };

    // Second, per-query filter.  Tiles are ordered (and the walk is ended) by their distance to the wave's QUERY BOX, which
    // under-estimates badly when the 16 queries of a wave are spread out or far from the references.  A tile that passes
    // the box test is scanned only if, for at least one query, the squared distance from the query POINT to the tile box
    // is within that query's own threshold (same rounding slack as the box test).
    auto wanted = [&](int tt) -> bool {
        const float *bx = boxes + tt * 6;  // wave-uniform address
        const float g0 = fmaxf(0.f, fmaxf(bx[0] - qx, qx - bx[3]));
        const float g1 = fmaxf(0.f, fmaxf(bx[1] - qy, qy - bx[4]));
        const float g2 = fmaxf(0.f, fmaxf(bx[2] - qz, qz - bx[5]));
        const float g = g0 * g0 + g1 * g1 + g2 * g2;
        return __builtin_amdgcn_ballot_w64(g <= tau + (tau * slack_rel + slack_abs)) != 0;  // dead lanes: tau = -inf
    };
    // next tile in bound order that passes both tests; -1 ends the walk (every later tile has a larger box bound)
    auto next_wanted = [&]() -> int {
        for (;;) {
            float bd = 0.f;
            const int tt = next_tile(bd);
            if (tt < 0 || !(bd <= taumax + (taumax * slack_rel + slack_abs))) return -1;
            if (wanted(tt)) return tt;
        }
    };

    float bound = 0.f;
    int t = next_tile(bound);  // the first tile is always visited (tau = +inf)
    TileRegs cur, nxt;
    if (t >= 0) fetch(t, cur);
    KNN_COUNT(0, 1);
    OLD_STAMP(0);
    while (t >= 0) {
        KNN_COUNT(1, 1);
        int t2 = next_wanted();  // chosen with the thresholds as they stand BEFORE this tile's scan ...
        if (t2 >= 0) fetch(t2, nxt);
        OLD_STAMP(1);
        // stage the tile: coordinates + squared norm in reference order, original indices grouped per sub-lane
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < PT / 64; ++u) {
            const int r = lane + 64 * u;
            const bool pad = cur.z[u] != cur.z[u];
            float4 v = make_float4(cur.x[u], cur.y[u], pad ? 0.f : cur.z[u], 0.f);
            v.w = pad ? (MODE == MCP_DIST_EXPANSION ? INFINITY : 0.f) : mcp_sqnorm3(v.x, v.y, v.z);
            tile[r] = v;
            tperm[(r % SUB) * RPL + r / SUB] = cur.pi[u];
        }
        __builtin_amdgcn_wave_barrier();
        OLD_STAMP(2);
        // lane scans references sub, sub+SUB, ... ; j-th reference of the lane is tile[j*SUB + sub]
        const int4 *myperm = reinterpret_cast<const int4 *>(tperm + sub * RPL);
        float4 rc[CHK];
        int4 pc = myperm[0];
#pragma unroll
        for (int u = 0; u < CHK; ++u) rc[u] = tile[u * SUB + sub];
        for (int j0 = 0; j0 < RPL; j0 += CHK) {
            float4 rn[CHK];
            const int jn = j0 + CHK < RPL ? j0 + CHK : j0;
#pragma unroll
            for (int u = 0; u < CHK; ++u) rn[u] = tile[(jn + u) * SUB + sub];
            const int4 pn = myperm[jn >> 2];
            float d[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u) d[u] = pair_dist<MODE>(qx, qy, qz, qn, rc[u]);
            const int pidx[CHK] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                if (d[u] <= tau) {
                    queue[cnt][lane] = make_uint2(__float_as_uint(d[u]), (uint32_t)pidx[u]);
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QS - CHK)) flush();
#pragma unroll
            for (int u = 0; u < CHK; ++u) rc[u] = rn[u];
            pc = pn;
        }
        OLD_STAMP(3);
        // tighten tau before the next pruning decision, but only when a queue is at least half full: a stale
        // (larger) tau is still a valid bound, it just prunes a little less
        if (__builtin_amdgcn_ballot_w64(cnt >= QS / 2)) flush();
        // ... and re-examined with the tightened ones: the prefetched tile may have become useless (its loads are then
        // wasted and the next candidate is fetched without overlap)
        if (t2 >= 0 && !wanted(t2)) {
            t2 = next_wanted();
            if (t2 >= 0) fetch(t2, nxt);
        }
        t = t2;
        cur = nxt;
        OLD_STAMP(1);
    }
    flush();
    // merge the SUB lists of each query (after each round both partners hold the union's K smallest)
    if (SUB >= 2) {
        u64 o[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
            o[j] = mcp_key_words(mcp_dpp<0xB1>(mcp_key_hi(a[j])), mcp_dpp<0xB1>(mcp_key_lo(a[j])));
        mcp_merge_sorted<K, K>(a, o);
    }
    if (SUB >= 4) {
        u64 o[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
            o[j] = mcp_key_words(mcp_dpp<0x4E>(mcp_key_hi(a[j])), mcp_dpp<0x4E>(mcp_key_lo(a[j])));
        mcp_merge_sorted<K, K>(a, o);
    }
    if (live && sub == 0) {
        const int row = qperm ? qperm[(size_t)b * q + qi] : qi;
        int *oi = idx + ((size_t)b * q + row) * kout;
        float *od = dist ? dist + ((size_t)b * q + row) * kout : nullptr;
        mcp_store_list<K>(a, kout, oi, od);
    }
    OLD_STAMP(5);
#ifdef MCP_KNN_DIAG
    atomicAdd(&g_knn_diag[3], (unsigned long long)opush_);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) atomicAdd(&g_knn_diag[i], (unsigned long long)ocn_[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&g_knn_phase[i], oph_[i]);
    }
#endif
}

template <int K, int MODE, int SUB, int TPL>
int launch_pruned_tpl(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                      const float *boxes, int *idx, float *dist, hipStream_t s) {
    const size_t lds = (size_t)PrunedLds<K>::WAVE_BYTES;
    auto kern = knn_pruned_kernel<K, MODE, SUB, TPL>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {  // lets the CU's whole 160 KB LDS count towards residency (default budget: 64 KB)
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 64 / SUB), b), dim3(64), lds, s, q, n, tiles, k, query, qperm,
                       ref, rperm, boxes, idx, dist);
    return mcp_launch_status();
}

template <int K, int MODE, int SUB>
int launch_pruned_sub(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                      const float *boxes, int *idx, float *dist, hipStream_t s) {
    // tile bounds held per lane: sized to the cloud so the per-visit argmin stays short
    if (tiles <= 64) return launch_pruned_tpl<K, MODE, SUB, 1>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 128) return launch_pruned_tpl<K, MODE, SUB, 2>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 256) return launch_pruned_tpl<K, MODE, SUB, 4>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    return launch_pruned_tpl<K, MODE, SUB, MAX_TPL>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

template <int K, int MODE>
int launch_pruned(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                  const float *boxes, int *idx, float *dist, hipStream_t s) {
    // 4 lanes per query (1 and 2 were measured slower at every shape of the pipeline)
    return launch_pruned_sub<K, MODE, 4>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

template <int MODE>
int launch_pruned_k(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                    const float *boxes, int *idx, float *dist, hipStream_t s) {
    if (k <= 4) return launch_pruned<4, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (k <= 16) return launch_pruned<16, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
#if defined(MCP_AB) || defined(MCP_KNN_DIAG)
    return launch_pruned<32, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);   // A/B builds only
#else
    return MCP_ERR_UNSUPPORTED;
#endif
}



}  // namespace

#ifdef MCP_KNN_DIAG
// out: 4 counters of round 3's kernel, 16 of the walk kernel (see g_walk_diag), 8 phase totals of round 3's kernel; all are reset
extern "C" __attribute__((visibility("default"))) int mcp_knn_diag_read(unsigned long long *out28) {
    hipError_t e = hipMemcpyFromSymbol(out28, HIP_SYMBOL(g_knn_diag), sizeof(unsigned long long) * 4);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out28 + 4, HIP_SYMBOL(g_walk_diag), sizeof(unsigned long long) * 16);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out28 + 20, HIP_SYMBOL(g_knn_phase), sizeof(unsigned long long) * 8);
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_diag), z, sizeof(unsigned long long) * 4);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_diag), z, sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_phase), z, sizeof(unsigned long long) * 8);
    return (int)e;
}
#endif
#if defined(MCP_AB) || defined(MCP_KNN_DIAG)
// A/B builds only: 0 = the product's choice (round 3's kernel for K <= 16, the walk kernel above), 1 = round 3's kernel for every K,
// 2 = the walk kernel for every K
static int g_knn_use_old = 0;
extern "C" __attribute__((visibility("default"))) void mcp_knn_pruned_use_old(int on) { g_knn_use_old = on; }
// resident workgroups per CU the runtime reports for the K = 32 kernels (walk, round 3)
extern "C" __attribute__((visibility("default"))) int mcp_knn_occupancy(int *walk, int *old) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(walk, knn_walk_kernel<32, MCP_DIST_EXPANSION, 2>, 64, WalkCfg<32>::WAVE_BYTES);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(old, knn_pruned_kernel<32, MCP_DIST_EXPANSION, 4, 2>, 64, PrunedLds<32>::WAVE_BYTES);
    return (int)e;
}
#endif

MCP_EXPORT int mcp_knn_tile_size(void) { return PT; }

MCP_EXPORT int mcp_morton_codes(int b, int n, const float *xyz, const float *box, int *codes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && box && codes);
    hipLaunchKernelGGL(morton_kernel, dim3(mcp_divup(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, xyz, box, codes);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_tile_boxes(int b, int n, const float *sorted_xyz, float *boxes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && sorted_xyz && boxes);
    const int tiles = (n + PT - 1) / PT;
    hipLaunchKernelGGL(tile_box_kernel, dim3(tiles, b), dim3(64), 0, (hipStream_t)stream, n, tiles, sorted_xyz, boxes);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_build_cloud(int b, int n, const float *xyz, float *sorted_xyz, int *perm, float *boxes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && sorted_xyz && perm && boxes);
    if (n > 16384) return MCP_ERR_UNSUPPORTED;  // larger clouds: mcp_morton_codes + an external sort + mcp_tile_boxes
    const int tiles = (n + PT - 1) / PT;
    hipStream_t s = (hipStream_t)stream;
    if (n <= BT) return launch_build_cloud<1>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 2 * BT) return launch_build_cloud<2>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 4 * BT) return launch_build_cloud<4>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 8 * BT) return launch_build_cloud<8>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    return launch_build_cloud<16>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
}

MCP_EXPORT int mcp_knn_pruned(int b, int q, int n, int k, int dist_form, const float *query_sorted, const int *qperm,
                              const float *ref_sorted, const int *rperm, const float *boxes, int *idx, float *dist,
                              mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && q > 0 && n > 0 && k > 0 && query_sorted && ref_sorted && rperm && boxes && idx);
    MCP_CHECK_ARGS(dist_form == MCP_DIST_EXPANSION || dist_form == MCP_DIST_DIRECT);
    const int tiles = (n + PT - 1) / PT;
    if (k > 32 || tiles > 64 * MAX_TPL) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    mcp_prof_begin(MCP_KERNEL_KNN, s);
    int rc;
    bool old = k <= 16;
#if defined(MCP_AB) || defined(MCP_KNN_DIAG)
    old = g_knn_use_old == 1 || (g_knn_use_old == 0 && old);
#endif
    if (old) {
        rc = dist_form == MCP_DIST_EXPANSION ? launch_pruned_k<MCP_DIST_EXPANSION>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s)
                                             : launch_pruned_k<MCP_DIST_DIRECT>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s);
        mcp_prof_end(MCP_KERNEL_KNN, s);
        return rc;
    }
    if (dist_form == MCP_DIST_EXPANSION)
        rc = launch_walk_k<MCP_DIST_EXPANSION>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s);
    else
        rc = launch_walk_k<MCP_DIST_DIRECT>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s);
    mcp_prof_end(MCP_KERNEL_KNN, s);
    return rc;
}
