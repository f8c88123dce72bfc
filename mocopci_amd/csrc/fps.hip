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
#include <stdlib.h>

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_scan.hpp>

#include "common.h"

// workgroup shape for 4 / 8 points per reference thread (T = 1024 >> J, P points per physical thread)
#ifndef MCP_FPS_J8
#define MCP_FPS_J8 1
#endif
#ifndef MCP_FPS_J4
#define MCP_FPS_J4 1
#endif
#define MCP_FPS_T8 (1024 >> MCP_FPS_J8)
#define MCP_FPS_P8 (8 << MCP_FPS_J8)
#define MCP_FPS_T4 (1024 >> MCP_FPS_J4)
#define MCP_FPS_P4 (4 << MCP_FPS_J4)

namespace {

typedef mcp_f2 f2;  // scalar pair: see common.h (no packed-fp32 instructions)
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return mcp_f2_fma(a, b, c); }

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

// T threads, P points per thread (k = tid + T*p).  The reference block has bs = 2^L threads; T = bs >> J.
// J == 0: a physical thread is exactly one reference thread and its strict-'>' scan returns the first
// maximum in ascending p.  J == 1 (T = bs/2): a thread holds two reference threads (even p -> tid,
// odd p -> tid + T); bitrev_L(tid + T) = bitrev_L(tid) + 1, so among equal values the even-p points win,
// each group in ascending p.  GENERIC (n < 64): every compare uses the full (ord, ~sec) pair.
#ifdef MCP_FPS_DIAG
// diagnostic build only: per-phase shader-cycle totals of wave 0 (never compiled into the product library)
__device__ unsigned long long g_fps_diag[8];
#define FPS_STAMP(slot)                                                                          \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_fps_diag[slot] += t_ - t_prev;                 \
        t_prev = t_;                                                                             \
    } while (0)
#else
#define FPS_STAMP(slot)
#endif

// Optional last step of every sampling kernel: the coordinates of the selected points, (m,3) per batch element, so that the
// furthest_point_sample + index_points_gather pair of the callers (mocopci.py:1378-1379) is one launch.  The indices were
// written by this workgroup; the barrier (workgroup-scope release / acquire) makes them visible to all of its threads.
template <int T>
__device__ __forceinline__ void fps_emit_points(const float *__restrict__ xyz, const int *idxs, float *__restrict__ pts, int m, int tid) {
    if (!pts) return;
    __syncthreads();
    for (int i = tid; i < m * 3; i += T) {
        const int j = i / 3;
        pts[i] = xyz[idxs[j] * 3 + (i - j * 3)];
    }
}

template <int T, int P, int J, bool GENERIC, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_resident_kernel(int n, int m, int L, const float *__restrict__ xyz,
                                                         float *__restrict__ temp, int *__restrict__ idxs, float *__restrict__ pts) {
    constexpr int W = T / 64;
    extern __shared__ float4 smem_f4[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_f4);  // [3] rotating max slots (+pad to 64 B)
    float *sxyz = reinterpret_cast<float *>(smem_f4) + 16;                        // [n*3] when LDS_XYZ
    // [m] selected indices, written out once at the end: a global store per iteration would sit in front of every barrier
    int *sidx = reinterpret_cast<int *>(sxyz + (LDS_XYZ ? n * 3 : 0));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;  // NULL: a fresh sampling -- every running distance starts at 1e10 and is not stored
    idxs += (size_t)blockIdx.x * m;
    if (pts) pts += (size_t)blockIdx.x * m * 3;

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
        pt[j] = ok ? (temp ? temp[kk] : 1e10f) : -INFINITY;  // padding: never selected, never stored
        nsec[j] = ~fps_sec((uint32_t)kk, L);
    }
    if (LDS_XYZ) {
        for (int i = tid; i < n * 3; i += T) sxyz[i] = xyz[i];
    }
    if (tid < 3) slots[tid] = 0ull;
    if (tid == 0) sidx[0] = 0;
    __syncthreads();

    int old = 0;
    int s_cur = 0, s_nxt = 1;
#ifdef MCP_FPS_DIAG
    unsigned long long t_prev = 0;
    FPS_STAMP(7);
#endif
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if (LDS_XYZ) {
            x1 = sxyz[old * 3 + 0]; y1 = sxyz[old * 3 + 1]; z1 = sxyz[old * 3 + 2];
        } else {
            x1 = xyz[old * 3 + 0]; y1 = xyz[old * 3 + 1]; z1 = xyz[old * 3 + 2];
        }
        FPS_STAMP(0);  // centre read
        uint32_t hi, lo;
        if (!GENERIC) {
            // track only the maximum VALUE in the scan (packed fp32 math); which point holds it is found afterwards
            float best;
            if constexpr (P >= 2) {
                const f2 c0 = {x1, x1}, c1 = {y1, y1}, c2 = {z1, z1};
                f2 m2 = {-1.0f, -1.0f};
#pragma unroll
                for (int p = 0; p < P; p += 2) {
                    const f2 dx = f2{px[p], px[p + 1]} - c0, dy = f2{py[p], py[p + 1]} - c1, dz = f2{pz[p], pz[p + 1]} - c2;
                    const f2 d = f2_fma(dz, dz, f2_fma(dy, dy, dx * dx));
                    pt[p] = mcp_min_raw(d.x, pt[p]);  // raw v_min / v_max: fminf / fmaxf first canonicalise an operand (an extra
                    pt[p + 1] = mcp_min_raw(d.y, pt[p + 1]);  // instruction each); nothing here is ever NaN
                    m2.x = mcp_max_raw(m2.x, pt[p]);
                    m2.y = mcp_max_raw(m2.y, pt[p + 1]);
                }
                best = mcp_max_raw(m2.x, m2.y);
            } else {
                pt[0] = fminf(mcp_sqdist3(px[0], py[0], pz[0], x1, y1, z1), pt[0]);
                best = fmaxf(-1.0f, pt[0]);
            }
            hi = mcp_ord(best);
            FPS_STAMP(1);  // scan
            const uint32_t whi = mcp_wave_max_u32(hi);
            // tie order inside the thread: reverse-priority assignment chain, highest priority assigned last.
            // (An independent-select + max-tree form was measured 4 % SLOWER: this phase is bound by instruction
            // count -- about 12 cycles per instruction with the DPP / SGPR-mask hazards -- not by the chain's depth.)
            uint32_t bsec = 0;
            if constexpr (J == 1) {
#pragma unroll
                for (int p = P - 1; p >= 1; p -= 2) bsec = pt[p] == best ? nsec[p] : bsec;  // odd p (reference thread tid+T)
#pragma unroll
                for (int p = P - 2; p >= 0; p -= 2) bsec = pt[p] == best ? nsec[p] : bsec;  // even p (reference thread tid)
            } else {
#pragma unroll
                for (int p = P - 1; p >= 0; --p) bsec = pt[p] == best ? nsec[p] : bsec;
            }
            lo = hi == whi ? bsec : 0u;
            hi = whi;
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
            const uint32_t whi = mcp_wave_max_u32(hi);
            lo = hi == whi ? lo : 0u;
            hi = whi;
        }
        uint32_t wlo = mcp_wave_max_u32(lo);
        FPS_STAMP(2);  // wave reductions + index select
        if (W > 1) {
            // cross-wave: one LDS atomic max per wave on a rotating slot, one barrier, one broadcast read
            if (lane == 0) __hip_atomic_fetch_max(&slots[s_cur], ((unsigned long long)hi << 32) | wlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tid == 0) slots[s_nxt] = 0ull;
            __syncthreads();
            FPS_STAMP(3);  // LDS atomic + barrier
            wlo = (uint32_t)slots[s_cur];
            const int s_new = 3 - s_cur - s_nxt;
            s_cur = s_nxt;
            s_nxt = s_new;
        }
        old = (int)fps_unsec(~wlo, L);
        if (tid == 0) sidx[j] = old;
        FPS_STAMP(4);  // slot read + decode + store
    }
    __syncthreads();
    for (int j = tid; j < m; j += T) idxs[j] = sidx[j];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int k = tid + T * p;
        if (temp && k < n) temp[k] = pt[p];
    }
    fps_emit_points<T>(xyz, idxs, pts, m, tid);
}

// ---------------------------------------------------------------------------------------------
// Spatially pruned variant (n >= MCP_FPS_SPATIAL_MIN).  The resident kernel above is bound by VALU issue:
// every wave re-evaluates all of its points on every iteration although a new centre can only lower the
// running distance of points closer to it than their current value.  Here the workgroup first sorts its
// cloud along a Morton curve (18-bit isotropic cells, in LDS), so a lane owns P consecutive points of the
// curve (a small cell, bounding box in 6 VGPRs) and a wave owns one compact region.  Per iteration a lane
// computes the squared distance from the centre to its box, in the same arithmetic as the point distances:
// fp32 subtraction, multiplication and fma are monotone, so that value is an exact lower bound of every
// point distance the lane would compute, and when it is >= the lane's largest running distance nothing in
// the lane can change.  A wave in which no lane can change skips the update and re-submits its cached
// (value, key) maximum.  On LiDAR-like clouds about 2 of 16 waves update per iteration.  Results are
// bit-identical to the resident kernel: the (ord(d), ~sec(k)) order is total, so neither the arrangement of
// points nor skipping unchanged waves can alter the winner.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fps_spread6(uint32_t v) {  // 6 bits -> every third bit
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) r |= ((v >> b) & 1u) << (3 * b);
    return r;
}

template <int T, int P>
using SpatialSort = rocprim::block_radix_sort<uint32_t, T, P>;

template <int T, int P, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_spatial_kernel(int n, int m, int lds_idx, const float *__restrict__ xyz,
                                                        float *__restrict__ temp, int *__restrict__ idxs, float *__restrict__ pts) {
    constexpr int W = T / 64;  // T * P >= n points are keyed and sorted (a power of two, <= 16384)
    extern __shared__ float4 smem_f4[];
    // header (512 B): [0,24) rotating max slots, [64,448) bbox reduction scratch, [448,472) cloud bbox
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_f4);
    float(*red)[16] = reinterpret_cast<float(*)[16]>(reinterpret_cast<float *>(smem_f4) + 16);
    float *bbox = reinterpret_cast<float *>(smem_f4) + 112;
    char *sort_lds = reinterpret_cast<char *>(smem_f4) + 512;                                  // rocPRIM sort storage first
    float *sxyz = reinterpret_cast<float *>(reinterpret_cast<char *>(smem_f4) + 512);        // [n*3] afterwards
    // [m] selected indices, flushed once at the end (a global store per iteration would sit in front of every barrier)
    int *sidx = reinterpret_cast<int *>(reinterpret_cast<char *>(smem_f4) + lds_idx);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;  // NULL: a fresh sampling -- every running distance starts at 1e10 and is not stored
    idxs += (size_t)blockIdx.x * m;
    if (pts) pts += (size_t)blockIdx.x * m * 3;

    // 1. bounding box of the cloud; a thread reads the P points it will key (blocked arrangement) once
    uint32_t keys[P];
    {
        float qx[P], qy[P], qz[P];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int i = tid * P + u;
            const int ii = i < n ? i : 0;
            qx[u] = xyz[(size_t)ii * 3 + 0];
            qy[u] = xyz[(size_t)ii * 3 + 1];
            qz[u] = xyz[(size_t)ii * 3 + 2];
            lo[0] = fminf(lo[0], qx[u]); lo[1] = fminf(lo[1], qy[u]); lo[2] = fminf(lo[2], qz[u]);
            hi[0] = fmaxf(hi[0], qx[u]); hi[1] = fmaxf(hi[1], qy[u]); hi[2] = fmaxf(hi[2], qz[u]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // max of ord() is max of the float; min via negation
            const float l = -mcp_unord(mcp_wave_max_u32(mcp_ord(-lo[a]))), h = mcp_unord(mcp_wave_max_u32(mcp_ord(hi[a])));
            if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
        }
        if (tid < 3) slots[tid] = 0ull;
        __syncthreads();
        if (tid < 6) {
            float v = red[tid][0];
            for (int w = 1; w < W; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
            bbox[tid] = v;
        }
        __syncthreads();
        // 2. keys: 18-bit Morton code of isotropic cells (LiDAR clouds are flat) above the 14-bit point index
        const float ext = fmaxf(fmaxf(bbox[3] - bbox[0], bbox[4] - bbox[1]), bbox[5] - bbox[2]);
        const float inv = ext > 0.f ? 64.f / ext : 0.f;  // cell assignment only orders the points: any monotone map will do
        const float b0 = bbox[0], b1 = bbox[1], b2 = bbox[2];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int i = tid * P + u;
            const uint32_t c0 = (uint32_t)fminf(fmaxf((qx[u] - b0) * inv, 0.f), 63.f), c1 = (uint32_t)fminf(fmaxf((qy[u] - b1) * inv, 0.f), 63.f),
                           c2 = (uint32_t)fminf(fmaxf((qz[u] - b2) * inv, 0.f), 63.f);
            keys[u] = i < n ? ((fps_spread6(c0) | (fps_spread6(c1) << 1) | (fps_spread6(c2) << 2)) << 14) | (uint32_t)i : 0xFFFFFFFFu;
        }
    }
    // 3. block radix sort of the keys (rocPRIM, four 8-bit passes); thread t ends up with sorted positions t*P .. t*P+P-1
    SpatialSort<T, P>().sort(keys, *reinterpret_cast<typename SpatialSort<T, P>::storage_type *>(sort_lds));
    // 4. a lane takes P consecutive points of the curve; box of the valid ones
    float px[P], py[P], pz[P], pt[P];
    // low key word: 14-bit tie rank above the point's BYTE OFFSET in the coordinate array (index * 12 < 2^18).  n >= 1024 here, so the reference block is 1024
    // threads (L = 10) and sec(k) = bitrev10(k mod 1024) << 22 | k >> 10 compacts, order preserved, to
    // bitrev10(k mod 1024) << 4 | k >> 10; the winner's coordinates are then one mask away (no decode, no multiply).
    uint32_t tag[P];
    float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
    float best = -1.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int s = tid * P + p;
        const bool ok = s < n;  // the n real keys sort in front of the padding
        const int src = ok ? (int)(keys[p] & 0x3FFFu) : 0;
        px[p] = xyz[src * 3 + 0];
        py[p] = xyz[src * 3 + 1];
        pz[p] = xyz[src * 3 + 2];
        pt[p] = ok ? (temp ? temp[src] : 1e10f) : -INFINITY;  // padding: never selected, never stored
        tag[p] = ((0x3FFFu - (((__brev((uint32_t)src & 1023u) >> 22) << 4) | ((uint32_t)src >> 10))) << 18) | ((uint32_t)src * 12u);
        if (ok) {
            lox = fminf(lox, px[p]); loy = fminf(loy, py[p]); loz = fminf(loz, pz[p]);
            hix = fmaxf(hix, px[p]); hiy = fmaxf(hiy, py[p]); hiz = fmaxf(hiz, pz[p]);
            best = fmaxf(best, pt[p]);
        }
    }
    __syncthreads();  // the sort storage is dead: the region is reused for the coordinates
    if (LDS_XYZ) {
        for (int i = tid; i < n * 3; i += T) sxyz[i] = xyz[i];
    }
    if (tid == 0) {
        if (lds_idx) sidx[0] = 0;
        else idxs[0] = 0;
    }
    __syncthreads();
    const f2 bx = {lox, -hix}, by = {loy, -hiy}, bz = {loz, -hiz};

    int old = 0;
    int s_cur = 0, s_nxt = 1;
    uint32_t c_hi = 0, c_lo = 0;  // the wave's cached maximum (wave-uniform)
#ifdef MCP_FPS_DIAG
    unsigned long long t_prev = 0;
    FPS_STAMP(7);
#endif
    // the loop below is a pure latency chain: its placement relative to the instruction-fetch lines is worth ~2 % (measured),
    // so it starts on a 64-byte boundary
    asm volatile(".p2align 6");
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if (LDS_XYZ) {
            const float *c = reinterpret_cast<const float *>(reinterpret_cast<const char *>(sxyz) + old);
            x1 = c[0]; y1 = c[1]; z1 = c[2];
        } else {
            const float *c = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) + old);
            x1 = c[0]; y1 = c[1]; z1 = c[2];
        }
        FPS_STAMP(0);  // centre read
        // exact lower bound of the lane's point distances (see the header comment); an all-padding lane has best = -1
        // (lo - c, c - hi) per axis as one packed add: (lo, -hi) + (-c, c)
        const f2 tx = bx + f2{-x1, x1}, ty = by + f2{-y1, y1}, tz = bz + f2{-z1, z1};
        const float ex = fmaxf(fmaxf(tx.x, tx.y), 0.f), ey = fmaxf(fmaxf(ty.x, ty.y), 0.f), ez = fmaxf(fmaxf(tz.x, tz.y), 0.f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        const bool need = lb < best;
        FPS_STAMP(1);  // box test
        if (j == 1 || __builtin_amdgcn_ballot_w64(need)) {
            const f2 c0 = {x1, x1}, c1 = {y1, y1}, c2 = {z1, z1};
            f2 m2 = {-1.0f, -1.0f};
#pragma unroll
            for (int p = 0; p < P; p += 2) {
                const f2 dx = f2{px[p], px[p + 1]} - c0, dy = f2{py[p], py[p + 1]} - c1, dz = f2{pz[p], pz[p + 1]} - c2;
                const f2 d = f2_fma(dz, dz, f2_fma(dy, dy, dx * dx));
                pt[p] = mcp_min_raw(d.x, pt[p]);
                pt[p + 1] = mcp_min_raw(d.y, pt[p + 1]);
                m2.x = mcp_max_raw(m2.x, pt[p]);
                m2.y = mcp_max_raw(m2.y, pt[p + 1]);
            }
            best = mcp_max_raw(m2.x, m2.y);
            const uint32_t hi = mcp_ord(best);
            const uint32_t whi = mcp_wave_max_u32(hi);
            uint32_t bsec = 0;  // points sit in curve order, so ties inside the lane compare the keys themselves
#pragma unroll
            for (int p = 0; p < P; ++p) bsec = max(bsec, pt[p] == best ? tag[p] : 0u);
            c_lo = mcp_wave_max_u32(hi == whi ? bsec : 0u);
            c_hi = whi;
        }
        FPS_STAMP(2);  // update + wave reductions
        if (lane == 0) {
            // one ds_max_u64, visible to the wait-count pass (the Makefile turns the atomic optimizer off for this file: it would
            // wrap the atomic in a second, redundant first-active-lane election)
            __hip_atomic_fetch_max(&slots[s_cur], ((unsigned long long)c_hi << 32) | c_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (tid == 0) slots[s_nxt] = 0ull;
        __syncthreads();
        FPS_STAMP(3);  // LDS atomic + barrier
        const uint32_t wlo = (uint32_t)slots[s_cur];
        const int s_new = 3 - s_cur - s_nxt;
        s_cur = s_nxt;
        s_nxt = s_new;
        old = (int)(wlo & 0x3FFFFu);  // byte offset of the new centre
        if (tid == 0) {
            if (lds_idx) sidx[j] = old;  // offsets; converted to indices in the final flush
            else idxs[j] = old / 12;
        }
        FPS_STAMP(4);  // slot read + store
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (temp && tid * P + p < n) temp[(tag[p] & 0x3FFFFu) / 12u] = pt[p];
    }
    if (lds_idx) {
        __syncthreads();
        for (int i = tid; i < m; i += T) idxs[i] = sidx[i] / 12;
    }
    fps_emit_points<T>(xyz, idxs, pts, m, tid);
}

// Large-N fallback (N > 16 points per lane at 1024 threads): temp stays in global memory,
// xyz is re-read from L2 each iteration, same (ord, ~sec) reduction.  Correct for any N.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, int L, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs, float *__restrict__ pts) {
    __shared__ uint2 slots[2][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;  // NULL: a fresh sampling -- every running distance starts at 1e10 and is not stored
    idxs += (size_t)blockIdx.x * m;
    if (pts) pts += (size_t)blockIdx.x * m * 3;
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
    fps_emit_points<1024>(xyz, idxs, pts, m, tid);
}

// ---------------------------------------------------------------------------------------------
// Tiled variant for 16384 < N <= 65536 (BASELINE config 5).  The points no longer fit in one workgroup's registers, so
// they stay in memory -- but in a Morton-cell order built once by an in-LDS counting sort (2^15 cells, key bits shared out to the axes by extent), as
// tiles of 64 consecutive points.  Every thread owns one tile: its bounding box, its largest running distance and that
// point's key live in the thread's registers, the candidate's coordinates in LDS.  Per iteration a lane runs the same
// exact box test as fps_spatial_kernel; only the tiles that can change are streamed (one coalesced 1 KB read + 256 B
// write each, from L2), everything else is register work.  The plain streaming kernel below re-reads all N points per
// iteration (30 ms at 8 x 65536 -> 2048; this one: a few ms).  Same (ord(d), ~sec(k)) total order, so the same indices.
// Workspace (sorted float4 points + running distances) is taken from the stream-ordered allocator for the call.
// ---------------------------------------------------------------------------------------------
constexpr int TL_T = 1024, TL_PTS = 64, TL_CELLS = 32768, TL_BATCH = 4;
using TiledScan = rocprim::block_scan<uint32_t, TL_T>;

__device__ __forceinline__ uint32_t tl_spread5(uint32_t v) {
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) r |= ((v >> b) & 1u) << (3 * b);
    return r;
}
// tie rank of original index k for a 1024-thread reference block: bitrev10(k mod 1024) << 6 | k >> 10  (k < 65536)
__device__ __forceinline__ uint32_t tl_rank(uint32_t k) { return ((__brev(k & 1023u) >> 22) << 6) | (k >> 10); }
__device__ __forceinline__ uint32_t tl_unrank(uint32_t r) { return (__brev(r >> 6) >> 22) | ((r & 63u) << 10); }

__global__ __launch_bounds__(TL_T) void fps_tiled_kernel(int n, int m, int tiles, int lds_idx, const float *__restrict__ xyz,
                                                         float *__restrict__ temp, int *__restrict__ idxs, float *__restrict__ pts,
                                                         float4 *__restrict__ sx, float *__restrict__ st) {
    extern __shared__ float4 smem_f4[];
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem_f4);                  // [3]
    float(*red)[16] = reinterpret_cast<float(*)[16]>(reinterpret_cast<float *>(smem_f4) + 16);   // [6][16]
    float *bbox = reinterpret_cast<float *>(smem_f4) + 112;                                       // [6]
    char *body = reinterpret_cast<char *>(smem_f4) + 512;
    uint32_t *hist = reinterpret_cast<uint32_t *>(body);                                          // [TL_CELLS] during the sort
    typename TiledScan::storage_type &scan_lds = *reinterpret_cast<typename TiledScan::storage_type *>(body + TL_CELLS * 4);
    float4 *tcand = reinterpret_cast<float4 *>(body);                                             // [tiles] afterwards
    int *sidx = reinterpret_cast<int *>(reinterpret_cast<char *>(smem_f4) + lds_idx);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = tiles * TL_PTS;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;  // NULL: a fresh sampling -- every running distance starts at 1e10 and is not stored
    idxs += (size_t)blockIdx.x * m;
    if (pts) pts += (size_t)blockIdx.x * m * 3;
    sx += (size_t)blockIdx.x * np;
    st += (size_t)blockIdx.x * np;

    // 1. bounding box, cleared histogram
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = tid; i < n; i += TL_T) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[(size_t)i * 3 + a];
                lo[a] = fminf(lo[a], v);
                hi[a] = fmaxf(hi[a], v);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = -mcp_unord(mcp_wave_max_u32(mcp_ord(-lo[a]))), h = mcp_unord(mcp_wave_max_u32(mcp_ord(hi[a])));
            if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
        }
        for (int c = tid; c < TL_CELLS; c += TL_T) hist[c] = 0u;
        if (tid < 3) slots[tid] = 0ull;
        __syncthreads();
        if (tid < 6) {
            float v = red[tid][0];
            for (int w = 1; w < TL_T / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
            bbox[tid] = v;
        }
        __syncthreads();
    }
    // 15 key bits shared out to the axes so that the cells come out as cubic as the extents allow (a LiDAR sweep is
    // 160 x 160 x 10 m: x and y get 6 bits, z 3); bit s of the key (from the top) belongs to axis axis_of[s]
    const float b0 = bbox[0], b1 = bbox[1], b2 = bbox[2];
    float size[3] = {bbox[3] - b0, bbox[4] - b1, bbox[5] - b2};
    int bits[3] = {0, 0, 0};
    uint32_t axis_of = 0;  // 2 bits per key bit, top key bit first
#pragma unroll
    for (int sbit = 0; sbit < 15; ++sbit) {
        const int a = size[0] >= size[1] ? (size[0] >= size[2] ? 0 : 2) : (size[1] >= size[2] ? 1 : 2);
        axis_of = (axis_of << 2) | (uint32_t)a;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k == a) { size[k] *= 0.5f; bits[k] += 1; }
    }
    const float ext0 = bbox[3] - b0, ext1 = bbox[4] - b1, ext2 = bbox[5] - b2;
    const float inv0 = ext0 > 0.f ? (float)(1 << bits[0]) / ext0 : 0.f, inv1 = ext1 > 0.f ? (float)(1 << bits[1]) / ext1 : 0.f,
                inv2 = ext2 > 0.f ? (float)(1 << bits[2]) / ext2 : 0.f;
    const float top0 = (float)((1 << bits[0]) - 1), top1 = (float)((1 << bits[1]) - 1), top2 = (float)((1 << bits[2]) - 1);
    auto cell_of = [&](float x, float y, float z) -> uint32_t {
        const uint32_t q[3] = {(uint32_t)fminf(fmaxf((x - b0) * inv0, 0.f), top0), (uint32_t)fminf(fmaxf((y - b1) * inv1, 0.f), top1),
                               (uint32_t)fminf(fmaxf((z - b2) * inv2, 0.f), top2)};
        int left[3] = {bits[0], bits[1], bits[2]};
        uint32_t key = 0;
#pragma unroll
        for (int sbit = 0; sbit < 15; ++sbit) {
            const int a = (int)((axis_of >> (2 * (14 - sbit))) & 3u);
            uint32_t bit = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k == a) { left[k] -= 1; bit = (q[k] >> left[k]) & 1u; }
            key = (key << 1) | bit;
        }
        return key;
    };
    // 2. counting sort by cell: histogram, exclusive scan, scatter
    for (int i = tid; i < n; i += TL_T) atomicAdd(&hist[cell_of(xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2])], 1u);
    __syncthreads();
    {
        constexpr int CPT = TL_CELLS / TL_T;  // 32 consecutive cells per thread
        uint32_t loc[CPT], sum = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) { loc[c] = sum; sum += hist[tid * CPT + c]; }
        uint32_t base;
        TiledScan().exclusive_scan(sum, base, 0u, scan_lds);
#pragma unroll
        for (int c = 0; c < CPT; ++c) hist[tid * CPT + c] = base + loc[c];
    }
    __syncthreads();
    for (int i = tid; i < n; i += TL_T) {
        const float x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
        const uint32_t pos = atomicAdd(&hist[cell_of(x, y, z)], 1u);
        const uint32_t tag = ((0xFFFFu - tl_rank((uint32_t)i)) << 10) | (pos >> 6);  // tie key above the tile id
        sx[pos] = make_float4(x, y, z, __uint_as_float(tag));
        st[pos] = temp ? temp[i] : 1e10f;
    }
    for (int i = n + tid; i < np; i += TL_T) {  // padding of the last tile: never selected
        sx[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
        st[i] = -INFINITY;
    }
    __syncthreads();  // workgroup-scope release/acquire: the sorted arrays are visible to every wave (same CU)
    // 3. every thread owns one tile: box, largest running distance + its key, candidate coordinates.  Tiles are dealt
    //    round-robin to the waves (lane l of wave w owns tile 16 l + w): a new centre changes a handful of ADJACENT tiles,
    //    which this way are streamed by different waves in parallel instead of queueing in one
    const int mytile = lane * (TL_T / 64) + wave;
    float lox = INFINITY, loy = INFINITY, loz = INFINITY, hix = -INFINITY, hiy = -INFINITY, hiz = -INFINITY;
    float tval = -INFINITY;
    uint32_t ttag = 0;
    if (mytile < tiles) {
        float4 cand = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int e = 0; e < TL_PTS; ++e) {
            const float4 p = sx[mytile * TL_PTS + e];
            const float t = st[mytile * TL_PTS + e];
            const uint32_t tg = __float_as_uint(p.w);
            if (t != -INFINITY) {  // a real point
                lox = fminf(lox, p.x); loy = fminf(loy, p.y); loz = fminf(loz, p.z);
                hix = fmaxf(hix, p.x); hiy = fmaxf(hiy, p.y); hiz = fmaxf(hiz, p.z);
                if (t > tval || (t == tval && tg > ttag)) { tval = t; ttag = tg; cand = p; }
            }
        }
        tcand[mytile] = cand;
    }
    if (tid == 0) {
        if (lds_idx) sidx[0] = 0;
        else idxs[0] = 0;
    }
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];  // the first centre is point 0
    int s_cur = 0, s_nxt = 1;
    uint32_t c_hi = 0, c_lo = 0;
    __syncthreads();

    for (int j = 1; j < m; ++j) {
        const float ex = fmaxf(fmaxf(lox - cx, cx - hix), 0.f), ey = fmaxf(fmaxf(loy - cy, cy - hiy), 0.f),
                    ez = fmaxf(fmaxf(loz - cz, cz - hiz), 0.f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        unsigned long long todo = __builtin_amdgcn_ballot_w64(lb < tval);  // an empty tile has tval = -inf
        const bool touched = j == 1 || todo != 0;
        while (todo) {
            // stream up to TL_BATCH of the wave's affected tiles at once (their loads overlap)
            int tb[TL_BATCH];
            float4 pv[TL_BATCH];
            float tv[TL_BATCH];
#pragma unroll
            for (int u = 0; u < TL_BATCH; ++u) {
                tb[u] = todo ? (int)__builtin_ctzll(todo) : -1;
                if (todo) todo &= todo - 1;
                if (tb[u] >= 0) {
                    const int e = (tb[u] * (TL_T / 64) + wave) * TL_PTS + lane;
                    pv[u] = sx[e];
                    tv[u] = st[e];
                }
            }
#pragma unroll
            for (int u = 0; u < TL_BATCH; ++u) {
                if (tb[u] < 0) continue;
                const int tile = tb[u] * (TL_T / 64) + wave;
                const float d = mcp_sqdist3(pv[u].x, pv[u].y, pv[u].z, cx, cy, cz);
                const float nt = fminf(d, tv[u]);
                if (nt != tv[u]) st[tile * TL_PTS + lane] = nt;
                const uint32_t hi = mcp_ord(nt), tg = __float_as_uint(pv[u].w);
                const uint32_t whi = mcp_wave_max_u32(hi);
                const uint32_t wtg = mcp_wave_max_u32(hi == whi ? tg : 0u);
                if (hi == whi && tg == wtg) tcand[tile] = pv[u];  // exactly one lane (tags are unique per real point)
                if (lane == tb[u]) { tval = mcp_unord(whi); ttag = wtg; }
            }
        }
        if (touched) {  // the wave's maximum over its 64 tiles
            const uint32_t hi = mcp_ord(tval);
            const uint32_t whi = mcp_wave_max_u32(hi);
            c_lo = mcp_wave_max_u32(hi == whi ? ttag : 0u);
            c_hi = whi;
        }
        if (lane == 0) {
            __hip_atomic_fetch_max(&slots[s_cur], ((unsigned long long)c_hi << 32) | c_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (tid == 0) slots[s_nxt] = 0ull;
        __syncthreads();  // also orders this iteration's tcand writes before the read below
        const uint32_t wlo = (uint32_t)slots[s_cur];
        const int s_new = 3 - s_cur - s_nxt;
        s_cur = s_nxt;
        s_nxt = s_new;
        const float4 c = tcand[wlo & 1023u];
        cx = c.x; cy = c.y; cz = c.z;
        if (tid == 0) {
            const int old = (int)tl_unrank(0xFFFFu - (wlo >> 10));
            if (lds_idx) sidx[j] = old;
            else idxs[j] = old;
        }
    }
    __syncthreads();
    // running distances back in the caller's order
    if (temp)
        for (int i = tid; i < n; i += TL_T) temp[tl_unrank(0xFFFFu - (__float_as_uint(sx[i].w) >> 10))] = st[i];
    if (lds_idx) {
        for (int i = tid; i < m; i += TL_T) idxs[i] = sidx[i];
    }
    fps_emit_points<TL_T>(xyz, idxs, pts, m, tid);
}

size_t tiled_workspace_bytes(int b, int n) {
    const size_t np = (size_t)((n + TL_PTS - 1) / TL_PTS) * TL_PTS;
    return (size_t)b * np * (sizeof(float4) + sizeof(float));
}

// ws: caller-provided scratch of at least tiled_workspace_bytes(b, n) bytes (the library never allocates)
int launch_tiled(int b, int n, int m, const float *xyz, float *temp, int *idx, float *pts, char *ws, hipStream_t s) {
    const int tiles = (n + TL_PTS - 1) / TL_PTS;
    const size_t np = (size_t)tiles * TL_PTS;
    size_t lds = 512 + (size_t)TL_CELLS * 4 + sizeof(typename TiledScan::storage_type);
    int lds_idx = 0;
    const size_t after = 512 + (size_t)tiles * sizeof(float4);  // the index list may alias the histogram, not the candidates
    if (after + (size_t)m * 4 <= 160 * 1024) {
        lds_idx = (int)after;
        if (after + (size_t)m * 4 > lds) lds = after + (size_t)m * 4;
    }
    if (lds > 160 * 1024) return MCP_ERR_UNSUPPORTED;
    float4 *sx = reinterpret_cast<float4 *>(ws);
    float *st = reinterpret_cast<float *>(ws + (size_t)b * np * sizeof(float4));
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(fps_tiled_kernel, dim3(b), dim3(TL_T), lds, s, n, m, tiles, lds_idx, xyz, temp, idx, pts, sx, st);
    return mcp_launch_status();
}

int ref_block_log2(int n) {
    // cuda_utils.h:10-14, same double arithmetic
    int pow_2 = (int)(log((double)n) / log(2.0));
    if (pow_2 > 10) pow_2 = 10;
    if (pow_2 < 0) pow_2 = 0;
    return pow_2;
}

template <int T, int P, int J, bool GENERIC>
int launch_resident(int b, int n, int m, int L, const float *xyz, float *temp, int *idx, float *pts, hipStream_t s) {
    const size_t slot_bytes = 64;
    const size_t xyz_bytes = (size_t)n * 3 * sizeof(float);
    const size_t idx_bytes = (size_t)(m > 0 ? m : 1) * sizeof(int);  // the selected indices are buffered in LDS (m <= n)
    if (xyz_bytes + slot_bytes + idx_bytes <= 150 * 1024) {
        auto kern = fps_resident_kernel<T, P, J, GENERIC, true>;
        static McpPerDeviceOnce attr_once;
        if (attr_once.need()) {
            { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
            attr_once.done();
        }
        hipLaunchKernelGGL(kern, dim3(b), dim3(T), slot_bytes + xyz_bytes + idx_bytes, s, n, m, L, xyz, temp, idx, pts);
    } else {
        if (slot_bytes + idx_bytes > 150 * 1024) return MCP_ERR_UNSUPPORTED;
        auto kern = fps_resident_kernel<T, P, J, GENERIC, false>;
        static McpPerDeviceOnce attr_once2;
        if (attr_once2.need()) {
            { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
            attr_once2.done();
        }
        hipLaunchKernelGGL(kern, dim3(b), dim3(T), slot_bytes + idx_bytes, s, n, m, L, xyz, temp, idx, pts);
    }
    return mcp_launch_status();
}

template <int T, int P>
int launch_spatial(int b, int n, int m, int L, const float *xyz, float *temp, int *idx, float *pts, hipStream_t s) {
    const size_t head = 512, key_bytes = sizeof(typename SpatialSort<T, P>::storage_type), xyz_bytes = (size_t)n * 3 * sizeof(float);
    const bool lds_xyz = head + xyz_bytes <= 160 * 1024;
    size_t lds = head + (lds_xyz && xyz_bytes > key_bytes ? xyz_bytes : key_bytes);
    int lds_idx = 0;  // byte offset of the in-LDS index list, 0 = write indices straight to global memory
    if (lds + (size_t)m * 4 <= 160 * 1024) {
        lds_idx = (int)lds;
        lds += (size_t)m * 4;
    }
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_spatial_kernel<T, P, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(fps_spatial_kernel<T, P, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    (void)L;  // n >= 1024: the reference block size is 1024 (asserted by the caller)
    if (lds_xyz) hipLaunchKernelGGL((fps_spatial_kernel<T, P, true>), dim3(b), dim3(T), lds, s, n, m, lds_idx, xyz, temp, idx, pts);
    else hipLaunchKernelGGL((fps_spatial_kernel<T, P, false>), dim3(b), dim3(T), lds, s, n, m, lds_idx, xyz, temp, idx, pts);
    return mcp_launch_status();
}

// tuning hooks exist only in -DMCP_AB builds; the shipped library reads no environment variable
int env_int(const char *name, int dflt) {
#ifdef MCP_AB
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// workgroup size for the spatial kernel by sort size (tuning override: MCP_FPS_T)
int launch_spatial_any(int b, int n, int m, int L, const float *xyz, float *temp, int *idx, float *pts, hipStream_t s) {
    static const int force_t = env_int("MCP_FPS_T", 0);
    int ns = 2048;
    while (ns < n) ns <<= 1;
    const int t = force_t ? force_t : 1024;
    switch (ns / t) {
        case 2: if (t == 1024) return launch_spatial<1024, 2>(b, n, m, L, xyz, temp, idx, pts, s); break;
        case 4:
            if (t == 1024) return launch_spatial<1024, 4>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 512) return launch_spatial<512, 4>(b, n, m, L, xyz, temp, idx, pts, s);
            break;
        case 8:
            if (t == 1024) return launch_spatial<1024, 8>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 512) return launch_spatial<512, 8>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 256) return launch_spatial<256, 8>(b, n, m, L, xyz, temp, idx, pts, s);
            break;
        case 16:
            if (t == 1024) return launch_spatial<1024, 16>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 512) return launch_spatial<512, 16>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 256) return launch_spatial<256, 16>(b, n, m, L, xyz, temp, idx, pts, s);
            break;
        case 32:
            if (t == 512) return launch_spatial<512, 32>(b, n, m, L, xyz, temp, idx, pts, s);
            if (t == 256) return launch_spatial<256, 32>(b, n, m, L, xyz, temp, idx, pts, s);
            break;
    }
    return MCP_ERR_UNSUPPORTED;
}

}  // namespace

#ifdef MCP_FPS_DIAG
extern "C" __attribute__((visibility("default"))) int mcp_fps_diag_read(unsigned long long *out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fps_diag), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_diag), z, sizeof(z));
    return (int)e;
}
#endif

namespace {
bool tiled_range(int n) { return n > 16384 && n <= 65536; }

int fps_dispatch(int b, int n, int m, const float *xyz, float *temp, int *idx, float *pts, char *ws, size_t ws_bytes, hipStream_t s) {
    const int L = ref_block_log2(n);
    const int bs = 1 << L;
    int rc;
    mcp_prof_begin(MCP_KERNEL_FPS, s);
    static const int spatial_min = env_int("MCP_FPS_SPATIAL_MIN", 1024);
    if (n >= spatial_min && L == 10 && n <= 16384 && m > 1) {
        rc = launch_spatial_any(b, n, m, L, xyz, temp, idx, pts, s);
    } else if (bs >= 64) {
        const int P = (n + bs - 1) / bs;
        if (bs == 1024) {
            // half-size workgroups (J = 1) from 4 points per reference thread up: fewer waves in the reduction
            if (P <= 1) rc = launch_resident<1024, 1, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
            else if (P <= 2) rc = launch_resident<1024, 2, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
            else if (P <= 4) rc = launch_resident<MCP_FPS_T4, MCP_FPS_P4, MCP_FPS_J4, false>(b, n, m, L, xyz, temp, idx, pts, s);
            else if (P <= 8) rc = launch_resident<MCP_FPS_T8, MCP_FPS_P8, MCP_FPS_J8, false>(b, n, m, L, xyz, temp, idx, pts, s);
            else if (P <= 16) rc = launch_resident<1024, 16, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
            else {
                // the tiled kernel needs scratch for the sorted cloud; without it (or beyond its range): plain streaming, any N
                rc = (tiled_range(n) && ws && ws_bytes >= tiled_workspace_bytes(b, n)) ? launch_tiled(b, n, m, xyz, temp, idx, pts, ws, s)
                                                                                      : MCP_ERR_UNSUPPORTED;
                if (rc == MCP_ERR_UNSUPPORTED && temp) {  // (the streaming kernel keeps its running distances IN temp)
                    hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, s, n, m, L, xyz, temp, idx, pts);
                    rc = mcp_launch_status();
                }
            }
        } else if (bs == 512) rc = launch_resident<512, 2, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
        else if (bs == 256) rc = launch_resident<256, 2, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
        else if (bs == 128) rc = launch_resident<128, 2, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
        else rc = launch_resident<64, 2, 0, false>(b, n, m, L, xyz, temp, idx, pts, s);
    } else {
        rc = launch_resident<64, 1, 0, true>(b, n, m, L, xyz, temp, idx, pts, s);  // n < 64
    }
    mcp_prof_end(MCP_KERNEL_FPS, s);
    return rc;
}
}  // namespace

MCP_EXPORT size_t mcp_fps_workspace_bytes(int b, int n, int m) {
    (void)m;
    return (b > 0 && tiled_range(n)) ? tiled_workspace_bytes(b, n) : 0;
}

MCP_EXPORT int mcp_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && temp && idx);
    if (m <= 0) return MCP_OK;  // sampling_gpu.cu:100
    return fps_dispatch(b, n, m, xyz, temp, idx, nullptr, nullptr, 0, (hipStream_t)stream);
}

MCP_EXPORT int mcp_furthest_point_sampling_ws(int b, int n, int m, const float *xyz, float *temp, int *idx, void *workspace,
                                              size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && temp && idx);
    if (m <= 0) return MCP_OK;
    return fps_dispatch(b, n, m, xyz, temp, idx, nullptr, static_cast<char *>(workspace), workspace_bytes, (hipStream_t)stream);
}

/* A fresh sampling (every call of the caller graph is one): the running distances start at 1e10 inside the kernel and are not
 * returned, so the caller neither fills nor allocates the (b,n) temp buffer of the reference interface; sampled_xyz (b,m,3),
 * optional, receives the coordinates of the selected points (the index_points_gather the callers run next, mocopci.py:1379).
 * Same indices as mcp_furthest_point_sampling_ws with temp = 1e10.  MCP_ERR_UNSUPPORTED where only the streaming kernel applies
 * (n > 65536, or 16384 < n <= 65536 without workspace): use the temp interface there. */
MCP_EXPORT int mcp_furthest_point_sampling_fresh(int b, int n, int m, const float *xyz, int *idx, float *sampled_xyz, void *workspace,
                                                 size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && idx);
    if (m <= 0) return MCP_OK;
    return fps_dispatch(b, n, m, xyz, nullptr, idx, sampled_xyz, static_cast<char *>(workspace), workspace_bytes, (hipStream_t)stream);
}
