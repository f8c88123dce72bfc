// pointconv.hip -- fused grouping + WeightNet + neighbour aggregation of PointConv / PointConvD for gfx950.
//
// Reference (mocopci.py:1218-1266 group/group_query, :1289-1300 WeightNet, :1330-1335 / :1381-1387):
//   idx -> two K5 gathers (+ permute copies) -> cat [dxyz | feats] (B,S,32,3+D) -> WeightNet 3->8->8->8
//   (three Conv2d+ReLU launches on (B,8,32,S)) -> batched matmul (3+D,32)x(32,8) per point -> (B,S,(3+D)*8).
// At level 0 the intermediates are 280 MiB per tensor.  Here one workgroup owns PPB points:
//   phase 1: one thread per (point, neighbour) gathers the neighbour coordinate, forms dxyz and runs
//            the 3->8->8->8 MLP in registers (weights arrive as scalar loads), leaving dxyz and the
//            8 kernel weights in LDS;
//   phase 2: one thread per (point, 4 consecutive channels) walks the 32 neighbours: the gathered feature
//            rows are read as one float4 per lane (coalesced row segments), the 8 weights are LDS
//            broadcasts shared by the four channels, and the (3+D) x 8 aggregate is accumulated as
//            ascending-k fma chains and written as 128 contiguous bytes.
// The following Linear((3+D)*8 -> C_out) + LeakyReLU (mocopci.py:1336-1342) runs in the same kernel for D = 32 / 64
// (pointconv_linear_kernel below: levels 0, 1 and the refinement stage); wider layers write the aggregate and use mcp_linear.
#include "common.h"
#include "mfma_split.h"

namespace {

constexpr int K = 32, WN = 8;

// PPB points per workgroup iteration; phase 2: one thread per (point, FOUR consecutive channels).  The round-2 kernel gave a
// thread one channel: per neighbour one 4-byte gather, two 16-byte LDS reads of the point's 8 kernel weights and 8 fma -- with four
// SIMDs behind one LDS port that is 64 LDS cycles per 32 VALU cycles: the loop was LDS-bound (VALU ~50 % busy in the PMC pass).
// Four channels share the weights: one 16-byte gather (the four channels are adjacent in the gathered row), the same two LDS
// reads, 32 fma -- a quarter of the LDS traffic and of the load instructions per fma.  Same ascending-k fma chain per output, so
// the results are bit-identical to the round-2 kernel.  PPB is chosen by the host so that PPB * d / 4 items fill the 256 threads.
template <int THREADS, int PPB>
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
    const int d4 = d >> 2;
    const McpUnits units = mcp_units_by_xcd(total, PPB);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p0 = units.first; p0 < units.limit; p0 += units.stride) {
        __syncthreads();
        // ---- phase 1: WeightNet 3 -> 8 -> 8 -> 8 per (point, neighbour) ----
        for (int pair = tid; pair < PPB * K; pair += THREADS) {
            const int pl = pair >> 5, k = pair & 31;
            const long long p = p0 + pl;
            if (p < total) {
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
        // ---- phase 2: (point, 4 channels) items; feature channels from the gathered rows, then the 3 coordinate channels ----
        for (int it = tid; it < PPB * d4; it += THREADS) {
            const int pl = it / d4, c = (it - pl * d4) * 4;
            const long long p = p0 + pl;
            if (p >= total) break;
            const long long bb = mcp_div(p, s, f32);
            float acc[4][WN];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[u][j] = 0.f;
            const float *fb = s_points + (long long)bb * n * d + c;
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                // row offsets within one batch element fit 32 bits whenever n * d does (checked by the host: off32)
                const float4 f = *reinterpret_cast<const float4 *>(off32 ? fb + (unsigned)il[pl][k] * (unsigned)d : fb + (long long)il[pl][k] * d);
                const float4 wa = *reinterpret_cast<const float4 *>(&wl[pl][k][0]);
                const float4 wb = *reinterpret_cast<const float4 *>(&wl[pl][k][4]);
                const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u][0] = __builtin_fmaf(fv[u], wa.x, acc[u][0]); acc[u][1] = __builtin_fmaf(fv[u], wa.y, acc[u][1]);
                    acc[u][2] = __builtin_fmaf(fv[u], wa.z, acc[u][2]); acc[u][3] = __builtin_fmaf(fv[u], wa.w, acc[u][3]);
                    acc[u][4] = __builtin_fmaf(fv[u], wb.x, acc[u][4]); acc[u][5] = __builtin_fmaf(fv[u], wb.y, acc[u][5]);
                    acc[u][6] = __builtin_fmaf(fv[u], wb.z, acc[u][6]); acc[u][7] = __builtin_fmaf(fv[u], wb.w, acc[u][7]);
                }
            }
            float4 *o = reinterpret_cast<float4 *>(out + (p * cin + 3 + c) * WN);  // 4 channels x 8 = 128 contiguous bytes
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o[2 * u] = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
                o[2 * u + 1] = make_float4(acc[u][4], acc[u][5], acc[u][6], acc[u][7]);
            }
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

// The few-point launches of the lower pyramid levels (<= 16384 centres) are latency chains on an otherwise idle chip: there the
// round-2 arrangement -- 8 points per workgroup, one channel per thread, 1024-thread groups for the widest layers -- finishes sooner
// (more, smaller workgroups; 12.8 vs 16.3 us at level 4, 16.4 vs 18.8 us at level 2).  Same arithmetic, same results.
constexpr int LPPB = 8;
template <int THREADS>
__global__ __launch_bounds__(THREADS) void pointconv_agg_lowlevel_kernel(long long total, int n, int s, int d, const float *__restrict__ s_xyz,
                                                                const float *__restrict__ new_xyz, const float *__restrict__ s_points,
                                                                const int *__restrict__ idx, const float *__restrict__ w0,
                                                                const float *__restrict__ b0, const float *__restrict__ w1,
                                                                const float *__restrict__ b1, const float *__restrict__ w2,
                                                                const float *__restrict__ b2, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float wl[LPPB][K][WN];  // kernel weights per (point, neighbour)
    __shared__ float gx[LPPB][K][3];                                // dxyz
    __shared__ int il[LPPB][K];
    const int tid = threadIdx.x;
    const int cin = d + 3;
    const bool f32 = mcp_fits32(total);
    const bool off32 = (long long)n * d < (1LL << 31);
    const McpUnits units = mcp_units_by_xcd(total, LPPB);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p0 = units.first; p0 < units.limit; p0 += units.stride) {
        __syncthreads();
        {   // ---- phase 1 ----
            const int pl = tid >> 5, k = tid & 31;
            const long long p = p0 + pl;
            if (tid < LPPB * K && p < total) {
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
        for (int it = tid; it < LPPB * d; it += THREADS) {
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
        if (tid < LPPB * 3) {  // the three coordinate channels of every point: dxyz from LDS
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

// ---- grouping + WeightNet + aggregation + Linear + LeakyReLU in one kernel (D = 32, 64) --------------------------------------
// A workgroup owns 32 points = one MFMA row tile.  Phases 1 and 2 are those of pointconv_agg_kernel (one (point, 4 channels) item per
// thread: 8 D threads), but the (3+D) x 8 aggregate of the 32 points never leaves the CU: after a barrier it is written over the
// WeightNet scratch as a 32 x K tile (K = 8 D + 24, padded to whole chunks of 32) in the channel order of the two-kernel form, and
// wave t < NT runs output tile t of the projection on it with EXACTLY the arithmetic of linear_kernel (linear.hip): rows on the MFMA
// column, fp32 operands split exactly into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16 k-values in the same order, K
// chunks in ascending order into one accumulator that starts from the bias, the same activation -- the output is BIT-IDENTICAL to
// mcp_pointconv_agg + mcp_linear at the row counts where mcp_linear runs linear_kernel (>= 16384 rows; the host uses this kernel
// only there), so no stored result moves.  The weight pieces come straight from mcp_linear_pack's image in L2 (55 KB / 206 KB per
// workgroup, against 72 KB / 137 KB of aggregate written to and re-read from HBM by the two-kernel form), each chunk's requested as
// soon as the previous chunk's MFMAs have issued; the other waves of the workgroup wait at the next barrier meanwhile (1.4 / 2.7 us
// of a ~25 us workgroup; the co-resident workgroups fill the SIMDs).  Output rows leave through a wave-private staging tile as
// coalesced float4 rows.
constexpr int FPPB = 32;
template <int D, int NT>
struct FusedCfg {
    static constexpr int THREADS = 8 * D;
    static constexpr int KTOT = 8 * D + 24, NCH = (KTOT + 31) / 32, KP = NCH * 32 + 4;   // chunks; padded tile row (floats): KP % 32 == 4
    static constexpr int SCRATCH_BYTES = FPPB * K * (WN + 3 + 1) * 4;                    // wl | gx | il = 48 KB
    static constexpr int TILE_BYTES = FPPB * KP * 4;
    static constexpr int XP = 36;                                                        // padded row of an output staging tile
    static constexpr int STAGE_BYTES = NT * 32 * XP * 4;                                 // behind the tile: a wave may store while another still reads
    static constexpr int LDS_BYTES = (SCRATCH_BYTES > TILE_BYTES + STAGE_BYTES ? SCRATCH_BYTES : TILE_BYTES + STAGE_BYTES);
    static_assert(NT <= THREADS / 64, "one wave per output tile");
    static_assert(2 * LDS_BYTES <= 160 * 1024 || D < 64, "two workgroups per CU");
};

template <int D, int NT>
__global__ __launch_bounds__(8 * D) __attribute__((amdgpu_waves_per_eu(D == 32 ? 3 : 4))) void pointconv_linear_kernel(long long total, int n, int s, const float *__restrict__ s_xyz,
                                                                  const float *__restrict__ new_xyz, const float *__restrict__ s_points,
                                                                  const int *__restrict__ idx, const float *__restrict__ w0,
                                                                  const float *__restrict__ b0, const float *__restrict__ w1,
                                                                  const float *__restrict__ b1, const float *__restrict__ w2,
                                                                  const float *__restrict__ b2, const float *__restrict__ packed, int cout,
                                                                  float slope, float *__restrict__ out) {
    using C = FusedCfg<D, NT>;
    constexpr int THREADS = C::THREADS, NCH = C::NCH, KP = C::KP, KTOT = C::KTOT, D4 = D / 4, XP = C::XP;
    constexpr int CF = NT * 2 * 3 * 64 * 4;  // floats per chunk of the weight image (chunk_floats(NT) of linear.hip)
    // neighbour-loop unrolling: D = 64 runs at 128 VGPRs (two 512-thread workgroups per CU).  Unrolled by 2 the compiler keeps 2 registers in
    // scratch (12 bytes); the spill-free build (unrolled by 1, 118 VGPRs) is SLOWER -- 74.6 vs 66.2 us at encoder level 1, 104.7 vs 93.7 us
    // at the refinement stage (profiles/r05_pointconv_unroll_ab.txt) -- so the two spilled registers stay; deeper unrolling spills more
    constexpr int UNR = D == 32 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) float flds[];
    float(*wl)[K][WN] = reinterpret_cast<float(*)[K][WN]>(flds);                               // [FPPB][K][WN]
    float(*gx)[K][3] = reinterpret_cast<float(*)[K][3]>(flds + FPPB * K * WN);                 // [FPPB][K][3]
    int(*il)[K] = reinterpret_cast<int(*)[K]>(flds + FPPB * K * (WN + 3));                     // [FPPB][K]
    float *tile = flds;                                                                        // [FPPB][KP], over the three above
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const bool f32 = mcp_fits32(total);
    const bool off32 = (long long)n * D < (1LL << 31);
    const McpUnits units = mcp_units_by_xcd(total, FPPB);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p0 = units.first; p0 < units.limit; p0 += units.stride) {
        __syncthreads();
        // ---- phase 1: WeightNet 3 -> 8 -> 8 -> 8 per (point, neighbour) ----
        for (int pair = tid; pair < FPPB * K; pair += THREADS) {
            const int pl = pair >> 5, k = pair & 31;
            const long long p = p0 + pl;
            if (p < total) {
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
        // ---- phase 2: one (point, 4 channels) item per thread, the aggregate stays in registers ----
        const int pl = tid / D4, c4 = tid - pl * D4;
        float acc[4][WN], accx[WN];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[u][j] = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) accx[j] = 0.f;
        if (p0 + pl < total) {
            const long long bb = mcp_div(p0 + pl, s, f32);
            const float *fb = s_points + (long long)bb * n * D + c4 * 4;
#pragma unroll UNR
            for (int k = 0; k < K; ++k) {
                const float4 f = *reinterpret_cast<const float4 *>(off32 ? fb + (unsigned)il[pl][k] * (unsigned)D : fb + (long long)il[pl][k] * D);
                const float4 wa = *reinterpret_cast<const float4 *>(&wl[pl][k][0]);
                const float4 wb = *reinterpret_cast<const float4 *>(&wl[pl][k][4]);
                const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u][0] = __builtin_fmaf(fv[u], wa.x, acc[u][0]); acc[u][1] = __builtin_fmaf(fv[u], wa.y, acc[u][1]);
                    acc[u][2] = __builtin_fmaf(fv[u], wa.z, acc[u][2]); acc[u][3] = __builtin_fmaf(fv[u], wa.w, acc[u][3]);
                    acc[u][4] = __builtin_fmaf(fv[u], wb.x, acc[u][4]); acc[u][5] = __builtin_fmaf(fv[u], wb.y, acc[u][5]);
                    acc[u][6] = __builtin_fmaf(fv[u], wb.z, acc[u][6]); acc[u][7] = __builtin_fmaf(fv[u], wb.w, acc[u][7]);
                }
            }
        }
        const int xl = tid / 3, xc = tid - xl * 3;  // the three coordinate channels of every point: threads 0..95
        if (tid < FPPB * 3 && p0 + xl < total) {
#pragma unroll UNR
            for (int k = 0; k < K; ++k) {
                const float f = gx[xl][k][xc];
                const float4 wa = *reinterpret_cast<const float4 *>(&wl[xl][k][0]);
                const float4 wb = *reinterpret_cast<const float4 *>(&wl[xl][k][4]);
                accx[0] = __builtin_fmaf(f, wa.x, accx[0]); accx[1] = __builtin_fmaf(f, wa.y, accx[1]);
                accx[2] = __builtin_fmaf(f, wa.z, accx[2]); accx[3] = __builtin_fmaf(f, wa.w, accx[3]);
                accx[4] = __builtin_fmaf(f, wb.x, accx[4]); accx[5] = __builtin_fmaf(f, wb.y, accx[5]);
                accx[6] = __builtin_fmaf(f, wb.z, accx[6]); accx[7] = __builtin_fmaf(f, wb.w, accx[7]);
            }
        }
        // the first weight pieces of this wave's output tile: requested before the hand-over so that they arrive under it
        struct Wts { uint4 w[2][3]; };
        auto request = [&](int c, Wts &g) {
            const uint4 *wc = reinterpret_cast<const uint4 *>(packed + (size_t)min(c, NCH - 1) * CF) + (size_t)min(wave, NT - 1) * 2 * 3 * 64 + lane;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) g.w[s2][pc] = wc[(size_t)(s2 * 3 + pc) * 64];
        };
        Wts wa_;
        if (wave < NT) request(0, wa_);
        __syncthreads();  // every read of wl / gx / il is done: the tile may overwrite them
        {
            float *row = tile + pl * KP + 24 + c4 * 32;  // channel 3 + 4 c4 of the aggregate, 8 values per channel
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<float4 *>(row + 8 * u) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
                *reinterpret_cast<float4 *>(row + 8 * u + 4) = make_float4(acc[u][4], acc[u][5], acc[u][6], acc[u][7]);
            }
            if (tid < FPPB * 3) {
                float *xr = tile + xl * KP + xc * 8;
                *reinterpret_cast<float4 *>(xr) = make_float4(accx[0], accx[1], accx[2], accx[3]);
                *reinterpret_cast<float4 *>(xr + 4) = make_float4(accx[4], accx[5], accx[6], accx[7]);
            } else if (tid < FPPB * 3 + FPPB) {  // the padding of the last chunk (its weights are zero in the image)
                float *xr = tile + (tid - FPPB * 3) * KP + KTOT;
#pragma unroll
                for (int i = 0; i < NCH * 32 - KTOT; i += 4) *reinterpret_cast<float4 *>(xr + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        // ---- phase 3: wave t < NT multiplies the 32 x K tile into output tile t, chunk after chunk (linear_kernel's order) ----
        if (wave < NT) {
            mcp_f32x16 o;
            {
                const float *bi = packed + (size_t)NCH * CF;  // bias image [tile][half][reg]
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = bi[(wave * 2 + h) * 16 + r];
            }
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                mcp_f32x16 xa;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4 *>(tile + col * KP + c * 32 + 8 * g + 4 * h);
                    xa[4 * g + 0] = v.x; xa[4 * g + 1] = v.y; xa[4 * g + 2] = v.z; xa[4 * g + 3] = v.w;
                }
                McpSplit3 xs[2];
                xs[0] = mcp_split_kstep(xa, 0);
                xs[1] = mcp_split_kstep(xa, 1);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    o = mcp_mfma_bf16(wa_.w[s2][2], xs[s2].p1, o);  // the order of mcp_tile_split: small terms first
                    o = mcp_mfma_bf16(wa_.w[s2][0], xs[s2].p3, o);
                    o = mcp_mfma_bf16(wa_.w[s2][1], xs[s2].p2, o);
                    o = mcp_mfma_bf16(wa_.w[s2][1], xs[s2].p1, o);
                    o = mcp_mfma_bf16(wa_.w[s2][0], xs[s2].p2, o);
                    o = mcp_mfma_bf16(wa_.w[s2][0], xs[s2].p1, o);
                }
                request(c + 1, wa_);  // the MFMAs above have read their operands; past the end: a re-read of the last chunk nobody uses
            }
            float *st = flds + C::TILE_BYTES / 4 + wave * 32 * XP;  // wave-private staging tile: accumulator layout -> coalesced rows
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = o[4 * g + u];
                    v[u] = a > 0.f ? a : a * slope;
                }
                *reinterpret_cast<float4 *>(st + col * XP + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
            }
            const int cr = lane >> 3, cq = lane & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long p = p0 + 8 * j + cr;
                const float4 v4 = *reinterpret_cast<const float4 *>(st + (8 * j + cr) * XP + 4 * cq);
                if (p < total) *reinterpret_cast<float4 *>(out + p * cout + 32 * wave + 4 * cq) = v4;
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
    const bool vec4 = !(d & 3) && !(((uintptr_t)s_points) & 15);  // float4 gathers of 4 adjacent channels need both
    mcp_prof_begin(MCP_KERNEL_POINTCONV, st);
    // points per workgroup so that PPB * d / 4 phase-2 items fill the 256 threads: 32 at d = 32, 16 at d = 64, 8 from d = 128 up;
    // one workgroup per PPB points (no persistent loop in practice): the dispatcher balances around whatever else holds CUs
    auto launch = [&](auto kern, int ppb) {
        const unsigned grid = (unsigned)min((total + ppb - 1) / ppb, 1LL << 20);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2, out);
    };
    if (total <= 16384 || !vec4) {
        const unsigned grid = (unsigned)min((total + LPPB - 1) / LPPB, 1LL << 20);
        if (d >= 256 && total <= 8192)
            hipLaunchKernelGGL(pointconv_agg_lowlevel_kernel<1024>, dim3(grid), dim3(1024), 0, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2, out);
        else
            hipLaunchKernelGGL(pointconv_agg_lowlevel_kernel<256>, dim3(grid), dim3(256), 0, st, total, n, s, d, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2, out);
    } else if (d <= 32) launch(pointconv_agg_kernel<256, 32>, 32);
    else if (d <= 64) launch(pointconv_agg_kernel<256, 16>, 16);
    else launch(pointconv_agg_kernel<256, 8>, 8);
    mcp_prof_end(MCP_KERNEL_POINTCONV, st);
    return mcp_launch_status();
}

// PointConv / PointConvD after the sampling, whole (mocopci.py:1330-1342, :1381-1393): grouping, WeightNet, aggregation,
// Linear((3+D)*8 -> C_out) and LeakyReLU(slope) in one launch.  packed: the mcp_linear_pack image of the Linear (one piece of
// (3+D)*8 columns) with its bias.  (D, C_out) in {(32, 32), (64, 64)}, 32 neighbours; anything else: MCP_ERR_UNSUPPORTED
// (use mcp_pointconv_agg + mcp_linear).  Bit-identical to that pair from 16384 rows up.
MCP_EXPORT int mcp_pointconv_linear(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points,
                                    const int *idx, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                                    const float *b2, const float *packed, int c_out, float slope, float *out, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && s > 0 && d > 0 && s_xyz && new_xyz && s_points && idx && w0 && b0 && w1 && b1 && w2 && b2 && packed && out);
    if (k != K || !((d == 32 && c_out == 32) || (d == 64 && c_out == 64))) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)out) | ((uintptr_t)s_points) | ((uintptr_t)packed)) & 15) return MCP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)b * s;
    const unsigned grid = (unsigned)min((total + FPPB - 1) / FPPB, 1LL << 20);
    mcp_prof_begin(MCP_KERNEL_POINTCONV, st);
    if (d == 32) {
        auto kern = pointconv_linear_kernel<32, 1>;
        static McpPerDeviceOnce attr_once;
        if (attr_once.need()) {
            { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
            attr_once.done();
        }
        constexpr int lds = FusedCfg<32, 1>::LDS_BYTES;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, total, n, s, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2,
                           b2, packed, c_out, slope, out);
    } else {
        auto kern = pointconv_linear_kernel<64, 2>;
        static McpPerDeviceOnce attr_once;
        if (attr_once.need()) {
            { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
            attr_once.done();
        }
        constexpr int lds = FusedCfg<64, 2>::LDS_BYTES;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, total, n, s, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2,
                           b2, packed, c_out, slope, out);
    }
    mcp_prof_end(MCP_KERNEL_POINTCONV, st);
    return mcp_launch_status();
}
