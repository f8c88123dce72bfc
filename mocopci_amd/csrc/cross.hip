// cross.hip -- fused PointPWC-style cost volume ("cross") for gfx950.
//
// Reference: CrossLayerLightFeatCosine.cross (pointconv_util.py:750-781),
// BidirectionalLayerFeatCosine.cross (:894-922), FlowEmbeddingLayer.forward (:1126-1161),
// after their two neighbour searches:
//     x0 = LeakyReLU_0.1( points2[idx] + points1 + Conv2d_{3->D}(xyz2[idx] - xyz1) )     (B,D,32,N1)
//     x1 = LeakyReLU_0.1( Conv2d_{D->D}(x0) )
//     out = max over the 32 neighbours
// The reference does this with 4 K5 gathers (each with two permute copies), ~8 elementwise /
// conv launches and a max-pool, materialising (B,D,32,N1) three times.  Here one wave owns one
// point; its 32 neighbours sit on the MFMA column (lane & 31):
//   * the positional term is an MFMA with K=4 ([dx,dy,dz,1] x [Wpos|bpos]) whose accumulator is
//     initialised with points1 (broadcast over neighbours);
//   * gathered rows of points2 are loaded straight INTO the accumulator layout: register r of
//     lane-half h is channel 32t + (r&3) + 8(r>>2) + 4h, i.e. four float4 loads per 32-channel tile;
//   * the D->D layer runs on v_mfma_f32_32x32x2_f32 with x0's accumulator tiles as its B operand
//     (weights pre-permuted in LDS, as in fusion.hip), bias as the initial accumulator;
//   * the max over neighbours is a DPP row reduction + one cross-row shuffle per register.
// HBM/L2 traffic is the compulsory gather (32 rows of D floats per point) + D floats out.
// D = 256 (cross3, pointconv_util.py:783-791): its weight pieces (393 KB) do not fit the CU's 160 KB of LDS; cross256_stream_kernel
// below builds a point's x0 once, keeps it in registers and streams the pieces past it (rounds 2-3 split the output channels over
// workgroups on the f32-input MFMA instead, each workgroup rebuilding x0).
// The D -> D layer runs on the bf16 matrix pipe through the exact three-way operand split of mfma_split.h (six bf16 MFMAs per 16
// k-values instead of eight f32-input ones) at every D; the weight pieces are prepared once by mcp_cross_pack, x0 is split in
// registers as layer 1 leaves it.
#include "common.h"
#include "mfma_split.h"

#ifdef MCP_CROSS_DIAG
// diagnostic build only (never in the product library): shader-cycle totals per phase of the per-point loop, wave 0 of workgroup 0
__device__ unsigned long long g_cross_diag[8];
#define CROSS_STAMP(slot)                                                                            \
    do {                                                                                             \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_cross_diag[slot] += t_ - t_prev; \
        t_prev = t_;                                                                                 \
    } while (0)
#else
#define CROSS_STAMP(slot)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KNB = 32;
constexpr float SLOPE = 0.1f;  // pointconv_util.py:10
constexpr int MAX_MAPPED_BATCH = 1024;  // batch elements a batch map may have (it is kept in LDS)

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// LeakyReLU with 0 < slope < 1 is max(v, slope*v): two instructions, same value for every finite v (and for +-0)
__device__ __forceinline__ float leaky(float v) { return mcp_max_raw(v, v * SLOPE); }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __uint_as_float(mcp_dpp<CTRL>(__float_as_uint(v)));
}
// dst <- max(src0 read through the DPP pattern, src1) in the lanes of the enabled banks; the other lanes keep dst.
// Raw instruction: fmaxf() would first canonicalise both operands (two extra v_max each), and the values here are
// never NaN-signalling.  s_nop 1 covers the two wait states a DPP read needs after a VALU write of its source.
#define MCP_MAX_DPP(dst, src0, src1, CTRL)                                                                              \
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 " CTRL : "+v"(dst) : "v"(src0), "v"(src1))

// Maximum over the 32 neighbours (the 32 lanes that share lane>>5) of all 16 accumulator registers, as a reduce-scatter:
// every butterfly step halves the number of live registers instead of reducing all of them everywhere (16 registers x
// 5 steps before; 16+8+4+3+1 instructions now).  Step over lane bit 4: v_permlane16_swap exchanges the odd rows of one
// register with the even rows of another, so one max folds two registers into one.  Bits 3 and 2: DPP row_ror:8 /
// row_half_mirror with bank masks choose per 4-lane bank which register of the pair is kept.  Bit 1 needs a select,
// bit 0 a plain exchange.  Result: lane l holds the maximum of register r(l) = 8*b4 + 4*b3 + 2*b2 + b1 (b_k = bit k
// of l), duplicated in lanes l and l^1.
__device__ __forceinline__ float scatter_max(const f32x16 &acc, int lane) {
    float v8[8], v4[4], v2[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[r]), __float_as_uint(acc[r + 8]), false, false);
        const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
        asm("v_max_f32 %0, %1, %2" : "=v"(v8[r]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // banks 0,1 (bit 3 clear) keep r, banks 2,3 keep r+4
        float d = v8[r + 4];
        MCP_MAX_DPP(d, v8[r], v8[r], "row_ror:8 row_mask:0xf bank_mask:0x3");
        MCP_MAX_DPP(d, v8[r + 4], d, "row_ror:8 row_mask:0xf bank_mask:0xc");
        v4[r] = d;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {  // banks 0,2 (bit 2 clear) keep r, banks 1,3 keep r+2; lane i pairs with 7-i of its half row
        float d = v4[r + 2];
        MCP_MAX_DPP(d, v4[r], v4[r], "row_half_mirror row_mask:0xf bank_mask:0x5");
        MCP_MAX_DPP(d, v4[r + 2], d, "row_half_mirror row_mask:0xf bank_mask:0xa");
        v2[r] = d;
    }
    const bool b1 = (lane & 2) != 0;
    float keep = b1 ? v2[1] : v2[0];
    const float send = b1 ? v2[0] : v2[1];
    MCP_MAX_DPP(keep, send, keep, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    MCP_MAX_DPP(keep, keep, keep, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    return keep;
}
__device__ __forceinline__ int scatter_reg(int lane) { return ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1); }

template <int D>
struct CrossLds {
    static constexpr int T = D / 32;
    static constexpr bool BF = true;      // D -> D layer on the split-bf16 path (D = 256 too: cross256_stream_kernel below)
    // f32 image [t_out][kquad][lane][4] = D*D floats; split image [t_out][kstep = 2T][piece = 3][lane] x uint4 = 1.5 D*D floats
    static constexpr int W_FLOATS = BF ? T * (2 * T) * 3 * 64 * 4 : T * (T * 16 / 4) * 64 * 4;
    static constexpr int POS_FLOATS = T * 2 * 64;               // [t][kstep][lane]
    static constexpr int B_FLOATS = T * 2 * 16;                 // [t][half][r]
    static constexpr int OFF_W = 0, OFF_POS = W_FLOATS, OFF_B = OFF_POS + POS_FLOATS;
    static constexpr int FLOATS = OFF_B + B_FLOATS;
};

// Packs (wpos, bpos, wmlp, bmlp) into the LDS image the kernel uses: done once per layer by the caller
// (mcp_cross_pack), so a workgroup stages its weights with plain float4 copies.
template <int D>
__global__ __launch_bounds__(256) void cross_pack_kernel(const float *__restrict__ wpos, const float *__restrict__ bpos,
                                                         const float *__restrict__ wmlp, const float *__restrict__ bmlp,
                                                         float *__restrict__ packed) {
    using L = CrossLds<D>;
    constexpr int T = L::T, KQ = T * 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < L::FLOATS; e += gridDim.x * 256) {
        float v;
        if (e < L::OFF_POS) {  // [t][q][lane][4]: k-step s = 4q + j -> input tile s >> 4, register s & 15
            if (L::BF) continue;  // written below as split pieces
            const int j = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % KQ, t = (e >> 8) / KQ;
            const int s = 4 * q + j, tin = s >> 4, r = s & 15;
            v = wmlp[(32 * t + (lane & 31)) * D + 32 * tin + chan_of(r, lane >> 5)];
        } else if (e < L::OFF_B) {  // [t][s][lane]: columns (dx,dy | dz,1)
            const int f = e - L::OFF_POS, lane = f & 63, s = (f >> 6) & 1, t = f >> 7;
            const int row = 32 * t + (lane & 31), c = 2 * s + (lane >> 5);
            v = c < 3 ? wpos[row * 3 + c] : bpos[row];
        } else {  // [t][h][r]
            const int f = e - L::OFF_B, r = f & 15, h = (f >> 4) & 1, t = f >> 5;
            v = bmlp[32 * t + chan_of(r, h)];
        }
        packed[e] = v;
    }
    if (L::BF) mcp_split_weights(reinterpret_cast<uint4 *>(packed + L::OFF_W), wmlp, D, T, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// The kernel is latency-bound (one point in flight per wave through idx -> gathers -> MFMA -> reduce), so residency
// matters more than unrolling: the output-tile loop is kept rolled and registers are capped for 3 (D=64) / 2 (D=128)
// waves per SIMD.
// Workgroup shape per D: the split image of D = 128 is 98 KB, one workgroup per CU, so that one carries 8 waves (2 per SIMD).
template <int D>
struct CrossShape {
    static constexpr int NW = D == 128 ? 8 : 4;
    static constexpr int WG_PER_CU = D == 64 ? 3 : 1;
    static constexpr int GRID = D == 64 ? 768 : 256;
};

template <int D, int SPLIT>
__global__ __launch_bounds__(64 * CrossShape<D>::NW, CrossShape<D>::WG_PER_CU) void cross_kernel(long long total, int n1, int n2, const float *__restrict__ xyz1,
                                                           const float *__restrict__ xyz2, const float *__restrict__ points1,
                                                           const float *__restrict__ points2, const int *__restrict__ idx,
                                                           const int *__restrict__ idx2, const int *__restrict__ bmap, int shared,
                                                           const float *__restrict__ packed, float *__restrict__ out) {
    using L = CrossLds<D>;
    constexpr int T = L::T, KQ = T * 4;  // k-quads per output tile (f32 image)
    constexpr int WAVES = CrossShape<D>::NW;
    constexpr int TO = T / SPLIT;        // output tiles of this workgroup: TO * split .. TO * split + TO - 1
    constexpr int WH = L::W_FLOATS / SPLIT, SMALL = L::POS_FLOATS + L::B_FLOATS;  // LDS image: [this half of W | pos | bias]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, split = blockIdx.y;
    static_assert(WH % 4 == 0 && SMALL % 4 == 0, "LDS image is copied as float4");
    for (int e = tid; e < WH / 4; e += 64 * WAVES)
        reinterpret_cast<float4 *>(lds)[e] = reinterpret_cast<const float4 *>(packed + L::OFF_W + (size_t)split * WH)[e];
    for (int e = tid; e < SMALL / 4; e += 64 * WAVES)
        reinterpret_cast<float4 *>(lds + WH)[e] = reinterpret_cast<const float4 *>(packed + L::OFF_POS)[e];
    if (bmap) {
        const int nb = (int)(total / n1);  // <= MCP_CROSS_MAX_MAPPED_BATCH (checked by the host)
        for (int e = tid; e < nb; e += 64 * WAVES) reinterpret_cast<int *>(lds + WH + SMALL + WAVES * D)[e] = bmap[e];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const float4 *wq = reinterpret_cast<const float4 *>(lds);
    const float *lpos = lds + WH, *lbias = lds + WH + L::POS_FLOATS;
    float *row1_lds = lds + WH + SMALL + wave * D;               // this wave's copy of the current point's points1 row
    const int *bmap_lds = reinterpret_cast<const int *>(lds + WH + SMALL + WAVES * D);

    // Software pipeline over the wave's points: the neighbour index of point i+2 and every load of point i+1 (coordinates, the
    // gathered rows of points2, the row of points1) are in flight while the D x D layer of point i runs on the MFMA pipe, so the
    // idx -> gather -> MFMA latency chain is paid once per wave, not once per point.  For that to hold nothing may WAIT between
    // issue and the next iteration: the prefetch step only issues loads and keeps the raw values (the coordinate differences are
    // formed where they are consumed), the batch map is read from LDS (a global lookup feeding an address would be a dependent
    // round trip in the middle of the issue sequence), and the points1 row -- the same 4 D bytes for all 32 neighbours -- is one
    // float4 per lane that goes through a wave-private LDS row into accumulator layout instead of 4 T broadcast loads per lane.
    // Which points a workgroup takes.  The hardware deals workgroups to the eight XCDs round-robin (workgroup i -> XCD i mod 8), and
    // every XCD has its own L2: with points dealt round-robin too, every L2 sees the gathered rows of every batch element.  Instead XCD x
    // takes the x-th eighth of the points -- whole batch elements when the batch is a multiple of 8 --, so the rows its waves gather
    // (points2 of 1-3 batch elements: 0.5-1.5 MB) stay in ITS L2.  `limit` replaces `total` in the loop below.
#ifndef MCP_NO_XCD_MAP
    const bool by_xcd = gridDim.x >= 8 && (gridDim.x & 7) == 0;
#else
    const bool by_xcd = false;
#endif
    const long long chunk = by_xcd ? (total + 7) / 8 : total;
    const long long stride = (long long)(by_xcd ? gridDim.x >> 3 : gridDim.x) * WAVES;
    const long long limit = by_xcd ? min(total, ((long long)(blockIdx.x & 7) + 1) * chunk) : total;
    long long p = by_xcd ? (long long)(blockIdx.x & 7) * chunk + (long long)(blockIdx.x >> 3) * WAVES + wave : (long long)blockIdx.x * WAVES + wave;
    float q2x = 0.f, q2y = 0.f, q2z = 0.f, p1x = 0.f, p1y = 0.f, p1z = 0.f;  // raw coordinates of the point in flight
    float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);                               // float4 number `lane` of its points1 row (lane < D/4)
    float4 rg[T][4];                                                          // its gathered points2 row, accumulator layout
    // bmap != NULL: the batch is a replication / selection of a smaller one (the three flow iterations of multiframe_attention see
    // the same features): batch element bb of the tensors flagged in `shared` (1: points1, 2: points2, 4: the first index list)
    // is read from element bmap[bb] of the unreplicated tensor -- nothing is copied.  xyz1 / xyz2 and the second list are per element.
    auto mapped = [&](int bb, int flag) -> long long { return (bmap && (shared & flag)) ? (long long)bmap_lds[bb] : (long long)bb; };
    // (batch element, point within it) of the wave's points, advanced by additions: a 64-bit division per point and per lookup
    // is ~130 instructions, as much issue time as the whole D = 64 MFMA block
    struct Pos { int bb, off; };
    const int stride_b = (int)(stride / n1), stride_o = (int)(stride % n1);
    auto advance = [&](Pos q) {
        q.bb += stride_b;
        q.off += stride_o;
        if (q.off >= n1) { q.off -= n1; ++q.bb; }
        return q;
    };
    auto fetch = [&](long long pp, Pos q, int id) {
        const float *q2 = xyz2 + ((long long)q.bb * n2 + id) * 3;
        q2x = q2[0]; q2y = q2[1]; q2z = q2[2];
        p1x = xyz1[pp * 3 + 0]; p1y = xyz1[pp * 3 + 1]; p1z = xyz1[pp * 3 + 2];
        const float4 *row2 = reinterpret_cast<const float4 *>(points2 + (mapped(q.bb, 2) * n2 + id) * D);
        const float4 *row1 = reinterpret_cast<const float4 *>(points1 + (mapped(q.bb, 1) * n1 + q.off) * D);
        r1 = row1[lane < D / 4 ? lane : 0];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) rg[t][g] = row2[(32 * t + 8 * g + 4 * h) >> 2];  // channels 32t + 8g + 4h .. +3 -> registers 4g .. 4g+3
    };
    // idx2 != NULL: the 16 feature-space and the 16 coordinate-space neighbours come as two (B,N1,16) lists
    auto nbr = [&](long long pp, Pos q) {
        if (!idx2) return idx[pp * KNB + col];
        const long long ps = mapped(q.bb, 4) * n1 + q.off;
        return col >= 16 ? idx2[pp * 16 + col - 16] : idx[ps * 16 + col];
    };
    long long pn = p + stride;
    Pos qp{(int)(p / n1), (int)(p % n1)};
    Pos qn = advance(qp), qnn = advance(qn);
    int idn = 0;
    if (p < limit) {
        fetch(p, qp, nbr(p, qp));
        if (pn < limit) idn = nbr(pn, qn);
    }
#ifdef MCP_CROSS_DIAG
    unsigned long long t_prev = 0;
    CROSS_STAMP(7);
#endif
    for (; p < limit; p = pn, pn += stride) {
        CROSS_STAMP(0);  // loop overhead / previous store
        f32x16 x0[L::BF ? 1 : T];
        McpSplit3 xs[L::BF ? 2 * T : 1];
        const float in0 = h ? q2y - p1y : q2x - p1x, in1 = h ? 1.0f : q2z - p1z;  // k-step 0: (dx,dy); k-step 1: (dz,1)
        if (lane < D / 4) reinterpret_cast<float4 *>(row1_lds)[lane] = r1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = reinterpret_cast<const float4 *>(row1_lds)[(32 * t + 8 * g + 4 * h) >> 2];
                acc[4 * g + 0] = a.x; acc[4 * g + 1] = a.y; acc[4 * g + 2] = a.z; acc[4 * g + 3] = a.w;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lpos[(t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lpos[(t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[4 * g + 0] = leaky(acc[4 * g + 0] + rg[t][g].x);
                acc[4 * g + 1] = leaky(acc[4 * g + 1] + rg[t][g].y);
                acc[4 * g + 2] = leaky(acc[4 * g + 2] + rg[t][g].z);
                acc[4 * g + 3] = leaky(acc[4 * g + 3] + rg[t][g].w);
            }
            if (L::BF) {
                xs[L::BF ? 2 * t : 0] = mcp_split_kstep(acc, 0);
                xs[L::BF ? 2 * t + 1 : 0] = mcp_split_kstep(acc, 1);
            } else {
                x0[L::BF ? 0 : t] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the row has been read: the next point's may be written over it
        CROSS_STAMP(1);  // x0 build: waits for the prefetched loads, pos MFMA, epilogue, split
        if (pn < limit) {
            fetch(pn, qn, idn);
            if (pn + stride < limit) idn = nbr(pn + stride, qnn);
        }
        qn = qnn;
        qnn = advance(qnn);
        CROSS_STAMP(2);  // issue of the next point's loads
#pragma unroll 1
        for (int tl = 0; tl < TO; ++tl) {
            const int t = TO * split + tl;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = lbias[(t * 2 + h) * 16 + r];
            if (L::BF) {
                const uint4 *ws = reinterpret_cast<const uint4 *>(lds) + (size_t)tl * (2 * T) * 3 * 64 + lane;
                acc = mcp_tile_split<(L::BF ? 2 * T : 1)>(ws, xs, acc);
            } else {
#pragma unroll
                for (int q4 = 0; q4 < KQ; ++q4) {
                    const float4 w = wq[(tl * KQ + q4) * 64 + lane];
                    const int tin = L::BF ? 0 : q4 >> 2, r0 = (q4 & 3) * 4;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x0[tin][r0 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x0[tin][r0 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x0[tin][r0 + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x0[tin][r0 + 3], acc, 0, 0, 0);
                }
            }
            CROSS_STAMP(3);  // D x D tile: bias + MFMA chain
            const float m = leaky(scatter_max(acc, lane));  // leaky is monotone: it commutes with the max
            if ((lane & 1) == 0) out[p * D + 32 * t + chan_of(scatter_reg(lane), h)] = m;
            CROSS_STAMP(4);  // neighbour max + store
        }
    }
}

// ---- D = 256 (cross3, pointconv_util.py:783-791) on the split-bf16 path -------------------------------------------------------
// The three bf16 pieces of the 256 x 256 layer are 393 KB: they do not fit the CU's LDS, and splitting the output channels over
// workgroups (round 3's form: two halves on the f32-input MFMA, 16/6 of the matrix time; a four-quarter split-bf16 build was no
// faster) makes every workgroup rebuild x0 -- the gather of 32 KB and ~1 500 vector instructions per point, which is what those
// forms spend their time on.  Here a point's x0 is built ONCE and stays in registers as the 16 split k-steps (192 VGPRs, one wave per
// SIMD), and the weights stream past it: the workgroup's four waves (one point each) walk the eight 32-channel output tiles in
// lockstep, tile t + 1's pieces (49 KB, the same for every point) are fetched from L2 into registers while tile t runs on the MFMA
// pipe and written to the other half of a double-buffered LDS tile behind it -- one barrier per tile, 393 KB of L2 reads per four
// points.  Same arithmetic as the D = 64 / 128 kernel (six bf16 MFMAs per 16 k-values, small terms first, bias as the initial
// accumulator), 768 MFMAs x 32 cycles per point against 1 024 x 64 on the f32-input MFMA.
constexpr int X256_T = 8, X256_KS = 16, X256_TILE_U4 = X256_KS * 3 * 64, X256_WAVES = 4;
__global__ __launch_bounds__(64 * X256_WAVES, 1) void cross256_stream_kernel(long long total, int n1, int n2, const float *__restrict__ xyz1,
                                                                          const float *__restrict__ xyz2, const float *__restrict__ points1,
                                                                          const float *__restrict__ points2, const int *__restrict__ idx,
                                                                          const int *__restrict__ idx2, const int *__restrict__ bmap, int shared,
                                                                          const float *__restrict__ packed, float *__restrict__ out) {
    using L = CrossLds<256>;
    constexpr int D = 256, T = X256_T, WAVES = X256_WAVES, SMALL = L::POS_FLOATS + L::B_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *wbuf = reinterpret_cast<uint4 *>(lds);                            // [2][X256_TILE_U4]
    float *small = lds + 2 * X256_TILE_U4 * 4;                               // pos | bias
    float *row1_all = small + SMALL;                                          // [WAVES][D]
    int *bmap_lds = reinterpret_cast<int *>(row1_all + WAVES * D);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const uint4 *wimg = reinterpret_cast<const uint4 *>(packed + L::OFF_W);   // [t_out][k-step][piece][lane]
    for (int e = tid; e < SMALL / 4; e += 64 * WAVES) reinterpret_cast<float4 *>(small)[e] = reinterpret_cast<const float4 *>(packed + L::OFF_POS)[e];
    if (bmap) {
        const int nb = (int)(total / n1);
        for (int e = tid; e < nb; e += 64 * WAVES) bmap_lds[e] = bmap[e];
    }
    for (int e = tid; e < X256_TILE_U4; e += 64 * WAVES) wbuf[e] = wimg[e];   // tile 0
    __syncthreads();
    const float *lpos = small, *lbias = small + L::POS_FLOATS;
    float *row1_lds = row1_all + wave * D;
    auto mapped = [&](int bb, int flag) -> long long { return (bmap && (shared & flag)) ? (long long)bmap_lds[bb] : (long long)bb; };

    const long long per_round = (long long)gridDim.x * WAVES;
    const long long first = (long long)blockIdx.x * WAVES;
    const int rounds = first < total ? (int)((total - first + per_round - 1) / per_round) : 0;  // the same for the four waves of the workgroup
    // No load pipeline across rounds here: the kernel is bound by the weight stream (402 MB of L2 reads per 4096 points), and a build
    // that kept the next round's gathered rows in flight (cross_kernel's scheme, 446 VGPRs) measured 125 vs 118 us alone, 174 vs 160 in the step.
    int cur = 0;
    for (int g = 0; g < rounds; ++g) {
        const long long pw = first + (long long)g * per_round + wave;
        const bool live = pw < total;
        const long long p = live ? pw : total - 1;  // a wave without a point works on the last one and does not store
        const int bb = (int)(p / n1), off = (int)(p - (long long)bb * n1);
        int id;
        if (!idx2) {
            id = idx[p * KNB + col];
        } else {
            const long long ps = mapped(bb, 4) * n1 + off;
            id = col >= 16 ? idx2[p * 16 + col - 16] : idx[ps * 16 + col];
        }
        const float *q2 = xyz2 + ((long long)bb * n2 + id) * 3;
        const float dx = q2[0] - xyz1[p * 3 + 0], dy = q2[1] - xyz1[p * 3 + 1], dz = q2[2] - xyz1[p * 3 + 2];
        const float in0 = h ? dy : dx, in1 = h ? 1.0f : dz;  // k-step 0: (dx,dy); k-step 1: (dz,1)
        const float4 *row2 = reinterpret_cast<const float4 *>(points2 + (mapped(bb, 2) * n2 + id) * D);
        const float4 *row1 = reinterpret_cast<const float4 *>(points1 + (mapped(bb, 1) * n1 + off) * D);
        reinterpret_cast<float4 *>(row1_lds)[lane] = row1[lane];  // D / 4 = 64 float4: one per lane
        __builtin_amdgcn_wave_barrier();
        McpSplit3 xs[2 * T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x16 acc;
            float4 rg[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 a = reinterpret_cast<const float4 *>(row1_lds)[(32 * t + 8 * gq + 4 * h) >> 2];
                rg[gq] = row2[(32 * t + 8 * gq + 4 * h) >> 2];
                acc[4 * gq + 0] = a.x; acc[4 * gq + 1] = a.y; acc[4 * gq + 2] = a.z; acc[4 * gq + 3] = a.w;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lpos[(t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lpos[(t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                acc[4 * gq + 0] = leaky(acc[4 * gq + 0] + rg[gq].x);
                acc[4 * gq + 1] = leaky(acc[4 * gq + 1] + rg[gq].y);
                acc[4 * gq + 2] = leaky(acc[4 * gq + 2] + rg[gq].z);
                acc[4 * gq + 3] = leaky(acc[4 * gq + 3] + rg[gq].w);
            }
            xs[2 * t + 0] = mcp_split_kstep(acc, 0);
            xs[2 * t + 1] = mcp_split_kstep(acc, 1);
        }
        __builtin_amdgcn_wave_barrier();  // the points1 row has been read: the next round may overwrite it
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            // the next tile's pieces (the first tile again after the last: the next round starts with it) on their way while this one runs
            const bool more = t + 1 < T || g + 1 < rounds;
            const uint4 *nsrc = wimg + (size_t)((t + 1) & (T - 1)) * X256_TILE_U4 + tid;
            uint4 nx[X256_TILE_U4 / (64 * X256_WAVES)];
#pragma unroll
            for (int i = 0; i < X256_TILE_U4 / (64 * WAVES); ++i) nx[i] = more ? nsrc[i * 64 * WAVES] : make_uint4(0u, 0u, 0u, 0u);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = lbias[(t * 2 + h) * 16 + r];
            acc = mcp_tile_split<2 * T>(wbuf + (size_t)cur * X256_TILE_U4 + lane, xs, acc);
            const float m = leaky(scatter_max(acc, lane));  // leaky is monotone: it commutes with the max
            if (live && (lane & 1) == 0) out[p * D + 32 * t + chan_of(scatter_reg(lane), h)] = m;
            uint4 *ndst = wbuf + (size_t)(cur ^ 1) * X256_TILE_U4 + tid;
#pragma unroll
            for (int i = 0; i < X256_TILE_U4 / (64 * WAVES); ++i) ndst[i * 64 * WAVES] = nx[i];
            __syncthreads();
            cur ^= 1;
        }
    }
}

int launch_cross256(long long total, int n1, int n2, const float *xyz1, const float *xyz2, const float *points1, const float *points2, const int *idx,
                    const int *idx2, const int *bmap, int shared, const float *packed, float *out, hipStream_t s) {
    using L = CrossLds<256>;
    const size_t lds = (size_t)2 * X256_TILE_U4 * 16 + (size_t)(L::POS_FLOATS + L::B_FLOATS + X256_WAVES * 256 + (bmap ? MAX_MAPPED_BATCH : 0)) * sizeof(float);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cross256_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    const long long want = (total + X256_WAVES - 1) / X256_WAVES;
    const unsigned grid = (unsigned)max(1LL, min(want, 256LL));  // one resident workgroup per CU
    hipLaunchKernelGGL(cross256_stream_kernel, dim3(grid), dim3(64 * X256_WAVES), lds, s, total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, bmap, shared,
                       packed, out);
    return mcp_launch_status();
}

template <int D, int SPLIT>
int launch_cross(long long total, int n1, int n2, const float *xyz1, const float *xyz2, const float *points1, const float *points2,
                 const int *idx, const int *idx2, const int *bmap, int shared, const float *packed, float *out, hipStream_t s) {
    using L = CrossLds<D>;
    constexpr int WAVES = CrossShape<D>::NW;
    // weights | pos + bias | one points1 row per wave | the batch map (only when one is passed)
    const size_t lds = (L::W_FLOATS / SPLIT + L::POS_FLOATS + L::B_FLOATS + WAVES * D + (bmap ? MAX_MAPPED_BATCH : 0)) * sizeof(float);
    auto kern = cross_kernel<D, SPLIT>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    // at least 8 points per wave so the weight staging is amortised
    const long long want = (total + WAVES * 8 - 1) / (WAVES * 8);
    // persistent-style grid = exactly the resident slots (256 CUs x 3 or 2 workgroups, see __launch_bounds__): a larger
    // grid leaves a partly filled second round of workgroups
    const unsigned grid = (unsigned)max(1LL, min(want, (long long)(CrossShape<D>::GRID / SPLIT)));
    hipLaunchKernelGGL(kern, dim3(grid, SPLIT), dim3(64 * WAVES), lds, s, total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, bmap, shared, packed, out);
    return mcp_launch_status();
}

}  // namespace

MCP_EXPORT int mcp_cross_packed_floats(int d) {
    return d == 64 ? CrossLds<64>::FLOATS : d == 128 ? CrossLds<128>::FLOATS : d == 256 ? CrossLds<256>::FLOATS : 0;
}

MCP_EXPORT int mcp_cross_pack(int d, const float *wpos, const float *bpos, const float *wmlp, const float *bmlp, float *packed,
                              mcp_stream_t stream) {
    MCP_CHECK_ARGS(wpos && bpos && wmlp && bmlp && packed);
    if (d != 64 && d != 128 && d != 256) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (d == 64) hipLaunchKernelGGL(cross_pack_kernel<64>, dim3(16), dim3(256), 0, s, wpos, bpos, wmlp, bmlp, packed);
    else if (d == 128) hipLaunchKernelGGL(cross_pack_kernel<128>, dim3(64), dim3(256), 0, s, wpos, bpos, wmlp, bmlp, packed);
    else hipLaunchKernelGGL(cross_pack_kernel<256>, dim3(256), dim3(256), 0, s, wpos, bpos, wmlp, bmlp, packed);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_cross_volume(int b, int n1, int n2, int d, int k, const float *xyz1, const float *xyz2, const float *points1,
                                const float *points2, const int *idx, const int *idx2, const int *bmap, int shared, const float *packed, float *out,
                                mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n1 > 0 && n2 > 0 && xyz1 && xyz2 && points1 && points2 && idx && packed && out);
    if (k != KNB || (d != 64 && d != 128 && d != 256)) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)points1) | ((uintptr_t)points2) | ((uintptr_t)out) | ((uintptr_t)packed)) & 15) return MCP_ERR_BAD_ARG;
    if (bmap && b > MAX_MAPPED_BATCH) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n1;
    mcp_prof_begin(MCP_KERNEL_CROSS, s);
    const int rc = d == 64    ? launch_cross<64, 1>(total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, bmap, shared, packed, out, s)
                   : d == 128 ? launch_cross<128, 1>(total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, bmap, shared, packed, out, s)
                              : launch_cross256(total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, bmap, shared, packed, out, s);
    mcp_prof_end(MCP_KERNEL_CROSS, s);
    return rc;
}

#ifdef MCP_CROSS_DIAG
MCP_EXPORT int mcp_cross_diag_read(unsigned long long *out8) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cross_diag), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cross_diag), z, sizeof(z));
    return (int)e;
}
#endif
