// cross_grad.hip -- backward of the fused cost volume (mcp_cross_volume; CrossLayerLightFeatCosine.cross, pointconv_util.py:750-781,
// BidirectionalLayerFeatCosine.cross :894-922, FlowEmbeddingLayer.forward :1126-1161) for gfx950.  The reference differentiates
//     u = points2[idx] + points1 + Conv2d_{3->D}(xyz2[idx] - xyz1),  x = LeakyReLU(u),  z = Conv2d_{D->D}(x),  out = max_j LeakyReLU(z_j)
// with autograd over three materialised (B,D,32,N1) tensors.  Here one wave owns one point at a time, re-evaluates x and z in the
// forward's MFMA layout (neighbours on the MFMA column, lane & 31) and back-propagates inside the kernel:
//   * arg-max neighbour of every channel: an all-reduce max over the 32 lanes of a lane half on order-preserving integer keys, ties
//     resolved to the lowest neighbour position with the comparison's own lane mask (the two 16-neighbour lists overlap, so exact
//     ties are common) -- dz is non-zero only there: dz_j[c] = g[c] LeakyReLU'(z_j[c]) [j = j*(c)];
//   * dx = Wmlp^T dz on the split-bf16 MFMA path, the gradient tile chained as B operand exactly like the forward's activations;
//     du = dx LeakyReLU'(u);  d_rows (B,N1,32,D) = du and d_dir (B,N1,32,3) = Wpos^T du leave for the caller's deterministic
//     segmented scatters into dL/dpoints2 and dL/dxyz2;  dL/dpoints1 = sum_j du_j, dL/dxyz1 = -sum_j d_dir_j;
//   * weight gradients contract over the NEIGHBOUR axis, so dz, x and du pass once through a per-wave LDS tile (written in
//     accumulator layout, read back with 8 consecutive neighbours per lane): dWmlp += dz . x^T on the MFMA (both operands split
//     three ways), dbmlp, dWpos, dbpos and dL/dpoints1 as in-lane sums with lane = channel.
//   Weight gradients accumulate in registers over a wave's points; waves are added in wave order through LDS, workgroups in
//   workgroup order by a second kernel: every sum has a fixed order, the gradients repeat bit for bit.
#include "common.h"
#include "mfma_grad.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KNB = 32;
constexpr float SLOPE = 0.1f;  // pointconv_util.py:10
constexpr int TS = MCP_TS;

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float leaky(float v) { return mcp_max_raw(v, v * SLOPE); }  // as the forward (cross.hip)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int CTRL>
__device__ __forceinline__ uint32_t max_dpp(uint32_t v) {
    const uint32_t o = mcp_dpp<CTRL>(v);
    return v > o ? v : o;
}
// maximum over the 32 lanes that share lane >> 5, in every lane
__device__ __forceinline__ uint32_t half_max_u32(uint32_t v) {
    v = max_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = max_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = max_dpp<0x141>(v);  // row_half_mirror
    v = max_dpp<0x140>(v);  // row_mirror
    const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // odd rows of one copy <-> even rows of the other
    return sw[0] > sw[1] ? sw[0] : sw[1];
}

template <int D>
struct GradShape {
    static constexpr int T = D / 32;
    static constexpr int WAVES = 4;
    // D = 128: the 256 accumulator registers of dWmlp do not fit beside the rest, so the workgroups come in three ROLES that all walk
    // every point: the data role (x, all of z, dx, the per-neighbour outputs, dWpos: 384 MFMAs per point) and two dWmlp roles that
    // own rows 0..63 / 64..127 and re-evaluate only what those need (x and their two z tiles: 192 MFMAs) -- grid shares 2 : 1 : 1.
    // D = 64: one role does both.
    static constexpr int ROLES = D == 64 ? 1 : 3;
    static constexpr bool WT_LDS = D == 64;  // the Wmlp^T image: in LDS beside Wmlp (D = 64), or read from the workspace through L2 (98 KB at D = 128)
    // weight-gradient vector (floats): dWpos (D,3) | dbpos (D) | dWmlp (D,D) | dbmlp (D)
    static constexpr int G_WP = 0, G_BP = 3 * D, G_WM = 4 * D, G_BM = 4 * D + D * D, G_FLOATS = G_BM + D;
    // LDS, floats: pos image [t][s][lane] | bmlp [t][h][r] | Wpos rows [t][h][r][4]
    static constexpr int L_POS = 0, L_B = T * 2 * 64, L_WPR = L_B + T * 32, L_F32 = L_WPR + T * 32 * 4;
    static constexpr int W_U4 = T * (2 * T) * 3 * 64;  // uint4 per split image (Wmlp, Wmlp^T)
    static constexpr int W_LDS_U4 = WT_LDS ? 2 * W_U4 : W_U4;
    // per wave (floats): tile [64][TS] (two 32-channel tiles at a time) | directions [32][4]
    static constexpr int S_T = 0, S_DIR = 64 * TS, S_FLOATS = S_DIR + 128;
    static constexpr size_t LDS_BYTES = (size_t)L_F32 * 4 + (size_t)W_LDS_U4 * 16 + (size_t)WAVES * S_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((size_t)G_FLOATS * 4 <= (size_t)L_F32 * 4 + (size_t)W_LDS_U4 * 16, "the reduction buffer overlays the weight images");
};

__global__ __launch_bounds__(256) void transposed_image_kernel(uint4 *dst, const float *__restrict__ w, int d) {
    mcp_split_weights_transposed(dst, w, d, d, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// One workgroup of a role: points first, first + step, ... below limit.  DATA: dx, the per-neighbour outputs, dWpos;
// OWN >= 0: rows 32 OWN .. 32 OWN + 63 of dWmlp (and of dbmlp).
template <int D, bool DATA, int OWN>
__device__ __forceinline__ void cross_grad_body(float *lds, long long total, int n1, int n2, long long first, long long step, long long limit,
                                                const float *__restrict__ xyz1,
                                                const float *__restrict__ xyz2, const float *__restrict__ points1, const float *__restrict__ points2,
                                                const int *__restrict__ idx, const int *__restrict__ idx2, const float *__restrict__ wpos,
                                                const float *__restrict__ bpos, const float *__restrict__ wmlp, const float *__restrict__ bmlp,
                                                const uint4 *__restrict__ wt_global, const float *__restrict__ gout, float *__restrict__ d_xyz1,
                                                float *__restrict__ d_dir, float *__restrict__ d_points1, float *__restrict__ d_rows,
                                                float *__restrict__ partial_row) {
    using S = GradShape<D>;
    constexpr int T = S::T, WAVES = S::WAVES;
    constexpr bool HAS_OWN = OWN >= 0;
    constexpr int OWN0 = HAS_OWN ? OWN : 0;
    static_assert(!DATA || OWN <= 0, "with DATA the own tiles must come first in the z loop");
    uint4 *wms = reinterpret_cast<uint4 *>(lds + S::L_F32);
    uint4 *wmts_lds = wms + S::W_U4;
    const uint4 *wmts = S::WT_LDS ? wmts_lds : wt_global;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *scr = reinterpret_cast<float *>(wms + S::W_LDS_U4) + wave * S::S_FLOATS;
    float *tb = scr + S::S_T;
    float4 *dirb = reinterpret_cast<float4 *>(scr + S::S_DIR);

    for (int e = tid; e < T * 128; e += 64 * WAVES) {  // [t][s][lane]: columns (dx,dy | dz,1) of [Wpos | bpos], as cross_pack_kernel
        const int l = e & 63, s = (e >> 6) & 1, t = e >> 7;
        const int row = 32 * t + (l & 31), c = 2 * s + (l >> 5);
        lds[S::L_POS + e] = c < 3 ? wpos[row * 3 + c] : bpos[row];
        const int k = e & 3, r = (e >> 2) & 15, hh = (e >> 6) & 1;  // Wpos rows in accumulator order, for d_dir = Wpos^T du
        lds[S::L_WPR + e] = k < 3 ? wpos[(32 * t + chan_of(r, hh)) * 3 + k] : 0.f;
    }
    for (int e = tid; e < T * 32; e += 64 * WAVES) {
        const int r = e & 15, hh = (e >> 4) & 1, t = e >> 5;
        lds[S::L_B + e] = bmlp[32 * t + chan_of(r, hh)];
    }
    mcp_split_weights(wms, wmlp, D, T, tid, 64 * WAVES);
    if (S::WT_LDS) mcp_split_weights_transposed(wmts_lds, wmlp, D, D, tid, 64 * WAVES);
    __syncthreads();

    f32x16 dWa[2][HAS_OWN ? T : 1];  // dWmlp tiles: row = 32 (OWN + a) + chan_of(r, h), column = 32 nt + col
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < (HAS_OWN ? T : 1); ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dWa[a][b][r] = 0.f;
    float dbm[2] = {0.f, 0.f};  // lane = channel 32 (OWN + a) + col; the two lane halves hold the two 8-neighbour groups
    float dbp[T], dWp[T][3];    // lane = channel 32 mt + col
#pragma unroll
    for (int t = 0; t < T; ++t) { dbp[t] = 0.f; dWp[t][0] = 0.f; dWp[t][1] = 0.f; dWp[t][2] = 0.f; }
    const uint32_t lower_lanes = (1u << col) - 1u;

    // D = 64 (316 registers, one wave per SIMD: nothing hides a gather): the point's operands -- its neighbour rows, its own row, its
    // output gradient, the coordinates -- are loaded one iteration AHEAD into 100 registers, the neighbour indices two ahead (the rows'
    // addresses depend on them); round 5 counters: 43 % of this kernel's wave time was s_waitcnt.  D = 128 has no registers to spare.
    constexpr bool PIPE = D == 64;
    struct Pre {
        float q[3], c[3];
        float4 r1[4 * T], r2[4 * T], gr[4 * T];
    };
    auto load_id = [&](long long pp) -> int { return idx2 ? (col >= 16 ? idx2[pp * 16 + col - 16] : idx[pp * 16 + col]) : idx[pp * KNB + col]; };
    auto fetch = [&](long long pp, int idn, Pre &o) {
        const long long bn = mcp_div(pp, n1, mcp_fits32(total));
        const float *q2n = xyz2 + ((long long)bn * n2 + idn) * 3;
        const float4 *r2n = reinterpret_cast<const float4 *>(points2 + ((long long)bn * n2 + idn) * D);
        const float4 *r1n = reinterpret_cast<const float4 *>(points1 + pp * D);
        const float4 *grn = reinterpret_cast<const float4 *>(gout + pp * D);
#pragma unroll
        for (int k = 0; k < 3; ++k) { o.q[k] = q2n[k]; o.c[k] = xyz1[pp * 3 + k]; }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int at = (32 * t + 8 * g + 4 * h) >> 2;
                o.r1[4 * t + g] = r1n[at];
                o.r2[4 * t + g] = r2n[at];
                o.gr[4 * t + g] = grn[at];
            }
    };
    Pre cur;
    int id_ahead = 0;
    if (PIPE) {
        const long long p0 = first + wave;
        if (p0 < limit) {
            fetch(p0, load_id(p0), cur);
            if (p0 + step < limit) id_ahead = load_id(p0 + step);
        }
    }

    for (long long p = first + wave; p < limit; p += step) {
        const long long bb = mcp_div(p, n1, mcp_fits32(total));
        Pre nxt;
        int id_ahead2 = 0;
        int id = 0;
        if (PIPE) {
            if (p + step < limit) fetch(p + step, id_ahead, nxt);          // wave-uniform
            if (p + 2 * step < limit) id_ahead2 = load_id(p + 2 * step);
        } else {
            id = load_id(p);
        }
        const float *q2 = xyz2 + ((long long)bb * n2 + id) * 3;
        const float dx = PIPE ? cur.q[0] - cur.c[0] : q2[0] - xyz1[p * 3 + 0], dy = PIPE ? cur.q[1] - cur.c[1] : q2[1] - xyz1[p * 3 + 1],
                    dzc = PIPE ? cur.q[2] - cur.c[2] : q2[2] - xyz1[p * 3 + 2];
        const float in0 = h ? dy : dx, in1 = h ? 1.0f : dzc;
        const float4 *row2 = reinterpret_cast<const float4 *>(points2 + ((long long)bb * n2 + id) * D);
        const float4 *row1 = reinterpret_cast<const float4 *>(points1 + p * D);
        const float4 *grow = reinterpret_cast<const float4 *>(gout + p * D);
        // ---- the forward again: x = LeakyReLU(points2[idx] + points1 + pos) ----
        f32x16 x[T], du[DATA ? T : 1];
        McpSplit3 as[2][2];
        {
            McpSplit3 xs[2 * T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x16 acc;
                float4 rg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = PIPE ? cur.r1[4 * t + g] : row1[(32 * t + 8 * g + 4 * h) >> 2];
                    rg[g] = PIPE ? cur.r2[4 * t + g] : row2[(32 * t + 8 * g + 4 * h) >> 2];
                    acc[4 * g + 0] = a.x; acc[4 * g + 1] = a.y; acc[4 * g + 2] = a.z; acc[4 * g + 3] = a.w;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[S::L_POS + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[S::L_POS + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[4 * g + 0] = leaky(acc[4 * g + 0] + rg[g].x);
                    acc[4 * g + 1] = leaky(acc[4 * g + 1] + rg[g].y);
                    acc[4 * g + 2] = leaky(acc[4 * g + 2] + rg[g].z);
                    acc[4 * g + 3] = leaky(acc[4 * g + 3] + rg[g].w);
                }
                x[t] = acc;
                xs[2 * t + 0] = mcp_split_kstep(acc, 0);
                xs[2 * t + 1] = mcp_split_kstep(acc, 1);
            }
            if (DATA) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) du[DATA ? t : 0][r] = 0.f;
            }
            // z = Wmlp x + b tile by tile (this role's own tiles first); dz_j[c] = g[c] LeakyReLU'(z_j[c]) at the arg-max neighbour
            // (lowest list position among equals), 0 elsewhere.  Own tiles go into the transposition tile (the A operand of
            // dWmlp); with DATA every tile's dz is folded into dx = Wmlp^T dz right away, as two k-steps of each output tile.
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int tz = 0; tz < (DATA ? T : 2); ++tz) {
                const int t = DATA ? tz : OWN0 + tz;      // DATA: OWN <= 0, so own tiles (0, 1) come first
                const bool own = HAS_OWN && (t == OWN0 || t == OWN0 + 1);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[S::L_B + (t * 2 + h) * 16 + r];
                acc = mcp_tile_split<2 * T>(wms + (size_t)t * (2 * T) * 3 * 64 + lane, xs, acc);
                f32x16 dzt;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 gv = PIPE ? cur.gr[4 * t + g] : grow[(32 * t + 8 * g + 4 * h) >> 2];
                    const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const float zv = acc[r];
                        const uint32_t key = mcp_ord(zv);
                        const bool top = key == half_max_u32(key);
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(top);
                        const uint32_t mine = h ? (uint32_t)(mask >> 32) : (uint32_t)mask;
                        const bool winner = top && (mine & lower_lanes) == 0u;
                        dzt[r] = winner ? (zv > 0.f ? gq[i] : SLOPE * gq[i]) : 0.f;
                        if (own) tb[(32 * (t - OWN0) + chan_of(r, h)) * TS + col] = dzt[r];
                    }
                }
                if (DATA) {
                    const McpSplit3 z0 = mcp_split_kstep(dzt, 0), z1 = mcp_split_kstep(dzt, 1);
#pragma unroll
                    for (int to = 0; to < T; ++to) {
                        const uint4 *wk = wmts + ((size_t)to * (2 * T) + 2 * t) * 3 * 64 + lane;
                        du[DATA ? to : 0] = mcp_mfma_split(wk, z0, du[DATA ? to : 0]);
                        du[DATA ? to : 0] = mcp_mfma_split(wk + 3 * 64, z1, du[DATA ? to : 0]);
                    }
                }
                if (HAS_OWN && tz == 1) {  // both own tiles are in the transposition tile: dWmlp's A operand, dbmlp
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            float v[8];
                            mcp_read8(tb + (32 * a + col) * TS + 16 * ks + 8 * h, v);
                            dbm[a] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                            as[a][ks] = mcp_split8(v);
                        }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        // ---- dWmlp (own rows) += dz . x^T, x through the transposition tile two 32-channel tiles at a time ----
#pragma unroll
        for (int hb = 0; hb < (HAS_OWN ? T / 2 : 0); ++hb) {
            __builtin_amdgcn_wave_barrier();
            mcp_write_tile(tb, x[2 * hb], col, h);
            mcp_write_tile(tb + 32 * TS, x[2 * hb + 1], col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * tt + col) * TS + 16 * ks + 8 * h, v);
                    const McpSplit3 bs = mcp_split8(v);
#pragma unroll
                    for (int a = 0; a < 2; ++a) dWa[a][HAS_OWN ? 2 * hb + tt : 0] = mcp_mfma_split6(as[a][ks], bs, dWa[a][HAS_OWN ? 2 * hb + tt : 0]);
                }
        }
        if (DATA) {
            // ---- du = LeakyReLU'(u) . dx;  per-neighbour outputs: d_rows = du, d_dir = Wpos^T du ----
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) du[DATA ? t : 0][r] = x[t][r] > 0.f ? du[DATA ? t : 0][r] : SLOPE * du[DATA ? t : 0][r];  // x = LeakyReLU(u) has u's sign
            float4 *orow = reinterpret_cast<float4 *>(d_rows + (p * KNB + col) * D);
            float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16 &dt = du[DATA ? t : 0];
                    orow[(32 * t + 8 * g + 4 * h) >> 2] = make_float4(dt[4 * g + 0], dt[4 * g + 1], dt[4 * g + 2], dt[4 * g + 3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 w = reinterpret_cast<const float4 *>(lds + S::L_WPR)[(t * 2 + h) * 16 + 4 * g + i];
                        ax = __builtin_fmaf(w.x, dt[4 * g + i], ax);
                        ay = __builtin_fmaf(w.y, dt[4 * g + i], ay);
                        az = __builtin_fmaf(w.z, dt[4 * g + i], az);
                    }
                }
            ax += __shfl_xor(ax, 32);
            ay += __shfl_xor(ay, 32);
            az += __shfl_xor(az, 32);
            if (h == 0) {
                float *o = d_dir + (p * KNB + col) * 3;
                o[0] = ax; o[1] = ay; o[2] = az;
            }
            const float sx = wave_sum(ax), sy = wave_sum(ay), sz = wave_sum(az);  // every neighbour sits in both lane halves
            if (lane == 0) {
                d_xyz1[p * 3 + 0] = -0.5f * sx;
                d_xyz1[p * 3 + 1] = -0.5f * sy;
                d_xyz1[p * 3 + 2] = -0.5f * sz;
            }
            // ---- sums over the neighbours with lane = channel: dL/dpoints1 (= this point's share of dbpos), dWpos ----
#pragma unroll
            for (int hb = 0; hb < T / 2; ++hb) {
                __builtin_amdgcn_wave_barrier();
                mcp_write_tile(tb, du[DATA ? 2 * hb : 0], col, h);
            mcp_write_tile(tb + 32 * TS, du[DATA ? 2 * hb + 1 : 0], col, h);
                if (hb == 0 && h == 0) dirb[col] = make_float4(dx, dy, dzc, 0.f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int mt = 2 * hb + tt;
                    float rowsum = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float v[8];
                        mcp_read8(tb + (32 * tt + col) * TS + 16 * ks + 8 * h, v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float4 dj = dirb[16 * ks + 8 * h + i];
                            dWp[mt][0] = __builtin_fmaf(v[i], dj.x, dWp[mt][0]);
                            dWp[mt][1] = __builtin_fmaf(v[i], dj.y, dWp[mt][1]);
                            dWp[mt][2] = __builtin_fmaf(v[i], dj.z, dWp[mt][2]);
                            rowsum += v[i];
                        }
                    }
                    rowsum += __shfl_xor(rowsum, 32);
                    if (h == 0) {
                        d_points1[p * D + 32 * mt + col] = rowsum;
                        dbp[mt] += rowsum;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (PIPE) {
            cur = nxt;
            id_ahead = id_ahead2;
        }
    }

    // ---- the workgroup's partial vector (zero where another role owns the entry): waves added in wave order through LDS ----
    __syncthreads();
    float *red = lds;
    for (int e = tid; e < S::G_FLOATS; e += 64 * WAVES) red[e] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < (HAS_OWN ? 2 : 0); ++a) {
#pragma unroll
                for (int nt = 0; nt < T; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[S::G_WM + (32 * (OWN0 + a) + chan_of(r, h)) * D + 32 * nt + col] += dWa[a][HAS_OWN ? nt : 0][r];
                const float vb = dbm[a] + __shfl_xor(dbm[a], 32);
                if (h == 0) red[S::G_BM + 32 * (OWN0 + a) + col] += vb;
            }
            if (DATA) {
#pragma unroll
                for (int mt = 0; mt < T; ++mt) {
                    float vw[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) vw[k] = dWp[mt][k] + __shfl_xor(dWp[mt][k], 32);
                    if (h == 0) {
                        const int c = 32 * mt + col;
                        red[S::G_BP + c] += dbp[mt];
#pragma unroll
                        for (int k = 0; k < 3; ++k) red[S::G_WP + c * 3 + k] += vw[k];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < S::G_FLOATS; e += 64 * WAVES) partial_row[e] = red[e];
}

// Workgroups 0 .. g0-1 take the data role, the next g1 dWmlp's rows 0..63, the rest rows 64..127 (D = 128); every role walks all points.
template <int D>
__global__ __launch_bounds__(64 * GradShape<D>::WAVES, 1) void cross_grad_kernel(
    long long total, int n1, int n2, int g0, int g1, const float *__restrict__ xyz1, const float *__restrict__ xyz2, const float *__restrict__ points1,
    const float *__restrict__ points2, const int *__restrict__ idx, const int *__restrict__ idx2, const float *__restrict__ wpos,
    const float *__restrict__ bpos, const float *__restrict__ wmlp, const float *__restrict__ bmlp, const uint4 *__restrict__ wt_global,
    const float *__restrict__ gout, float *__restrict__ d_xyz1, float *__restrict__ d_dir, float *__restrict__ d_points1, float *__restrict__ d_rows,
    float *__restrict__ partial) {
    using S = GradShape<D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *prow = partial + (size_t)blockIdx.x * S::G_FLOATS;
    const int bx = (int)blockIdx.x;
#define MCP_CROSS_GRAD_ARGS xyz1, xyz2, points1, points2, idx, idx2, wpos, bpos, wmlp, bmlp, wt_global, gout, d_xyz1, d_dir, d_points1, d_rows, prow
    // the points of workgroup k of a role with g workgroups that starts at workgroup `base` of the launch: XCD x (= blockIdx mod 8) takes
    // the x-th eighth of the points when the role's workgroups are spread evenly over the XCDs (common.h: mcp_units_by_xcd), else round-robin
    long long first, step, limit = total;
    auto deal = [&](int k, int g, int base) {
        first = (long long)k * S::WAVES;
        step = (long long)g * S::WAVES;
#ifndef MCP_NO_XCD_MAP
        if (g >= 8 && (g & 7) == 0 && (base & 7) == 0) {
            const long long steps = (total + S::WAVES - 1) / S::WAVES, chunk = ((steps + 7) / 8) * S::WAVES, x = k & 7;
            first = x * chunk + (long long)(k >> 3) * S::WAVES;
            step = (long long)(g >> 3) * S::WAVES;
            limit = (x + 1) * chunk < total ? (x + 1) * chunk : total;
        }
#endif
    };
    if (S::ROLES == 1) {
        deal(bx, g0, 0);
        cross_grad_body<D, true, 0>(lds, total, n1, n2, first, step, limit, MCP_CROSS_GRAD_ARGS);
    } else if (bx < g0) {
        deal(bx, g0, 0);
        cross_grad_body<D, true, -1>(lds, total, n1, n2, first, step, limit, MCP_CROSS_GRAD_ARGS);
    } else if (bx < g0 + g1) {
        deal(bx - g0, g1, g0);
        cross_grad_body<D, false, 0>(lds, total, n1, n2, first, step, limit, MCP_CROSS_GRAD_ARGS);
    } else {
        deal(bx - g0 - g1, (int)gridDim.x - g0 - g1, g0 + g1);
        cross_grad_body<D, false, (S::ROLES > 1 ? 2 : 0)>(lds, total, n1, n2, first, step, limit, MCP_CROSS_GRAD_ARGS);
    }
#undef MCP_CROSS_GRAD_ARGS
}

// out[e] = sum over the workgroups' partial vectors, in workgroup order
__global__ __launch_bounds__(256) void cross_grad_reduce_kernel(const float *__restrict__ partial, int parts, int floats, float *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= floats) return;
    float s = 0.f;
    for (int g = 0; g < parts; ++g) s += partial[(size_t)g * floats + e];
    out[e] = s;
}

// workgroups per role (g0 data, g1 each dWmlp half): one resident workgroup per CU, points dealt out statically within a role
template <int D>
void grad_grid(long long total, unsigned *g0, unsigned *g1) {
    using S = GradShape<D>;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = (total + S::WAVES - 1) / S::WAVES;
    const long long a = S::ROLES == 1 ? cus : (cus / 2 > 0 ? cus / 2 : 1), b = S::ROLES == 1 ? 0 : (cus / 4 > 0 ? cus / 4 : 1);
    *g0 = (unsigned)(want < a ? want : a);
    *g1 = (unsigned)(want < b ? want : b);
}
template <int D>
size_t workspace_bytes(long long total) {
    using S = GradShape<D>;
    unsigned g0, g1;
    grad_grid<D>(total, &g0, &g1);
    return (size_t)(g0 + (S::ROLES - 1) * g1) * S::G_FLOATS * sizeof(float) + (S::WT_LDS ? 0 : (size_t)S::W_U4 * 16);
}

template <int D>
int launch_cross_grad(long long total, int n1, int n2, const float *xyz1, const float *xyz2, const float *points1, const float *points2, const int *idx,
                      const int *idx2, const float *wpos, const float *bpos, const float *wmlp, const float *bmlp, const float *gout, float *d_xyz1,
                      float *d_dir, float *d_points1, float *d_rows, float *d_weights, void *workspace, hipStream_t s) {
    using S = GradShape<D>;
    auto kern = cross_grad_kernel<D>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    unsigned g0, g1;
    grad_grid<D>(total, &g0, &g1);
    uint4 *wt = S::WT_LDS ? nullptr : static_cast<uint4 *>(workspace);  // 16-byte aligned: the workspace is, and the image comes first
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + (S::WT_LDS ? 0 : (size_t)S::W_U4 * 16));
    if (!S::WT_LDS) hipLaunchKernelGGL(transposed_image_kernel, dim3(32), dim3(256), 0, s, wt, wmlp, D);
    const unsigned grid = g0 + (S::ROLES - 1) * g1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * S::WAVES), S::LDS_BYTES, s, total, n1, n2, (int)g0, (int)g1, xyz1, xyz2, points1, points2, idx, idx2, wpos,
                       bpos, wmlp, bmlp, wt, gout, d_xyz1, d_dir, d_points1, d_rows, partial);
    hipLaunchKernelGGL(cross_grad_reduce_kernel, dim3((S::G_FLOATS + 255) / 256), dim3(256), 0, s, partial, (int)grid, S::G_FLOATS, d_weights);
    return mcp_launch_status();
}

}  // namespace

MCP_EXPORT int mcp_cross_grad_floats(int d) { return d == 64 ? GradShape<64>::G_FLOATS : d == 128 ? GradShape<128>::G_FLOATS : 0; }

MCP_EXPORT size_t mcp_cross_grad_workspace_bytes(int b, int n1, int d) {
    if (b <= 0 || n1 <= 0) return 0;
    return d == 64 ? workspace_bytes<64>((long long)b * n1) : d == 128 ? workspace_bytes<128>((long long)b * n1) : 0;
}

MCP_EXPORT int mcp_cross_grad(int b, int n1, int n2, int d, int k, const float *xyz1, const float *xyz2, const float *points1, const float *points2,
                              const int *idx, const int *idx2, const float *wpos, const float *bpos, const float *wmlp, const float *bmlp,
                              const float *grad_out, float *grad_xyz1, float *grad_dir, float *grad_points1, float *grad_rows, float *grad_weights,
                              void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n1 > 0 && n2 > 0 && xyz1 && xyz2 && points1 && points2 && idx && wpos && bpos && wmlp && bmlp && grad_out && grad_xyz1 &&
                   grad_dir && grad_points1 && grad_rows && grad_weights && workspace);
    if (k != KNB || (d != 64 && d != 128)) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)points1) | ((uintptr_t)points2) | ((uintptr_t)grad_out) | ((uintptr_t)grad_rows) | ((uintptr_t)workspace)) & 15) return MCP_ERR_BAD_ARG;
    if (workspace_bytes < mcp_cross_grad_workspace_bytes(b, n1, d)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n1;
    mcp_prof_begin(MCP_KERNEL_CROSS, s);
    const int rc = d == 64 ? launch_cross_grad<64>(total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, wpos, bpos, wmlp, bmlp, grad_out, grad_xyz1,
                                                   grad_dir, grad_points1, grad_rows, grad_weights, workspace, s)
                           : launch_cross_grad<128>(total, n1, n2, xyz1, xyz2, points1, points2, idx, idx2, wpos, bpos, wmlp, bmlp, grad_out, grad_xyz1,
                                                    grad_dir, grad_points1, grad_rows, grad_weights, workspace, s);
    mcp_prof_end(MCP_KERNEL_CROSS, s);
    return rc;
}
