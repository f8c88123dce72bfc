// ptblock_grad.hip -- backward of the fused Point-Transformer vector attention (mcp_ptblock_attention; TransformerBlock.forward,
// models/pointT_layer2.py:58-77, d_model = 64, k = 16) for gfx950.  The reference differentiates five materialised (B,N,16,64)
// tensors with autograd.  Here a wave owns TWO points (their 2 x 16 neighbours on the MFMA column, as in the forward), re-evaluates
//     d1 = relu(Wd1 d + bd1), delta = Wd2 d1 + bd2, g = (q_i - k_j) + delta, a1 = relu(Wg1 g + bg1), attn = Wg2 a1 + bg2
// in the forward's layout and arithmetic and back-propagates inside the kernel:
//   * per-channel softmax over a point's 16 neighbours (one DPP row) backwards: dattn = w (dw - sum_j w dw) / 8, dval = w G;
//   * da1 = Wg2^T dattn, dg = Wg1^T dz1, dd1 = Wd2^T ddelta on the split-bf16 MFMA path (transposed weight images read through
//     L2), the gradient tile chained as B operand like the forward's activations;
//   * the three 64 x 64 weight gradients contract over the NEIGHBOUR axis: both operands of each pass once through a per-wave LDS
//     tile (written in accumulator layout, read back with 8 consecutive neighbours per lane) -- 192 accumulator registers over all
//     of a wave's points; biases, dWd1, dL/dq as in-lane sums with lane = channel;
//   * per-neighbour gradients leave as rows for the caller's deterministic segmented scatter: dk_rows = -dg, dv_rows = dval
//     (B,N,16,64), dxyz_rows = -dd (B,N,16,3); dL/dq and the centre's share of dL/dxyz are written directly.
// Waves are added in wave order through LDS, workgroups in workgroup order by a second kernel: bit-reproducible.
#include "common.h"
#include "mfma_grad.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 64, KNB = 16, WAVES = 4;
constexpr int TS = MCP_TS;
// weight-gradient vector (floats): dWd1 (64,3) | dbd1 | dWd2 (64,64) | dbd2 | dWg1 (64,64) | dbg1 | dWg2 (64,64) | dbg2
constexpr int G_WD1 = 0, G_BD1 = 192, G_WD2 = 256, G_BD2 = G_WD2 + 4096, G_WG1 = G_BD2 + 64, G_BG1 = G_WG1 + 4096, G_WG2 = G_BG1 + 64,
              G_BG2 = G_WG2 + 4096, G_FLOATS = G_BG2 + 64;  // 12736
// LDS, floats: Wd1 MFMA image [t][s][lane] | bd2, bg1, bg2 [t][h][r] | Wd1 rows [t][h][r][4]
constexpr int L_D1 = 0, L_BD2 = 256, L_BG1 = L_BD2 + 64, L_BG2 = L_BG1 + 64, L_WD1R = L_BG2 + 64, L_F32 = L_WD1R + 256;
constexpr int W_U4 = 2 * 4 * 3 * 64;  // uint4 per 64 x 64 split image
// per wave (floats): two tiles [64][TS] | directions [32][4]
constexpr int S_TA = 0, S_TB = 64 * TS, S_DIR = 2 * 64 * TS, S_FLOATS = S_DIR + 128;
constexpr size_t LDS_BYTES = (size_t)L_F32 * 4 + (size_t)3 * W_U4 * 16 + (size_t)WAVES * S_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((size_t)G_FLOATS * 4 <= (size_t)L_F32 * 4 + (size_t)3 * W_U4 * 16, "the reduction buffer overlays the weight images");

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __uint_as_float(mcp_dpp<CTRL>(__float_as_uint(v)));
}
__device__ __forceinline__ float row_max(float v) {  // as the forward (ptblock.hip): over a 16-lane DPP row = one point's neighbours
    v = fmaxf(v, dppf<0xB1>(v));
    v = fmaxf(v, dppf<0x4E>(v));
    v = fmaxf(v, dppf<0x141>(v));
    return fmaxf(v, dppf<0x140>(v));
}
__device__ __forceinline__ float row_sum(float v) {
    v += dppf<0xB1>(v);
    v += dppf<0x4E>(v);
    v += dppf<0x141>(v);
    return v + dppf<0x140>(v);
}

// the three transposed images (Wd2^T | Wg1^T | Wg2^T) into the workspace
__global__ __launch_bounds__(256) void ptblock_transposed_images_kernel(uint4 *dst, const float *__restrict__ wd2, const float *__restrict__ wg1,
                                                                        const float *__restrict__ wg2) {
    const float *w = blockIdx.y == 0 ? wd2 : blockIdx.y == 1 ? wg1 : wg2;
    mcp_split_weights_transposed(dst + (size_t)blockIdx.y * W_U4, w, C, C, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// y = W x + b (64 -> 64) on accumulator-layout input, as the forward's layer64
__device__ __forceinline__ void layer64(const uint4 *ws, const float *bias, int lane, int h, const f32x16 (&x)[2], f32x16 (&y)[2]) {
    McpSplit3 xs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xs[s] = mcp_split_kstep(x[s >> 1], s & 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias ? bias[(t * 2 + h) * 16 + r] : 0.f;
        y[t] = mcp_tile_split<4>(ws + (size_t)t * 4 * 3 * 64 + lane, xs, acc);
    }
}
// dW += dy . x^T over the wave's 32 neighbour columns: x^T is in tile `tx` already, dy goes through tile `ty`; dbias = sum of dy
__device__ __forceinline__ void weight_grad(float *ty, const float *tx, const f32x16 *dy, int col, int h, f32x16 (&dW)[2][2], float (&db)[2]) {
    McpSplit3 as[2][2];
    __builtin_amdgcn_wave_barrier();
    mcp_write_tiles(ty, dy, col, h);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
            mcp_read8(ty + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
            db[mt] += mcp_sum8(v);
            as[mt][ks] = mcp_split8(v);
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
            mcp_read8(tx + (32 * nt + col) * TS + 16 * ks + 8 * h, v);
            const McpSplit3 bs = mcp_split8(v);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) dW[mt][nt] = mcp_mfma_split6(as[mt][ks], bs, dW[mt][nt]);
        }
}

__global__ __launch_bounds__(64 * WAVES, 1) void ptblock_grad_kernel(long long total, int n, int rs, const float *__restrict__ xyz, const float *__restrict__ q,
                                                                  const float *__restrict__ kf, const float *__restrict__ vf, const int *__restrict__ idx,
                                                                  const float *__restrict__ wd1, const float *__restrict__ bd1,
                                                                  const float *__restrict__ wd2, const float *__restrict__ bd2,
                                                                  const float *__restrict__ wg1, const float *__restrict__ bg1,
                                                                  const float *__restrict__ wg2, const float *__restrict__ bg2,
                                                                  const uint4 *__restrict__ wt, float scale_log2e, float inv_sqrt_c,
                                                                  const float *__restrict__ gout, float *__restrict__ d_xyz_c, float *__restrict__ d_xyz_rows,
                                                                  float *__restrict__ d_q, float *__restrict__ d_k_rows, float *__restrict__ d_v_rows,
                                                                  float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *wd2s = reinterpret_cast<uint4 *>(lds + L_F32);
    uint4 *wg1s = wd2s + W_U4, *wg2s = wg1s + W_U4;
    const uint4 *wd2t = wt, *wg1t = wt + W_U4, *wg2t = wt + 2 * W_U4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31, sel = col >> 4, j = col & 15;
    float *scr = reinterpret_cast<float *>(wg2s + W_U4) + wave * S_FLOATS;
    float *ta = scr + S_TA, *tb = scr + S_TB;
    float4 *dirb = reinterpret_cast<float4 *>(scr + S_DIR);

    for (int e = tid; e < 256; e += 64 * WAVES) {  // [t][s][lane]: columns (dx,dy | dz,1) of [wd1 | bd1], as ptblock_pack_kernel
        const int l = e & 63, s = (e >> 6) & 1, t = e >> 7;
        const int row = 32 * t + (l & 31), c = 2 * s + (l >> 5);
        lds[L_D1 + e] = c < 3 ? wd1[row * 3 + c] : bd1[row];
        const int k = e & 3, r = (e >> 2) & 15, hh = (e >> 6) & 1;  // Wd1 rows in accumulator order, for dd = Wd1^T dzd
        lds[L_WD1R + e] = k < 3 ? wd1[(32 * t + chan_of(r, hh)) * 3 + k] : 0.f;
    }
    for (int e = tid; e < 64; e += 64 * WAVES) {
        const int r = e & 15, hh = (e >> 4) & 1, t = e >> 5, c = 32 * t + chan_of(r, hh);
        lds[L_BD2 + e] = bd2[c];
        lds[L_BG1 + e] = bg1[c];
        lds[L_BG2 + e] = bg2[c];
    }
    mcp_split_weights(wd2s, wd2, C, 2, tid, 64 * WAVES);
    mcp_split_weights(wg1s, wg1, C, 2, tid, 64 * WAVES);
    mcp_split_weights(wg2s, wg2, C, 2, tid, 64 * WAVES);
    __syncthreads();

    f32x16 dWd2a[2][2], dWg1a[2][2], dWg2a[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dWd2a[a][b][r] = 0.f; dWg1a[a][b][r] = 0.f; dWg2a[a][b][r] = 0.f; }
    float dbd2[2] = {0.f, 0.f}, dbg1[2] = {0.f, 0.f}, dbg2[2] = {0.f, 0.f}, dbd1[2] = {0.f, 0.f};  // lane = channel 32 mt + col
    float dWd1a[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};

    const long long pairs = (total + 1) / 2;
    const McpUnits units = mcp_units_by_xcd(pairs, WAVES);   // XCD x takes the x-th eighth of the point pairs (common.h)
    for (long long pp = units.first + wave; pp < units.limit; pp += units.stride) {
        long long p = 2 * pp + sel;
        const bool live = p < total;
        if (!live) p = total - 1;  // odd tail: the second half works on the last point with a zero upstream gradient
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const int id = idx[p * KNB + j];
        const float *cj = xyz + ((long long)bb * n + id) * 3;
        const float dx = xyz[p * 3 + 0] - cj[0], dy = xyz[p * 3 + 1] - cj[1], dz = xyz[p * 3 + 2] - cj[2];  // xyz_i - xyz_j
        const float in0 = h ? dy : dx, in1 = h ? 1.0f : dz;
        const float4 *qrow = reinterpret_cast<const float4 *>(q + p * rs);
        const float4 *krow = reinterpret_cast<const float4 *>(kf + ((long long)bb * n + id) * rs);
        const float4 *vrow = reinterpret_cast<const float4 *>(vf + ((long long)bb * n + id) * rs);
        const float4 *grow = reinterpret_cast<const float4 *>(gout + p * C);
        const long long nrow = p * KNB + j;  // this lane's (point, neighbour) row of the per-neighbour outputs
        // ---------------- the forward again ----------------
        auto first_layer = [&](f32x16 (&d1)[2]) {  // d1 = relu(Wd1 d + bd1): evaluated twice per point pair rather than kept (registers)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_D1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_D1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                d1[t] = acc;
            }
        };
        f32x16 val[2], attn[2];
        uint32_t live_a1 = 0u;  // bit 16 t + r: a1 > 0
        {
            f32x16 d1[2], delta[2], g[2], a1[2];
            first_layer(d1);
            layer64(wd2s, lds + L_BD2, lane, h, d1, delta);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c4 = (32 * t + 8 * gq + 4 * h) >> 2;
                    const float4 qq = qrow[c4], kk = krow[c4], vv = vrow[c4];
                    g[t][4 * gq + 0] = (qq.x - kk.x) + delta[t][4 * gq + 0];
                    g[t][4 * gq + 1] = (qq.y - kk.y) + delta[t][4 * gq + 1];
                    g[t][4 * gq + 2] = (qq.z - kk.z) + delta[t][4 * gq + 2];
                    g[t][4 * gq + 3] = (qq.w - kk.w) + delta[t][4 * gq + 3];
                    val[t][4 * gq + 0] = vv.x + delta[t][4 * gq + 0];
                    val[t][4 * gq + 1] = vv.y + delta[t][4 * gq + 1];
                    val[t][4 * gq + 2] = vv.z + delta[t][4 * gq + 2];
                    val[t][4 * gq + 3] = vv.w + delta[t][4 * gq + 3];
                }
            layer64(wg1s, lds + L_BG1, lane, h, g, a1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    live_a1 |= a1[t][r] > 0.f ? 1u << (16 * t + r) : 0u;
                    a1[t][r] = fmaxf(a1[t][r], 0.f);
                }
            layer64(wg2s, lds + L_BG2, lane, h, a1, attn);
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(ta, a1, col, h);  // a1^T: the B operand of dWg2
            mcp_write_tiles(tb, g, col, h);   // g^T: the B operand of dWg1
            if (h == 0) dirb[col] = make_float4(dx, dy, dz, 0.f);
        }
        // ---------------- softmax over the 16 neighbours, backwards ----------------
        // dval = w G is dL/dv of the gathered row: written now, read back (L2) where fc_delta's input gradient is formed
        f32x16 dattn[2];
        float4 *ov = reinterpret_cast<float4 *>(d_v_rows + nrow * C);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 gv4 = grow[(32 * t + 8 * gq + 4 * h) >> 2];
                const float gvs[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
                float dv4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * gq + u;
                    const float gc = live ? gvs[u] : 0.f;
                    const float a = attn[t][r] * scale_log2e;
                    const float e = __builtin_amdgcn_exp2f(a - row_max(a));
                    const float w = e / row_sum(e);
                    const float dw = gc * val[t][r];
                    const float dot = row_sum(w * dw);
                    dattn[t][r] = inv_sqrt_c * (w * (dw - dot));
                    dv4[u] = w * gc;
                }
                if (live) ov[(32 * t + 8 * gq + 4 * h) >> 2] = make_float4(dv4[0], dv4[1], dv4[2], dv4[3]);
            }
        // ---------------- fc_gamma backwards ----------------
        f32x16 dz1[2];
        {
            // dWg2 += dattn . a1^T (a1^T in tile A; dattn through tile... B holds g^T, so dattn goes through a read-ahead of tile A)
            McpSplit3 bs[2][2];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(ta + (32 * nt + col) * TS + 16 * ks + 8 * h, v);
                    bs[nt][ks] = mcp_split8(v);
                }
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(ta, dattn, col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(ta + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
                    dbg2[mt] += mcp_sum8(v);
                    const McpSplit3 as = mcp_split8(v);
                    dWg2a[mt][0] = mcp_mfma_split6(as, bs[0][ks], dWg2a[mt][0]);
                    dWg2a[mt][1] = mcp_mfma_split6(as, bs[1][ks], dWg2a[mt][1]);
                }
            f32x16 da1[2];
            layer64(wg2t, nullptr, lane, h, dattn, da1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dz1[t][r] = (live_a1 >> (16 * t + r)) & 1u ? da1[t][r] : 0.f;
        }
        weight_grad(ta, tb, dz1, col, h, dWg1a, dbg1);  // dWg1 += dz1 . g^T (g^T in tile B, dz1 through tile A)
        f32x16 ddelta[2];
        {
            f32x16 dg[2];
            layer64(wg1t, nullptr, lane, h, dz1, dg);
            // dL/dq = sum_j dg_j: lane = channel over the transposed tile; a point's 16 neighbours are one k-step (two lane halves)
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(ta, dg, col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(ta + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
                    float s = mcp_sum8(v);
                    s += __shfl_xor(s, 32);
                    const long long pq = 2 * pp + ks;
                    if (h == 0 && pq < total) d_q[pq * C + 32 * mt + col] = s;
                }
            if (live) {
                float4 *ok = reinterpret_cast<float4 *>(d_k_rows + nrow * C);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        ok[(32 * t + 8 * gq + 4 * h) >> 2] = make_float4(-dg[t][4 * gq + 0], -dg[t][4 * gq + 1], -dg[t][4 * gq + 2], -dg[t][4 * gq + 3]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) dv = ov[(32 * t + 8 * gq + 4 * h) >> 2];
                    ddelta[t][4 * gq + 0] = dv.x + dg[t][4 * gq + 0];
                    ddelta[t][4 * gq + 1] = dv.y + dg[t][4 * gq + 1];
                    ddelta[t][4 * gq + 2] = dv.z + dg[t][4 * gq + 2];
                    ddelta[t][4 * gq + 3] = dv.w + dg[t][4 * gq + 3];
                }
        }
        // ---------------- fc_delta backwards ----------------
        f32x16 d1[2];
        first_layer(d1);
        __builtin_amdgcn_wave_barrier();
        mcp_write_tiles(tb, d1, col, h);  // d1^T: the B operand of dWd2
        weight_grad(ta, tb, ddelta, col, h, dWd2a, dbd2);
        f32x16 dzd[2];
        {
            f32x16 dd1[2];
            layer64(wd2t, nullptr, lane, h, ddelta, dd1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dzd[t][r] = d1[t][r] > 0.f ? dd1[t][r] : 0.f;
        }
        // dWd1 += dzd . d^T, dbd1 (lane = channel); dd = Wd1^T dzd
        __builtin_amdgcn_wave_barrier();
        mcp_write_tiles(ta, dzd, col, h);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
                mcp_read8(ta + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
                dbd1[mt] += mcp_sum8(v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 dj = dirb[16 * ks + 8 * h + i];
                    dWd1a[mt][0] = __builtin_fmaf(v[i], dj.x, dWd1a[mt][0]);
                    dWd1a[mt][1] = __builtin_fmaf(v[i], dj.y, dWd1a[mt][1]);
                    dWd1a[mt][2] = __builtin_fmaf(v[i], dj.z, dWd1a[mt][2]);
                }
            }
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 w = reinterpret_cast<const float4 *>(lds + L_WD1R)[(t * 2 + h) * 16 + r];
                ax = __builtin_fmaf(w.x, dzd[t][r], ax);
                ay = __builtin_fmaf(w.y, dzd[t][r], ay);
                az = __builtin_fmaf(w.z, dzd[t][r], az);
            }
        ax += __shfl_xor(ax, 32);
        ay += __shfl_xor(ay, 32);
        az += __shfl_xor(az, 32);
        if (live && h == 0) {  // d = xyz_i - xyz_j: the neighbour gets -dd
            float *o = d_xyz_rows + nrow * 3;
            o[0] = -ax; o[1] = -ay; o[2] = -az;
        }
        const float sx = row_sum(ax), sy = row_sum(ay), sz = row_sum(az);  // over the point's 16 neighbours
        if (live && h == 0 && j == 0) {
            d_xyz_c[p * 3 + 0] = sx;
            d_xyz_c[p * 3 + 1] = sy;
            d_xyz_c[p * 3 + 2] = sz;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- the workgroup's partial vector: waves added in wave order through LDS (over the weight images, no longer needed) ----
    __syncthreads();
    float *red = lds;
    for (int e = tid; e < G_FLOATS; e += 64 * WAVES) red[e] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = (32 * mt + chan_of(r, h)) * C + 32 * nt + col;
                        red[G_WD2 + o] += dWd2a[mt][nt][r];
                        red[G_WG1 + o] += dWg1a[mt][nt][r];
                        red[G_WG2 + o] += dWg2a[mt][nt][r];
                    }
                const float b0 = dbd1[mt] + __shfl_xor(dbd1[mt], 32), b1 = dbd2[mt] + __shfl_xor(dbd2[mt], 32);
                const float b2 = dbg1[mt] + __shfl_xor(dbg1[mt], 32), b3 = dbg2[mt] + __shfl_xor(dbg2[mt], 32);
                float vw[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) vw[k] = dWd1a[mt][k] + __shfl_xor(dWd1a[mt][k], 32);
                if (h == 0) {
                    const int c = 32 * mt + col;
                    red[G_BD1 + c] += b0; red[G_BD2 + c] += b1; red[G_BG1 + c] += b2; red[G_BG2 + c] += b3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) red[G_WD1 + c * 3 + k] += vw[k];
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < G_FLOATS; e += 64 * WAVES) partial[(size_t)blockIdx.x * G_FLOATS + e] = red[e];
}

__global__ __launch_bounds__(256) void ptblock_grad_reduce_kernel(const float *__restrict__ partial, int parts, float *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G_FLOATS) return;
    float s = 0.f;
    for (int g = 0; g < parts; ++g) s += partial[(size_t)g * G_FLOATS + e];
    out[e] = s;
}

unsigned grad_grid(long long total) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = ((total + 1) / 2 + WAVES - 1) / WAVES;
    return (unsigned)(want < cus ? want : cus);  // one resident workgroup per CU, point pairs dealt out statically
}

}  // namespace

MCP_EXPORT int mcp_ptblock_grad_floats(void) { return G_FLOATS; }

MCP_EXPORT size_t mcp_ptblock_grad_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (size_t)3 * W_U4 * 16 + (size_t)grad_grid((long long)b * n) * G_FLOATS * sizeof(float);
}

MCP_EXPORT int mcp_ptblock_grad(int b, int n, int c, int k, int qkv_stride, const float *xyz, const float *q, const float *kf, const float *vf, const int *idx,
                                const float *wd1, const float *bd1, const float *wd2, const float *bd2, const float *wg1, const float *bg1, const float *wg2,
                                const float *bg2, const float *grad_out, float *grad_xyz_c, float *grad_xyz_rows, float *grad_q, float *grad_k_rows,
                                float *grad_v_rows, float *grad_weights, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && q && kf && vf && idx && wd1 && bd1 && wd2 && bd2 && wg1 && bg1 && wg2 && bg2 && grad_out && grad_xyz_c &&
                   grad_xyz_rows && grad_q && grad_k_rows && grad_v_rows && grad_weights && workspace);
    if (c != C || k != KNB) return MCP_ERR_UNSUPPORTED;
    if (qkv_stride < C || (qkv_stride & 3)) return MCP_ERR_BAD_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)kf) | ((uintptr_t)vf) | ((uintptr_t)grad_out) | ((uintptr_t)grad_k_rows) | ((uintptr_t)grad_v_rows) | ((uintptr_t)workspace)) & 15)
        return MCP_ERR_BAD_ARG;
    if (workspace_bytes < mcp_ptblock_grad_workspace_bytes(b, n)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n;
    const unsigned grid = grad_grid(total);
    uint4 *wt = static_cast<uint4 *>(workspace);
    float *partial = reinterpret_cast<float *>(wt + 3 * W_U4);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ptblock_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    mcp_prof_begin(MCP_KERNEL_PTBLOCK, s);
    hipLaunchKernelGGL(ptblock_transposed_images_kernel, dim3(4, 3), dim3(256), 0, s, wt, wd2, wg1, wg2);
    hipLaunchKernelGGL(ptblock_grad_kernel, dim3(grid), dim3(64 * WAVES), LDS_BYTES, s, total, n, qkv_stride, xyz, q, kf, vf, idx, wd1, bd1, wd2, bd2, wg1, bg1,
                       wg2, bg2, wt, 1.44269504088896340736f / 8.0f, 0.125f, grad_out, grad_xyz_c, grad_xyz_rows, grad_q, grad_k_rows, grad_v_rows, partial);
    hipLaunchKernelGGL(ptblock_grad_reduce_kernel, dim3((G_FLOATS + 255) / 256), dim3(256), 0, s, partial, (int)grid, grad_weights);
    mcp_prof_end(MCP_KERNEL_PTBLOCK, s);
    return mcp_launch_status();
}
