// fusion_grad.hip -- backward of the fused attentive fusion (mcp_fusion; MultiFrameEstimatier.knn_group + fusion,
// mocopci.py:798-819) for gfx950.  The reference differentiates the layer with autograd over the materialised
// (B,128,N,64) activations; here one wave owns one point at a time, re-evaluates the layer in the forward's MFMA layout and
// back-propagates inside the kernel:
//
//   phase 1  the forward again (both 32-neighbour halves): score s_j = max_c h3_j[c] with its arg-max channel c*_j, softmax a_j,
//            then ds_j = a_j (g.nb_j - sum_k a_k g.nb_k) and dz3_j = [s_j > 0] ds_j -- the only non-zero of dL/dz3_j sits at c*_j;
//   phase 2  per half: layers 1 and 2 again (h1, h2 stay in registers), and
//              dh2_j = dz3_j W3[c*_j,:]           a row gather (W3 rows from L2), no MFMA
//              dh1   = W2^T dz2                   split-bf16 MFMA, the gradient tile chained as B operand like the forward's activations
//              dx0_j = W1^T dz1_j, dr_j, dnb_j    VALU; d_nb (B,N,64,3) leaves for the caller's deterministic segmented scatter
//            and the weight gradients, whose contraction runs over the NEIGHBOUR axis (the MFMA column of every tensor above), so
//            each operand goes through a per-wave LDS tile once, written in accumulator layout and read back with 8 consecutive
//            neighbours per lane:
//              dW3 += OneHot(c*) . (dz3 h2)^T     the one-hot operand is exact in bf16: 3 MFMAs per product instead of 6
//              dW2 += dz2 . h1^T                  6 MFMAs per product (both operands split three ways)
//              dW1, db1, db2, db3                 in-lane sums over the transposed operands (lane = channel)
//   The weight gradients accumulate in registers over all points of a wave (one wave per SIMD: 192 accumulator registers), points
//   are dealt to waves statically, the four waves of a workgroup are added in wave order through LDS, and a second kernel adds the
//   workgroups' partial vectors in workgroup order: every sum has a fixed order, so the gradients repeat bit for bit.
//   Per point 672 bf16 MFMAs (forward: 288) + 16 f32 ones.
#include "common.h"
#include "mfma_grad.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C1 = 64, C2 = 64, C3 = 128, NB = 64;
constexpr int WAVES = 4;
constexpr int TS = MCP_TS;

// layout of the weight-gradient vector (floats): dW1 (64,4) | db1 | dW2 (64,64) | db2 | dW3 (128,64) | db3
constexpr int G_W1 = 0, G_B1 = G_W1 + C1 * 4, G_W2 = G_B1 + C1, G_B2 = G_W2 + C2 * C1, G_W3 = G_B2 + C2, G_B3 = G_W3 + C3 * C2,
              G_FLOATS = G_B3 + C3;  // 12800

// LDS image.  floats: W1 MFMA image [t][s][lane] | b1 [t][h][r] | b2 | b3 | W1 rows [t][h][r][4]
constexpr int L_W1 = 0, L_B1 = 256, L_B2 = L_B1 + 64, L_B3 = L_B2 + 64, L_W1R = L_B3 + 128, L_F32 = L_W1R + 256;
// then uint4: W2 image (forward) | W3 image (forward) | W2^T image (dh1)
constexpr int L_W2_U4 = 2 * 4 * 3 * 64, L_W3_U4 = 4 * 4 * 3 * 64;
// then per wave (floats): tile [64][TS] | x0 [32][4] | dz3 [32] | c* [32] (int)
constexpr int S_T = 0, S_X0 = 64 * TS, S_DZ = S_X0 + 128, S_CS = S_DZ + 32, S_FLOATS = S_CS + 32;
constexpr size_t LDS_BYTES = (size_t)L_F32 * 4 + (size_t)(2 * L_W2_U4 + L_W3_U4) * 16 + (size_t)WAVES * S_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((size_t)G_FLOATS * 4 <= (size_t)L_F32 * 4 + (size_t)(2 * L_W2_U4 + L_W3_U4) * 16, "the reduction buffer overlays the weight image");

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(64 * WAVES, 1) void fusion_grad_kernel(long long total, int n, const float *__restrict__ p1, const float *__restrict__ p2,
                                                                 const int *__restrict__ idx, const int *__restrict__ idx2,
                                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                                 const float *__restrict__ w3, const float *__restrict__ b3,
                                                                 const float *__restrict__ gout, float *__restrict__ d_p1,
                                                                 float *__restrict__ d_nb, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *w2s = reinterpret_cast<uint4 *>(lds + L_F32);
    uint4 *w3s = w2s + L_W2_U4;
    uint4 *w2ts = w3s + L_W3_U4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *scr = reinterpret_cast<float *>(w2ts + L_W2_U4) + wave * S_FLOATS;
    float *tb = scr + S_T;
    float4 *x0b = reinterpret_cast<float4 *>(scr + S_X0);
    float *dzb = scr + S_DZ;
    int *csb = reinterpret_cast<int *>(scr + S_CS);

    for (int e = tid; e < 256; e += 64 * WAVES) {  // w1 (fp32, K = 4): [t][s][lane] = W1[32t + (lane&31)][2s + (lane>>5)]
        const int l = e & 63, s = (e >> 6) & 1, t = e >> 7;
        lds[L_W1 + e] = w1[(32 * t + (l & 31)) * 4 + 2 * s + (l >> 5)];
        const int k = e & 3, r = (e >> 2) & 15, hh = (e >> 6) & 1, tt = e >> 7;  // W1 rows in accumulator order, for dx0 = W1^T dz1
        lds[L_W1R + e] = w1[(32 * tt + chan_of(r, hh)) * 4 + k];
    }
    mcp_split_weights(w2s, w2, C1, 2, tid, 64 * WAVES);
    mcp_split_weights(w3s, w3, C2, 4, tid, 64 * WAVES);
    mcp_split_weights_transposed(w2ts, w2, C1, C2, tid, 64 * WAVES);
    for (int e = tid; e < 64; e += 64 * WAVES) {  // biases: [t][h][r]
        const int r = e & 15, hh = (e >> 4) & 1, t = e >> 5;
        lds[L_B1 + e] = b1[32 * t + chan_of(r, hh)];
        lds[L_B2 + e] = b2[32 * t + chan_of(r, hh)];
    }
    for (int e = tid; e < 128; e += 64 * WAVES) {
        const int r = e & 15, hh = (e >> 4) & 1, t = e >> 5;
        lds[L_B3 + e] = b3[32 * t + chan_of(r, hh)];
    }
    __syncthreads();

    // weight-gradient accumulators of this wave.  MFMA tiles: row = 32 mt + chan_of(r, h), column = 32 nt + col.
    f32x16 dW2a[2][2], dW3a[4][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            dW2a[0][a][r] = 0.f; dW2a[1][a][r] = 0.f;
            dW3a[0][a][r] = 0.f; dW3a[1][a][r] = 0.f; dW3a[2][a][r] = 0.f; dW3a[3][a][r] = 0.f;
        }
    }
    // in-lane sums over transposed operands: lane = channel 32 mt + col, the two lane halves hold the two 8-neighbour groups
    float db1a[2] = {0.f, 0.f}, db2a[2] = {0.f, 0.f}, db3a[4] = {0.f, 0.f, 0.f, 0.f}, dW1a[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    const McpUnits units = mcp_units_by_xcd(total, WAVES);   // XCD x takes the x-th eighth of the points (common.h)
    for (long long p = units.first + wave; p < units.limit; p += units.stride) {
        const long long bb = mcp_div(p, n, mcp_fits32(total));
        const float cx = p1[p * 3 + 0], cy = p1[p * 3 + 1], cz = p1[p * 3 + 2];
        const float gx = gout[p * 3 + 0], gy = gout[p * 3 + 1], gz = gout[p * 3 + 2];
        // ---------------- phase 1: the forward, with the arg-max channel of every neighbour ----------------
        float score[2], nbx[2], nby[2], nbz[2];
        int cstar[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int id = idx2 ? (ct ? idx2 : idx)[p * 32 + col] : idx[p * NB + 32 * ct + col];
            const float *q = p2 + ((long long)bb * n + id) * 3;
            const float x = q[0], y = q[1], z = q[2];
            nbx[ct] = x; nby[ct] = y; nbz[ct] = z;
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            McpSplit3 x1[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[L_B1 + (t * 2 + h) * 16 + r];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                x1[2 * t + 0] = mcp_split_kstep(acc, 0);
                x1[2 * t + 1] = mcp_split_kstep(acc, 1);
            }
            McpSplit3 x2[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[L_B2 + (t * 2 + h) * 16 + r];
                acc = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                x2[2 * t + 0] = mcp_split_kstep(acc, 0);
                x2[2 * t + 1] = mcp_split_kstep(acc, 1);
            }
            float m = 0.f;
            int mr = 0;  // 16 t + r of the running maximum
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = lds[L_B3 + (t * 2 + h) * 16 + r];
                acc = mcp_tile_split<4>(w3s + (size_t)t * 4 * 3 * 64 + lane, x2, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool up = acc[r] > m;
                    m = up ? acc[r] : m;
                    mr = up ? 16 * t + r : mr;
                }
            }
            int mc = 32 * (mr >> 4) + chan_of(mr & 15, h);
            const float om = __shfl_xor(m, 32);
            const int oc = __shfl_xor(mc, 32);
            const bool other = om > m || (om == m && oc < mc);
            score[ct] = other ? om : m;
            cstar[ct] = other ? oc : mc;
        }
        // softmax over the 64 neighbours (every neighbour sits in both lane halves: wave sums count it twice)
        const float mx = wave_max(fmaxf(score[0], score[1]));
        const float e0 = expf(score[0] - mx), e1 = expf(score[1] - mx);
        const float den = 0.5f * wave_sum(e0 + e1);
        const float a0 = e0 / den, a1 = e1 / den;
        const float da0 = (gx * nbx[0] + gy * nby[0]) + gz * nbz[0], da1 = (gx * nbx[1] + gy * nby[1]) + gz * nbz[1];
        const float sdot = 0.5f * wave_sum(a0 * da0 + a1 * da1);
        const float dz3_0 = score[0] > 0.f ? a0 * (da0 - sdot) : 0.f, dz3_1 = score[1] > 0.f ? a1 * (da1 - sdot) : 0.f;

        // ---------------- phase 2: per half, layers 1-2 again and the backward chain ----------------
        float dcx = 0.f, dcy = 0.f, dcz = 0.f;
#pragma unroll 1
        for (int ct = 0; ct < 2; ++ct) {
            const float x = ct ? nbx[1] : nbx[0], y = ct ? nby[1] : nby[0], z = ct ? nbz[1] : nbz[0];
            const float dz3 = ct ? dz3_1 : dz3_0, aw = ct ? a1 : a0;
            const int cst = ct ? cstar[1] : cstar[0];
            const float rx = x - cx, ry = y - cy, rz = z - cz;
            const float dist = sqrtf((rx * rx + ry * ry) + rz * rz);
            const float in0 = h ? ry : rx, in1 = h ? dist : rz;
            f32x16 h1[2], h2[2];
            {
                McpSplit3 x1[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = lds[L_B1 + (t * 2 + h) * 16 + r];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 0) * 64 + lane], in0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W1 + (t * 2 + 1) * 64 + lane], in1, acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                    h1[t] = acc;
                    x1[2 * t + 0] = mcp_split_kstep(acc, 0);
                    x1[2 * t + 1] = mcp_split_kstep(acc, 1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = lds[L_B2 + (t * 2 + h) * 16 + r];
                    acc = mcp_tile_split<4>(w2s + (size_t)t * 4 * 3 * 64 + lane, x1, acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
                    h2[t] = acc;
                }
            }
            // ---- dW3 += OneHot(c*) . (dz3 h2)^T, db3 ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[(32 * t + chan_of(r, h)) * TS + col] = dz3 * h2[t][r];
            if (h == 0) {
                csb[col] = cst;
                dzb[col] = dz3;
                x0b[col] = make_float4(rx, ry, rz, dist);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int j0 = 16 * ks + 8 * h;  // this lane's 8 neighbours of the k-step
                const int4 ca = reinterpret_cast<const int4 *>(csb + j0)[0], cb = reinterpret_cast<const int4 *>(csb + j0)[1];
                const int cs8[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                float dz8[8];
                mcp_read8(dzb + j0, dz8);
                uint4 oh[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c = 32 * mt + col;
                    uint32_t d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[q] = (cs8[2 * q] == c ? 0x00003F80u : 0u) | (cs8[2 * q + 1] == c ? 0x3F800000u : 0u);
                    oh[mt] = make_uint4(d[0], d[1], d[2], d[3]);
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s += cs8[i] == c ? dz8[i] : 0.f;
                    db3a[mt] += s;
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[8];
                    mcp_read8(tb + (32 * nt + col) * TS + j0, v);
                    const McpSplit3 bs = mcp_split8(v);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        dW3a[mt][nt] = mcp_mfma_bf16(oh[mt], bs.p3, dW3a[mt][nt]);
                        dW3a[mt][nt] = mcp_mfma_bf16(oh[mt], bs.p2, dW3a[mt][nt]);
                        dW3a[mt][nt] = mcp_mfma_bf16(oh[mt], bs.p1, dW3a[mt][nt]);
                    }
                }
            }
            // ---- dz2 = relu'(h2) . dz3 W3[c*,:] ----
            f32x16 dz2[2];
            {
                const float4 *w3row = reinterpret_cast<const float4 *>(w3 + (size_t)cst * C2);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w = w3row[8 * t + 2 * q + h];  // channels 32 t + 8 q + 4 h + (0..3) = chan_of(4 q + i, h)
                        dz2[t][4 * q + 0] = h2[t][4 * q + 0] > 0.f ? dz3 * w.x : 0.f;
                        dz2[t][4 * q + 1] = h2[t][4 * q + 1] > 0.f ? dz3 * w.y : 0.f;
                        dz2[t][4 * q + 2] = h2[t][4 * q + 2] > 0.f ? dz3 * w.z : 0.f;
                        dz2[t][4 * q + 3] = h2[t][4 * q + 3] > 0.f ? dz3 * w.w : 0.f;
                    }
            }
            // ---- dz1 = relu'(h1) . W2^T dz2 ----
            f32x16 dz1[2];
            {
                McpSplit3 xs[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xs[2 * t + 0] = mcp_split_kstep(dz2[t], 0);
                    xs[2 * t + 1] = mcp_split_kstep(dz2[t], 1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = mcp_tile_split<4>(w2ts + (size_t)t * 4 * 3 * 64 + lane, xs, acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz1[t][r] = h1[t][r] > 0.f ? acc[r] : 0.f;
                }
            }
            // ---- dW2 += dz2 . h1^T, db2 ----
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(tb, dz2, col, h);
            __builtin_amdgcn_wave_barrier();
            McpSplit3 as[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
                    db2a[mt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    as[mt][ks] = mcp_split8(v);
                }
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(tb, h1, col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * nt + col) * TS + 16 * ks + 8 * h, v);
                    const McpSplit3 bs = mcp_split8(v);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) dW2a[mt][nt] = mcp_mfma_split6(as[mt][ks], bs, dW2a[mt][nt]);
                }
            // ---- dW1 += dz1 . x0^T, db1 ----
            __builtin_amdgcn_wave_barrier();
            mcp_write_tiles(tb, dz1, col, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
                    mcp_read8(tb + (32 * mt + col) * TS + 16 * ks + 8 * h, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 xj = x0b[16 * ks + 8 * h + i];
                        dW1a[mt][0] = __builtin_fmaf(v[i], xj.x, dW1a[mt][0]);
                        dW1a[mt][1] = __builtin_fmaf(v[i], xj.y, dW1a[mt][1]);
                        dW1a[mt][2] = __builtin_fmaf(v[i], xj.z, dW1a[mt][2]);
                        dW1a[mt][3] = __builtin_fmaf(v[i], xj.w, dW1a[mt][3]);
                        db1a[mt] += v[i];
                    }
                }
            // ---- dx0 = W1^T dz1, back through [r, |r|] to the neighbour and the centre ----
            float dx0 = 0.f, dx1 = 0.f, dx2 = 0.f, dx3 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 w = reinterpret_cast<const float4 *>(lds + L_W1R)[(t * 2 + h) * 16 + r];
                    dx0 = __builtin_fmaf(w.x, dz1[t][r], dx0);
                    dx1 = __builtin_fmaf(w.y, dz1[t][r], dx1);
                    dx2 = __builtin_fmaf(w.z, dz1[t][r], dx2);
                    dx3 = __builtin_fmaf(w.w, dz1[t][r], dx3);
                }
            dx0 += __shfl_xor(dx0, 32);
            dx1 += __shfl_xor(dx1, 32);
            dx2 += __shfl_xor(dx2, 32);
            dx3 += __shfl_xor(dx3, 32);
            const float sc = dist > 0.f ? dx3 / dist : 0.f;  // d|r|/dr = r/|r|, 0 at r = 0 (torch.norm's subgradient)
            const float drx = __builtin_fmaf(sc, rx, dx0), dry = __builtin_fmaf(sc, ry, dx1), drz = __builtin_fmaf(sc, rz, dx2);
            if (h == 0) {
                float *o = d_nb + (p * NB + 32 * ct + col) * 3;
                o[0] = __builtin_fmaf(aw, gx, drx);
                o[1] = __builtin_fmaf(aw, gy, dry);
                o[2] = __builtin_fmaf(aw, gz, drz);
            }
            dcx += drx; dcy += dry; dcz += drz;
        }
        const float sx = wave_sum(dcx), sy = wave_sum(dcy), sz = wave_sum(dcz);  // both lane halves hold every neighbour
        if (lane == 0) {
            d_p1[p * 3 + 0] = -0.5f * sx;
            d_p1[p * 3 + 1] = -0.5f * sy;
            d_p1[p * 3 + 2] = -0.5f * sz;
        }
    }

    // ---- the workgroup's partial vector: waves added in wave order through LDS (over the weight image, no longer needed) ----
    __syncthreads();
    float *red = lds;
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
            const bool first = w == 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *o = red + G_W2 + (32 * mt + chan_of(r, h)) * C1 + 32 * nt + col;
                        *o = first ? dW2a[mt][nt][r] : *o + dW2a[mt][nt][r];
                    }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *o = red + G_W3 + (32 * mt + chan_of(r, h)) * C2 + 32 * nt + col;
                        *o = first ? dW3a[mt][nt][r] : *o + dW3a[mt][nt][r];
                    }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float v3 = db3a[mt] + __shfl_xor(db3a[mt], 32);
                if (h == 0) red[G_B3 + 32 * mt + col] = first ? v3 : red[G_B3 + 32 * mt + col] + v3;
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float v1 = db1a[mt] + __shfl_xor(db1a[mt], 32), v2 = db2a[mt] + __shfl_xor(db2a[mt], 32);
                if (h == 0) {
                    red[G_B1 + 32 * mt + col] = first ? v1 : red[G_B1 + 32 * mt + col] + v1;
                    red[G_B2 + 32 * mt + col] = first ? v2 : red[G_B2 + 32 * mt + col] + v2;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float vw = dW1a[mt][k] + __shfl_xor(dW1a[mt][k], 32);
                    if (h == 0) red[G_W1 + (32 * mt + col) * 4 + k] = first ? vw : red[G_W1 + (32 * mt + col) * 4 + k] + vw;
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < G_FLOATS; e += 64 * WAVES) partial[(size_t)blockIdx.x * G_FLOATS + e] = red[e];
}

// out[e] = sum over the workgroups' partial vectors, in workgroup order
__global__ __launch_bounds__(256) void fusion_grad_reduce_kernel(const float *__restrict__ partial, int parts, float *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G_FLOATS) return;
    float s = 0.f;
    for (int g = 0; g < parts; ++g) s += partial[(size_t)g * G_FLOATS + e];
    out[e] = s;
}

unsigned grad_grid(long long total) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long want = (total + WAVES - 1) / WAVES;
    return (unsigned)(want < cus ? want : cus);  // one resident workgroup per CU (LDS), points dealt out statically
}

}  // namespace

MCP_EXPORT int mcp_fusion_grad_floats(void) { return G_FLOATS; }

MCP_EXPORT size_t mcp_fusion_grad_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (size_t)grad_grid((long long)b * n) * G_FLOATS * sizeof(float);
}

MCP_EXPORT int mcp_fusion_grad(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1,
                               const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, const float *grad_out,
                               float *grad_p1, float *grad_nb, float *grad_weights, void *workspace, size_t workspace_bytes,
                               mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && p1 && p2 && idx && w1 && b1 && w2 && b2 && w3 && b3 && grad_out && grad_p1 && grad_nb && grad_weights && workspace);
    MCP_CHECK_ARGS(((uintptr_t)w3 & 15) == 0);  // W3 rows are read as float4
    if (nb != NB) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n;
    const unsigned grid = grad_grid(total);
    if (workspace_bytes < (size_t)grid * G_FLOATS * sizeof(float)) return MCP_ERR_BAD_ARG;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fusion_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.done();
    }
    mcp_prof_begin(MCP_KERNEL_FUSION, s);
    hipLaunchKernelGGL(fusion_grad_kernel, dim3(grid), dim3(64 * WAVES), LDS_BYTES, s, total, n, p1, p2, idx, idx2, w1, b1, w2, b2, w3, b3, grad_out,
                       grad_p1, grad_nb, static_cast<float *>(workspace));
    hipLaunchKernelGGL(fusion_grad_reduce_kernel, dim3((G_FLOATS + 255) / 256), dim3(256), 0, s, static_cast<const float *>(workspace), (int)grid,
                       grad_weights);
    mcp_prof_end(MCP_KERNEL_FUSION, s);
    return mcp_launch_status();
}
