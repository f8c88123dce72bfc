// linear_grad.hip -- weight and bias gradient of a tall per-point Linear for gfx950 (training).
//
// Reference: autograd over the Conv1d / Linear layers of the caller graph (mocopci.py:1111-1127, :438-468, :1021-1053) under
// train.py:162.  For y = act(x W^T + b) with rows >> n, k the backward needs
//     gz = gy * act'(z)                       (rows, n)       -- formed by the caller (one elementwise pass, shared with dx)
//     dx = gz W                               (rows, k)       -- the forward kernel again (mcp_linear on W^T)
//     dW = gz^T x                             (n, k)          -- THIS kernel
//     db = column sums of gz                  (n)             -- THIS kernel
// The library GEMM sees dW as a (n x rows) x (rows x k) product with a 64 x 32 result: one or two workgroups' worth of output tiles
// and a reduction axis of 196608 -- 188 us per layer at the pipeline's shapes for 75 MB of operands (round 5 trace), and db is a
// separate reduce launch.  Here the rows are the MFMA's contraction axis: a workgroup stages 32 rows of gz and x in LDS (coalesced
// float4 loads), its four waves share out the (n/32) x (k/32) output tiles and run 16 v_mfma_f32_32x32x2_f32 per tile and stage
// (f32 inputs: exact products, the kernel is bound by the operand stream, not the matrix pipe); the tiles stay in registers over all
// of the workgroup's stages.  Workgroup partials are written out and added in workgroup order by a second kernel: the result does
// not depend on how the hardware schedules anything (bit-reproducible), unlike a split-K GEMM with atomics.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WG_ROWS = 32, WAVES = 4, MAX_WGS = 256;

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// TPW: output tiles per wave (tiles wave, wave + 4, ...); the host picks the smallest of 1, 2, 4, 8, 16 that covers (n/32)(k/32)/4
template <int TPW>
__global__ __launch_bounds__(64 * WAVES) void linear_wgrad_kernel(long long rows, int n, int k, const float *__restrict__ gz, int gs,
                                                                const float *__restrict__ x, int xs, float *__restrict__ partial,
                                                                int stages, int stages_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    // n, k need not be multiples of 32 (the 3 -> 32 lift and the 32 -> 3 head of the caller graph): the staged tiles are padded with
    // zero columns to whole MFMA tiles, rows that cannot be read as float4 (odd widths / strides) are read element by element
    const int np = (n + 31) & ~31, kp = (k + 31) & ~31;
    const int NS = np + 4, KS = kp + 4;  // padded rows (float4-aligned)
    float *gzt = wg_lds, *xt = wg_lds + WG_ROWS * NS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int kt = kp / 32, tiles = (np / 32) * kt;
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float dbacc = 0.f;
    const int s0 = blockIdx.x * stages_per_wg, s1 = min(stages, s0 + stages_per_wg);
    const bool gv = !(n & 31) && !(gs & 3) && !(((uintptr_t)gz) & 15), xv = !(k & 31) && !(xs & 3) && !(((uintptr_t)x) & 15);
    auto stage = [&](float *dst, int DS, const float *src, int ss, int w, int wp, bool vec, long long row0) {
        if (vec) {
            const int w4 = w / 4;
            for (int e = tid; e < WG_ROWS * w4; e += 64 * WAVES) {
                const int r = e / w4, q = e - r * w4;
                const long long row = row0 + r;
                const float4 v = row < rows ? *reinterpret_cast<const float4 *>(src + row * ss + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(dst + r * DS + 4 * q) = v;
            }
        } else {
            for (int e = tid; e < WG_ROWS * wp; e += 64 * WAVES) {
                const int r = e / wp, q = e - r * wp;
                const long long row = row0 + r;
                dst[r * DS + q] = (row < rows && q < w) ? src[row * ss + q] : 0.f;
            }
        }
    };
    for (int s = s0; s < s1; ++s) {
        const long long row0 = (long long)s * WG_ROWS;
        stage(gzt, NS, gz, gs, n, np, gv, row0);
        stage(xt, KS, x, xs, k, kp, xv, row0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + WAVES * i;
            if (t < tiles) {   // wave-uniform
                const int a = t / kt, b = t - a * kt;
                const float *ga = gzt + h * NS + 32 * a + c, *xb = xt + h * KS + 32 * b + c;
#pragma unroll
                for (int s2 = 0; s2 < WG_ROWS / 2; ++s2)   // rows 2 s2 + h: A[m][kk] = gz[row][32a + m], B[kk][j] = x[row][32b + j]
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * s2 * NS], xb[2 * s2 * KS], acc[i], 0, 0, 0);
            }
        }
        if (tid < n) {
#pragma unroll 8
            for (int r = 0; r < WG_ROWS; ++r) dbacc += gzt[r * NS + tid];   // rows in ascending order
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.x * ((size_t)n * k + n);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + WAVES * i;
        if (t < tiles) {
            const int a = t / kt, b = t - a * kt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 32 * a + chan_of(r, h), col = 32 * b + c;
                if (ch < n && col < k) out[(size_t)ch * k + col] = acc[i][r];
            }
        }
    }
    if (tid < n) out[(size_t)n * k + tid] = dbacc;
}

// Workgroup partials added in a FIXED order: a workgroup takes 64 output elements; its four 64-thread groups each add a quarter of
// the partials (ascending), the quarters are then added in order.  (One thread per element walking all 256 partials alone -- nine
// workgroups for a 64 x 32 layer, 256 dependent loads each -- took 61 us: longer than the MFMA kernel in front of it.)
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ partial, int parts, int count, int nk,
                                                                float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float q[4][64];
    const int el = threadIdx.x & 63, gl = threadIdx.x >> 6, e = blockIdx.x * 64 + el;
    const int per = (parts + 3) / 4, g0 = gl * per, g1 = min(parts, g0 + per);
    float v = 0.f;
    if (e < count) {
#pragma unroll 8
        for (int g = g0; g < g1; ++g) v += partial[(size_t)g * count + e];
    }
    q[gl][el] = v;
    __syncthreads();
    if (gl != 0 || e >= count) return;
    v = ((q[0][el] + q[1][el]) + q[2][el]) + q[3][el];
    if (e < nk) dw[e] = v;
    else if (db) db[e - nk] = v;
}

bool supported(long long rows, int n, int k) {
    const int np = (n + 31) & ~31, kp = (k + 31) & ~31;
    return rows > 0 && n > 0 && k > 0 && n <= 256 && (np / 32) * (kp / 32) <= 16 * WAVES && (size_t)WG_ROWS * (np + kp + 8) * 4 <= 160 * 1024;
}
int wgs_of(long long rows) {
    const long long stages = (rows + WG_ROWS - 1) / WG_ROWS;
    return (int)(stages < MAX_WGS ? stages : MAX_WGS);   // a function of the row count alone: the summation order never depends on the device
}

}  // namespace

MCP_EXPORT size_t mcp_linear_wgrad_workspace_bytes(long long rows, int n, int k) {
    if (!supported(rows, n, k)) return 0;
    return (size_t)wgs_of(rows) * ((size_t)n * k + n) * sizeof(float);
}

MCP_EXPORT int mcp_linear_wgrad(long long rows, int n, int k, const float *gz, int gz_stride, const float *x, int x_stride, float *dw, float *db,
                                void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(gz && x && dw && workspace);
    if (!supported(rows, n, k)) return MCP_ERR_UNSUPPORTED;
    if (workspace_bytes < mcp_linear_wgrad_workspace_bytes(rows, n, k)) return MCP_ERR_BAD_ARG;
    if ((((uintptr_t)gz) | ((uintptr_t)x)) & 3 || gz_stride < n || x_stride < k) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int wgs = wgs_of(rows);
    const int stages = (int)((rows + WG_ROWS - 1) / WG_ROWS), per = (stages + wgs - 1) / wgs;
    const int np = (n + 31) & ~31, kp = (k + 31) & ~31;
    const int tiles = (np / 32) * (kp / 32), tpw = (tiles + WAVES - 1) / WAVES;
    const size_t lds = (size_t)WG_ROWS * (np + kp + 8) * sizeof(float);
    float *partial = reinterpret_cast<float *>(workspace);
#define MCP_WGRAD_GO(T)                                                                                                                        \
    do {                                                                                                                                       \
        auto kern = linear_wgrad_kernel<T>;                                                                                                    \
        static McpPerDeviceOnce attr_once;                                                                                                     \
        if (attr_once.need()) {                                                                                                                \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e_ != hipSuccess) return (int)e_;                                                                                              \
            attr_once.done();                                                                                                                  \
        }                                                                                                                                      \
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * WAVES), lds, s, rows, n, k, gz, gz_stride, x, x_stride, partial, stages, per);           \
    } while (0)
    if (tpw <= 1) MCP_WGRAD_GO(1);
    else if (tpw <= 2) MCP_WGRAD_GO(2);
    else if (tpw <= 4) MCP_WGRAD_GO(4);
    else if (tpw <= 8) MCP_WGRAD_GO(8);
    else MCP_WGRAD_GO(16);
#undef MCP_WGRAD_GO
    int rc = mcp_launch_status();
    if (rc != MCP_OK) return rc;
    const int count = n * k + n;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(mcp_divup(count, 64)), dim3(256), 0, s, partial, wgs, count, n * k, dw, db);
    return mcp_launch_status();
}
