// common.h -- shared device helpers for libmocopci_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mocopci_hip.h"

#define MCP_EXPORT extern "C" __attribute__((visibility("default")))

// ---- floating-point canon (must match oracle/pointset_oracle.c; built with -ffp-contract=off) ----
// sum of three squared differences: fma(dz,dz, fma(dy,dy, dx*dx))
__device__ __forceinline__ float mcp_sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
// |p|^2 as torch.sum(p ** 2, -1): rounded squares, sequential sum
__device__ __forceinline__ float mcp_sqnorm3(float x, float y, float z) { return (x * x + y * y) + z * z; }
// square_distance expansion form (mocopci.py:1152-1154): (-2*dot + |q|^2) + |r|^2
__device__ __forceinline__ float mcp_expdist(float qx, float qy, float qz, float qn, float rx, float ry, float rz, float rn) {
    float dot = __builtin_fmaf(qz, rz, __builtin_fmaf(qy, ry, qx * rx));
    return __builtin_fmaf(-2.0f, dot, qn) + rn;
}

// monotone float -> uint32 map (total order incl. negatives; -0 < +0)
__device__ __forceinline__ uint32_t mcp_ord(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float mcp_unord(uint32_t u) {
    u ^= ((u >> 31) - 1u) | 0x80000000u;
    return __uint_as_float(u);
}

// ---- DPP helpers (wave64; row = 16 lanes) ----
template <int CTRL>
__device__ __forceinline__ uint32_t mcp_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ uint32_t mcp_row_max_u32(uint32_t v) {
    uint32_t t;
    t = mcp_dpp<0xB1>(v);  v = v > t ? v : t;  // quad_perm [1,0,3,2]
    t = mcp_dpp<0x4E>(v);  v = v > t ? v : t;  // quad_perm [2,3,0,1]
    t = mcp_dpp<0x141>(v); v = v > t ? v : t;  // row_half_mirror
    t = mcp_dpp<0x140>(v); v = v > t ? v : t;  // row_mirror
    return v;
}
// max over the whole wave, returned wave-uniform: 4 row steps, then row_bcast15 / row_bcast31 carry the
// row maxima forward so that lane 63 holds the wave maximum
__device__ __forceinline__ uint32_t mcp_wave_max_u32(uint32_t v) {
    v = mcp_row_max_u32(v);
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1,3
    v = v > t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2,3
    v = v > t ? v : t;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- host-side helpers ----
struct McpProf;
void mcp_prof_begin(int kernel_id, hipStream_t s);
void mcp_prof_end(int kernel_id, hipStream_t s);

#define MCP_CHECK_ARGS(cond) \
    do {                     \
        if (!(cond)) return MCP_ERR_BAD_ARG; \
    } while (0)

// hipFuncSetAttribute applies to the CURRENT device, so "done" is remembered per device (one process may drive several
// GPUs, e.g. under nn.DataParallel, train.py:73-80).  need() only tests; the caller marks the device with done() AFTER the
// attribute calls have returned, so a second thread on the same device either sees the bit (attribute set) or repeats the
// idempotent calls itself -- it can never launch ahead of them.
struct McpPerDeviceOnce {
    unsigned long long bits = 0ull;
    static unsigned long long device_bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool need() const { return !(__atomic_load_n(&bits, __ATOMIC_ACQUIRE) & device_bit()); }
    void done() { __atomic_fetch_or(&bits, device_bit(), __ATOMIC_RELEASE); }
};

static inline int mcp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MCP_OK : (int)e;
}
static inline unsigned mcp_divup(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Which units (points, point pairs) a workgroup of a persistent-style kernel takes.  The hardware deals workgroups to the chip's eight
// XCDs round-robin (workgroup i -> XCD i mod 8) and every XCD has its own L2.  With units dealt round-robin too (unit = blockIdx * per_wg,
// step gridDim * per_wg) every L2 sees the rows that every batch element's points gather; here XCD x takes the x-th eighth of the units
// instead -- whole batch elements when the batch is a multiple of 8 -- so the rows its waves gather stay in ITS L2 (cross_kernel<64> at
// 40 x 2048 points: FETCH_SIZE 104 -> 27 thousand units per launch, round 5).  A workgroup starts at `first`, advances by `stride`, stops
// at `limit`; eighths are multiples of per_wg, so a workgroup's step never straddles two of them.  Grids that are not a multiple of 8
// keep the round-robin deal.
struct McpUnits {
    long long first, limit, stride;
};
__device__ __forceinline__ McpUnits mcp_units_by_xcd(long long total, int per_wg) {
#ifndef MCP_NO_XCD_MAP
    if (gridDim.x >= 8 && (gridDim.x & 7) == 0) {
        const long long steps = (total + per_wg - 1) / per_wg, chunk = ((steps + 7) / 8) * per_wg;
        const long long x = blockIdx.x & 7;
        const long long lim = (x + 1) * chunk;
        return {x * chunk + (long long)(blockIdx.x >> 3) * per_wg, lim < total ? lim : total, (long long)(gridDim.x >> 3) * per_wg};
    }
#endif
    return {(long long)blockIdx.x * per_wg, total, (long long)gridDim.x * per_wg};
}

// A pair of floats with element-wise arithmetic -- deliberately NOT an ext-vector type, and the library is built with
// -fno-slp-vectorize -fno-vectorize: no packed-fp32 instruction (v_pk_add/mul/fma_f32) may be formed.  Cause, established in round 3
// (tools/ab/): a v_pk_*_f32 result needs more wait states before its consumer than this compiler inserts.
// It inserts ONE, and only for VALU consumers, and only because the packed op's src0 op_sel_hi bit aliases the DST_OP_SEL bit
// of its dst-sel forwarding check (a packed op with op_sel_hi:[0,..] gets none); DS / VMEM consumers get none at all.  Observed:
// the fusion epilogue's `v_pk_add_f32 v[0:1]; ds_bpermute_b32 v2, v109, v0` (0 wait states) lost exactly one butterfly step of
// the x-sum, and furthest point sampling beside the fusion kernel picked different points in 7 of 180 launches with
// `v_pk_fma_f32; s_nop 0; v_min_f32` (1 wait state), 1 of 180 with two, 0 of 180 with three (s_nop 1 after every packed op) and
// 0 of 180 in the scalar build -- with or without the LDS index buffering that went in with the same commit.
struct mcp_f2 {
    float x, y;
};
__device__ __forceinline__ mcp_f2 operator-(mcp_f2 a, mcp_f2 b) { return mcp_f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ mcp_f2 operator+(mcp_f2 a, mcp_f2 b) { return mcp_f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ mcp_f2 operator*(mcp_f2 a, mcp_f2 b) { return mcp_f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ mcp_f2 mcp_f2_fma(mcp_f2 a, mcp_f2 b, mcp_f2 c) { return mcp_f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }

// Quotient of a flat index.  gfx950 has no integer divider: a 64-bit division is ~130 VALU instructions, a 32-bit one ~25 -- for
// a gather kernel that moves one float4 per thread, or a per-point loop of a few hundred instructions, the 64-bit form IS the
// kernel.  `fits32` = "the largest dividend fits 32 bits" (a kernel argument, so the branch is wave-uniform; true for every
// shape of this pipeline), the 64-bit path keeps very large launches correct.
__device__ __forceinline__ long long mcp_div(long long a, int b, bool fits32) {
    return fits32 ? (long long)((uint32_t)a / (uint32_t)b) : a / b;
}
__device__ __forceinline__ bool mcp_fits32(long long total) { return total <= 0xFFFFFFFFLL; }

// Raw max / min / relu.  fmaxf()/fminf() first canonicalise any operand the compiler cannot prove quiet (loop-carried values,
// loads, DPP outputs): an extra v_max_f32 x,x per operand.  The kernels' values are never signalling NaNs, so the hot loops use
// the bare instructions.  Inline asm is invisible to the compiler's hazard recogniser: it is not a "VALU" for the MFMA -> VALU,
// transcendental -> VALU or VALU -> v_readlane / v_permlane*_swap rules.  So: NEVER feed these an MFMA or v_exp/v_rcp result
// directly and never read their result with readlane / permlane-swap directly (pass through a compiler-visible instruction) --
// and because that is a property of the surrounding code, tools/isa_lint.py checks the final instruction stream of every
// kernel for exactly these distances (tests/test_isa_lint_cpu.py).
__device__ __forceinline__ float mcp_max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mcp_min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
