// Census: how many 64-thread workgroups with X KB of dynamic LDS are co-resident on one CU (MI355X)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void census(int *cnt, int *maxseen, long long spin) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID: [3:0] wave, [5:4] simd, [11:8] cu, [15:13] se  (gfx9 layout)
    const int cu = ((xcc & 7) * 16 + ((hwid >> 13) & 7)) * 16 + ((hwid >> 8) & 15);
    int me = 0;
    if (threadIdx.x == 0) {
        me = atomicAdd(&cnt[cu], 1) + 1;
        long long t0 = clock64();
        int mx = me;
        while (clock64() - t0 < spin) { int v = atomicAdd(&cnt[cu], 0); mx = v > mx ? v : mx; }
        atomicMax(&maxseen[cu], mx);
        atomicSub(&cnt[cu], 1);
    }
    __syncthreads();
}
int main() {
    int *cnt, *mx;
    hipMalloc(&cnt, 4096 * 4); hipMalloc(&mx, 4096 * 4);
    hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {1, 4, 8, 11, 16, 24, 32, 48, 64, 80, 96}) {
        hipMemset(cnt, 0, 4096 * 4); hipMemset(mx, 0, 4096 * 4);
        hipLaunchKernelGGL(census, dim3(256 * 48), dim3(64), kb * 1024, 0, cnt, mx, 200000LL);
        hipDeviceSynchronize();
        std::vector<int> h(4096);
        hipMemcpy(h.data(), mx, 4096 * 4, hipMemcpyDeviceToHost);
        int best = 0, used = 0; long long sum = 0;
        for (int v : h) if (v) { used++; sum += v; best = v > best ? v : best; }
        printf("LDS %3d KB/WG (64 thr): CUs seen %d, max resident WGs per CU %d, mean %.1f\n", kb, used, best, used ? (double)sum / used : 0.0);
    }
    return 0;
}
