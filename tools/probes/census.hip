// Which CU / XCD does block b land on, for a 256-thread, 72-KiB-LDS kernel (2 blocks per CU)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 0] = hwid;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t0;
    }
    // keep the block alive for a while so that residency is real
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t < (unsigned long long)spin) {}
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = smem[5];
}
int main() {
    const int nb = 1536;
    unsigned* d; hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 73728, 0, d, 20000);   // 200 us at 100 MHz
    std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b * 4], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[b * 4 + 1] & 15;
        printf("%d xcc %u se %u sh %u cu %u t %u\n", b, xcc, se, sh, cu, h[b * 4 + 2]);
    }
    return 0;
}
