// ubench_launch_lds.hip -- what does a small launch cost as a function of its dynamic LDS size and workgroup size?  (The fused
// mean-shift kernel takes 16 us with zero rounds: 7 workgroups of 1024 threads with 152 KB of LDS each.)
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_launch_lds tools/ubench_launch_lds.hip && tools/bin/ubench_launch_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(unsigned *out)
{
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (s[(threadIdx.x + 1) % blockDim.x] == 0xFFFFFFFFu) out[0] = 1;
}

int main()
{
    unsigned *out;
    CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int lds[] = {4096, 32768, 65536, 131072, 155648};
    const int thr[] = {256, 1024};
    for (int t : thr) for (int l : lds) for (int grid : {7, 256}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(t), l, 0, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(t), l, 0, out);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("threads %4d  lds %6d B  grid %3d : %.2f us per launch (back to back on one stream)\n", t, l, grid, ms / 200 * 1e3);
    }
    return 0;
}
