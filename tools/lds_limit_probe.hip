// What one workgroup may ask for on this device: the attributes HIP reports, and whether a launch with > 64 KB of dynamic LDS runs (gfx950: 160 KB per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) { extern __shared__ int s[]; for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i; __syncthreads(); if (threadIdx.x == 0) { long t = 0; for (int i = 0; i < n; i += 1024) t += s[i]; out[0] = (int)t; } }
int main() {
    int a = 0, b = 0; hipDeviceProp_t p;
    (void)hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    (void)hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
    (void)hipGetDeviceProperties(&p, 0);
    printf("MaxSharedMemoryPerBlock %d  PerMultiprocessor %d  prop.sharedMemPerBlock %zu  prop.maxSharedMemoryPerMultiProcessor %zu\n", a, b, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    int* d; (void)hipMalloc(&d, 4);
    for (int kb : {60, 64, 96, 128, 150, 160}) {
        hipError_t e0 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), (size_t)kb * 1024, 0, d, kb * 256);
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        printf("%3d KB dynamic LDS: setattr %s, launch %s, sync %s\n", kb, hipGetErrorName(e0), hipGetErrorName(e1), hipGetErrorName(e2));
    }
    return 0;
}
