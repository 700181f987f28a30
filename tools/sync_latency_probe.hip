// How long does a host thread wait for a tiny kernel: hipStreamSynchronize (the runtime's default wait) against polling hipStreamQuery /
// hipEventQuery, and against a kernel that writes a flag into host-mapped pinned memory which the host polls.
//   hipcc --offload-arch=gfx950 -O2 tools/sync_latency_probe.hip -o tools/bin/sync_latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_tiny(volatile uint32_t* flag, uint32_t v) { if (threadIdx.x == 0 && flag) { __threadfence_system(); *flag = v; } }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    uint32_t* hflag; hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped); *hflag = 0;
    uint32_t* dflag; hipHostGetDevicePointer((void**)&dflag, hflag, 0);
    uint32_t* dres; hipMalloc((void**)&dres, 4096); uint32_t* hres; hipHostMalloc((void**)&hres, 4096, hipHostMallocDefault);
    const int N = 2000;
    for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipStreamSynchronize(s); }
    double t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipStreamSynchronize(s); }
    printf("launch + hipStreamSynchronize          %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("launch + poll hipStreamQuery           %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipEventRecord(e, s); while (hipEventQuery(e) == hipErrorNotReady) {} }
    printf("launch + record + poll hipEventQuery   %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, dflag, (uint32_t)(i + 1)); while (*(volatile uint32_t*)hflag != (uint32_t)(i + 1)) {} }
    printf("launch + poll host-mapped flag         %.2f us\n", (now() - t) / N);
    hipStreamSynchronize(s);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipMemcpyAsync(hres, dres, 4096, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    printf("launch + 4 KB d2h + hipStreamSynchronize %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipMemcpyAsync(hres, dres, 4096, hipMemcpyDeviceToHost, s); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("launch + 4 KB d2h + poll hipStreamQuery  %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipMemcpyAsync(dres, hres, 4096, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipMemcpyAsync(hres, dres, 4096, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    printf("4 KB h2d + launch + 4 KB d2h + sync      %.2f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipMemcpyAsync(dres, hres, 4096, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); hipMemcpyAsync(hres, dres, 4096, hipMemcpyDeviceToHost, s); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("4 KB h2d + launch + 4 KB d2h + poll      %.2f us\n", (now() - t) / N);
    // seven dependent launches (the pyramid of a single pair)
    t = now();
    for (int i = 0; i < N; i++) { for (int k = 0; k < 7; k++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("7 launches + poll                       %.2f us\n", (now() - t) / N);
    return 0;
}
