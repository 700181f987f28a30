// pcie_duplex_probe.hip - what the link gives uploads and downloads, each alone and both at once (round 5: the PCIe-inclusive bench line reaches 0.85 of the
// upload-alone rate while 18 MB of results per step travel the other way).  hipcc --offload-arch=gfx950 -O2 tools/pcie_duplex_probe.hip -o tools/bin/pcie_duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t UP = 92400000, DN = 18000000;     // bytes per step of the stereo headline: 256 images up, keypoints + descriptors + stereo results down
    const int N = 24;
    void *hu, *hd, *du, *dd;
    CK(hipHostMalloc(&hu, UP, hipHostMallocDefault)); CK(hipHostMalloc(&hd, DN, hipHostMallocDefault));
    CK(hipMalloc(&du, UP)); CK(hipMalloc(&dd, DN));
    memset(hu, 1, UP); memset(hd, 0, DN);
    hipStream_t su, sd;
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    for (int w = 0; w < 2; w++) { CK(hipMemcpyAsync(du, hu, UP, hipMemcpyHostToDevice, su)); CK(hipMemcpyAsync(hd, dd, DN, hipMemcpyDeviceToHost, sd)); }
    CK(hipDeviceSynchronize());
    double t = now();
    for (int i = 0; i < N; i++) CK(hipMemcpyAsync(du, hu, UP, hipMemcpyHostToDevice, su));
    CK(hipStreamSynchronize(su));
    const double up_alone = N * (double)UP / (now() - t) * 1e-9;
    t = now();
    for (int i = 0; i < N; i++) CK(hipMemcpyAsync(hd, dd, DN, hipMemcpyDeviceToHost, sd));
    CK(hipStreamSynchronize(sd));
    const double dn_alone = N * (double)DN / (now() - t) * 1e-9;
    // both directions, the per-step proportion: one upload and one download per step on their own streams
    t = now();
    for (int i = 0; i < N; i++) { CK(hipMemcpyAsync(du, hu, UP, hipMemcpyHostToDevice, su)); CK(hipMemcpyAsync(hd, dd, DN, hipMemcpyDeviceToHost, sd)); }
    CK(hipStreamSynchronize(su)); const double t_up = now() - t;
    CK(hipStreamSynchronize(sd)); const double t_all = now() - t;
    // both directions on ONE stream (what a handle does when its results are fetched on the stream that also carries an upload): serial by construction
    t = now();
    for (int i = 0; i < N; i++) { CK(hipMemcpyAsync(du, hu, UP, hipMemcpyHostToDevice, su)); CK(hipMemcpyAsync(hd, dd, DN, hipMemcpyDeviceToHost, su)); }
    CK(hipStreamSynchronize(su)); const double t_one = now() - t;
    // four upload streams at once (four handles each with a copy stream)
    hipStream_t s4[4]; void* d4[4];
    for (int k = 0; k < 4; k++) { CK(hipStreamCreateWithFlags(&s4[k], hipStreamNonBlocking)); CK(hipMalloc(&d4[k], UP)); }
    t = now();
    for (int i = 0; i < N; i++) CK(hipMemcpyAsync(d4[i & 3], hu, UP, hipMemcpyHostToDevice, s4[i & 3]));
    for (int k = 0; k < 4; k++) CK(hipStreamSynchronize(s4[k]));
    const double up4 = N * (double)UP / (now() - t) * 1e-9;
    printf("upload alone            %.1f GB/s\n", up_alone);
    printf("download alone          %.1f GB/s\n", dn_alone);
    printf("both, two streams       upload %.1f GB/s while a download of %.0f MB runs beside each (%.1f GB/s both ways over the whole region)\n", N * (double)UP / t_up * 1e-9, DN * 1e-6, N * (double)(UP + DN) / t_all * 1e-9);
    printf("both, one stream        %.1f GB/s of uploads (%.1f GB/s both ways)\n", N * (double)UP / t_one * 1e-9, N * (double)(UP + DN) / t_one * 1e-9);
    printf("uploads on four streams %.1f GB/s\n", up4);
    return 0;
}
