// What does an immediately returning kernel cost in a stream, behind a busy kernel, as a function of its launch shape?
// (The embed call's second-pass dispatcher: 64 workgroups x 512 threads x ~150 KB of LDS cost 4-7 us for nothing.)
// time(busy + probe) - time(busy), averaged over 300 back-to-back pairs.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[160]; };                 // a kernarg block the size of the embed kernels' parameter struct
__global__ void busy(float* sink, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void noop_small(const int* flag, int* out) {
    if (*flag == 12345) out[0] = 1;
}
__global__ void noop_big(const Big b, const int* flag, int* out) {
    extern __shared__ unsigned char smem[];
    if (*flag == 12345) { smem[threadIdx.x] = (unsigned char)b.v[threadIdx.x % 160]; out[0] = smem[0]; }
}
static float pair_ms(float* sink, int* flag, int* out, int mode, int grid, int block, int lds, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Big b = {};
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, 0, sink, 2000);
        if (mode == 1) hipLaunchKernelGGL(noop_small, dim3(grid), dim3(block), 0, 0, flag, out);
        if (mode == 2) hipLaunchKernelGGL(noop_big, dim3(grid), dim3(block), lds, 0, b, flag, out);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;
}
int main() {
    float* sink; int *flag, *out;
    hipMalloc(&sink, 2048 * 256 * 4); hipMalloc(&flag, 4); hipMalloc(&out, 4); hipMemset(flag, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(noop_big), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, 0, sink, 2000);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        const float base = pair_ms(sink, flag, out, 0, 0, 0, 0, 300);
        printf("busy kernel alone: %.2f us\n", base);
        printf("  + 1 x 64, no LDS, 16-byte kernarg:        +%.2f us\n", pair_ms(sink, flag, out, 1, 1, 64, 0, 300) - base);
        printf("  + 1 x 64, no LDS, 640-byte kernarg:       +%.2f us\n", pair_ms(sink, flag, out, 2, 1, 64, 0, 300) - base);
        printf("  + 1 x 512, 150 KB LDS:                    +%.2f us\n", pair_ms(sink, flag, out, 2, 1, 512, 150 * 1024, 300) - base);
        printf("  + 8 x 512, 150 KB LDS:                    +%.2f us\n", pair_ms(sink, flag, out, 2, 8, 512, 150 * 1024, 300) - base);
        printf("  + 64 x 512, 150 KB LDS:                   +%.2f us\n", pair_ms(sink, flag, out, 2, 64, 512, 150 * 1024, 300) - base);
        printf("  + 64 x 512, 40 KB LDS:                    +%.2f us\n", pair_ms(sink, flag, out, 2, 64, 512, 40 * 1024, 300) - base);
        printf("  + 64 x 64, no LDS:                        +%.2f us\n", pair_ms(sink, flag, out, 2, 64, 64, 0, 300) - base);
        printf("  + 256 x 512, 150 KB LDS:                  +%.2f us\n", pair_ms(sink, flag, out, 2, 256, 512, 150 * 1024, 300) - base);
    }
    return 0;
}
