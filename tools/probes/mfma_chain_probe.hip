// Issue / latency of v_mfma_f32_16x16x32_bf16 on gfx950: one wave, NCH independent accumulator chains.
// Also calibrates clock64() (s_memtime) against wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NCH>
__global__ __launch_bounds__(64) void probe(int iters, float* sink, unsigned long long* cyc) {
    f32x4 c[NCH];
    bf16x8 x, y;
    for (int q = 0; q < 8; ++q) { x[q] = (short)(threadIdx.x + q); y[q] = (short)(threadIdx.x * 3 + q); }
#pragma unroll
    for (int j = 0; j < NCH; ++j) c[j] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12 / NCH; ++r)
#pragma unroll
            for (int j = 0; j < NCH; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c[j], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    float acc = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc += c[j][0];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}
template <int NCH>
void run() {
    float* sink; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NCH><<<1, 64>>>(iters, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NCH><<<1, 64>>>(iters, sink, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 12;
    printf("chains %d: %.2f clock64 ticks / MFMA, %.2f ns / MFMA (wall), ticks per ns %.3f\n", NCH, h / n, ms * 1e6 / n, h / (ms * 1e6));
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); return 0; }
