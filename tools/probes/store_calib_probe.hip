// Known-bytes calibration of the WRITE_SIZE counter for the all-pairs tail's store shapes.
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -- tools/probes/store_calib_probe
// Every kernel writes each element of an R x M fp32 matrix exactly once (4 R M bytes, no read); what differs is the
// shape of a store instruction:
//   tail16   the tail's: lane (g, l15) stores 16 bytes at row 4 w + g, columns 64 sb + 4 l15 .. +3  (4 rows x 256 B per
//            instruction), ld = M = 4541: rows start on 4-byte, not 16-byte, boundaries
//   tail16a  the same with ld = 4544 (rows 128-byte aligned)
//   seg64    round 1's: lane (g, l15) stores 4 bytes at row 4 w + g, column 16 sb + l15  (4 rows x 64 B per instruction)
//   row1k    one row per wave: lane l stores 16 bytes at columns 256 sb + 4 l .. +3 (1 row x 1 KB per instruction)
//   tail16 + delay: the tail's shape AND its timing - a dependent FMA chain of ~1 us between two stores of a wave, so that
//            the 4096 resident waves interleave their 1-KB pieces like the real kernel does
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// busy work between two stores of a wave (the tail computes ~1 us between its stores): a dependent FMA chain
__device__ __forceinline__ float spin(float x, int n) {
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int R, int M, long ld, int delay) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
    const int groups = (R + 15) / 16;
    for (int rg = blockIdx.x; rg < groups; rg += gridDim.x) {
        if (MODE == 2) {                                   // row1k: 16 rows per workgroup, 4 per wave, one at a time
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rg * 16 + wave * 4 + rr;
                if (r >= R) continue;
                for (int c0 = 4 * lane; c0 < M; c0 += 256) {
                    float* dst = out + (size_t)r * ld + c0;
                    if (c0 + 3 < M) *reinterpret_cast<f32x4u*>(dst) = f32x4u{1.f, 2.f, 3.f, (float)r};
                    else for (int b = 0; b < 4; ++b) if (c0 + b < M) dst[b] = 1.f;
                }
            }
            continue;
        }
        const int r = rg * 16 + wave * 4 + g;
        if (r >= R) continue;
        if (MODE == 0) {
            float w = (float)r;
            for (int c0 = 4 * l15; c0 < M; c0 += 64) {
                float* dst = out + (size_t)r * ld + c0;
                if (delay) w = spin(w, delay);
                if (c0 + 3 < M) *reinterpret_cast<f32x4u*>(dst) = f32x4u{1.f, 2.f, 3.f, w};
                else for (int b = 0; b < 4; ++b) if (c0 + b < M) dst[b] = 1.f;
            }
        } else {
            for (int c = l15; c < M; c += 16) out[(size_t)r * ld + c] = (float)r;
        }
    }
}

int main() {
    const int R = 4541, M = 4541;
    float* d;
    hipMalloc(&d, (size_t)R * 4544 * 4 + 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(store_kernel<0>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4541, 0);
        hipLaunchKernelGGL(store_kernel<0>, dim3(1024), dim3(256), 0, 0, d + 1, R, M, (long)4544, 0);   // +4 B: never 16-B aligned
        hipLaunchKernelGGL(store_kernel<0>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4544, 0);
        hipLaunchKernelGGL(store_kernel<1>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4541, 0);
        hipLaunchKernelGGL(store_kernel<2>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4541, 0);
        hipLaunchKernelGGL(store_kernel<0>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4541, 200);     // ~1 us between a wave's stores
        hipLaunchKernelGGL(store_kernel<0>, dim3(1024), dim3(256), 0, 0, d, R, M, (long)4544, 200);
    }
    hipDeviceSynchronize();
    printf("expected bytes per launch: %zu (= %.1f KiB); launch order per repetition: tail16 ld=4541, tail16 ld=4544 misaligned by 4 B, "
           "tail16 ld=4544 aligned, seg64 ld=4541, row1k ld=4541, tail16 ld=4541 with ~1 us of work between stores, the same with ld=4544\n", (size_t)R * M * 4, R * (double)M * 4 / 1024);
    return 0;
}
