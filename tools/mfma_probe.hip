// Ceiling probe for v_mfma_f32_32x32x2_f32 on gfx950: how close to 157.3 TFLOP/s does a loop of independent fp32 MFMAs get
//   mode 0: operands in registers, 4 independent accumulators per wave
//   mode 1: B operands read from LDS each step (2 x ds_read_b32 per 4 MFMAs), A in registers
//   mode 2: like 1, plus a workgroup barrier every 64 MFMAs (the step structure of k_conv128)
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe      run: tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float seed) {
    __shared__ float Bs[32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32 * 128; i += 256) Bs[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float a[2][16];
    for (int m = 0; m < 2; ++m)
        for (int s = 0; s < 16; ++s) a[m][s] = seed * (float)(lane + s + m);
    float breg[2] = {seed, seed * 2.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float b[2];
            if (MODE == 0) {
                b[0] = breg[0];
                b[1] = breg[1];
            } else {
#pragma unroll
                for (int n = 0; n < 2; ++n) b[n] = Bs[(h * 16 + s) * 128 + n * 32 + i32];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[n], acc[m][n], 0, 0, 0);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(int wg_per_cu, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k_probe<MODE><<<grid, 256>>>(out, iters, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<MODE><<<grid, 256>>>(out, iters, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 64.0 * 4096.0;
    printf("mode %d  wg/cu %d  iters %d  %8.1f us  %6.1f TFLOP/s\n", MODE, wg_per_cu, iters, ms * 1e3, flop / ms / 1e9);
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    for (int w = 1; w <= 4; ++w) run<0>(w, 200 / w, out);
    for (int w = 1; w <= 4; ++w) run<1>(w, 200 / w, out);
    for (int w = 1; w <= 4; ++w) run<2>(w, 200 / w, out);
    hipFree(out);
    return 0;
}
