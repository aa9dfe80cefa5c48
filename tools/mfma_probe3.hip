// Probe: what one wave per SIMD gets out of v_mfma_f32_32x32x2_f32 in the shape of k_conv128_rows' step (16 k-pairs x 4 column blocks):
//   mode 0: MFMAs only (operands in registers)      mode 1: + weight fragments read from LDS (two k-pairs ahead)
//   mode 2: + one workgroup barrier per step         mode 3: + 16 KB LDS store per step before the barrier (values from registers)
//   mode 4: + the stored values are this step's 16 KB weight chunk loaded from global memory (L2 hits)
//   mode 5: + four gathered quads per lane (random 128-B row pieces of a 45 MB array), as the kernel prefetches its A rows
//   mode 6: mode 5 without the weight loads
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe3.hip -o tools/mfma_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float* out, int steps, float seed, const f32x4* __restrict__ W4, const f32x4* __restrict__ X4, int rows) {
    __shared__ f32x4 Bs4[2][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 32 * 32; i += 256) ((f32x4*)Bs4)[i] = f32x4{seed, seed, seed, seed};
    __syncthreads();
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    f32x4 cur[4], rb[4];
    for (int j = 0; j < 4; ++j) cur[j] = f32x4{seed * lane, seed, seed * j, seed}, rb[j] = cur[j];
    int buf = 0;
    f32x4 nxt[4];
    for (int j = 0; j < 4; ++j) nxt[j] = cur[j];
    unsigned rnd = (blockIdx.x * 256 + tid) * 2654435761u;
    for (int st = 0; st < steps; ++st) {
        rnd = rnd * 1664525u + 1013904223u;
        const f32x4* pw = W4 + ((st % 36) * 1024) + tid;
        const f32x4* pa = X4 + (size_t)((rnd >> 8) % (unsigned)rows) * 32 + (st & 3) * 8 + h * 4;
        const f32x4* Bp = &Bs4[buf][(h * 16) * 32 + i32];
        f32x4 b0 = cur[0], b1 = cur[1];
        if (MODE >= 1) b0 = Bp[0], b1 = Bp[32];
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            f32x4 n0 = b0, n1 = b1;
            if (MODE >= 1 && s + 2 < 16) n0 = Bp[(s + 2) * 32], n1 = Bp[(s + 3) * 32];
            if (MODE >= 5 && s == 0)
                for (int j = 0; j < 4; ++j) nxt[j] = pa[j];
            if ((MODE == 4 || MODE == 5) && s < 4)
                for (int j = s; j < s + 2; ++j) rb[j] = pw[j * 256];
            __builtin_amdgcn_sched_barrier(0);
            const float a0 = cur[s >> 2][s & 3], a1 = cur[(s + 1) >> 2][(s + 1) & 3];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[n], acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[n], acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = n0, b1 = n1;
        }
        if (MODE >= 2) {
            buf ^= 1;
            if (MODE >= 3)
                for (int j = 0; j < 4; ++j) Bs4[buf][j * 256 + tid] = rb[j];
            __syncthreads();
        }
        if (MODE >= 5)
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(float* out, int wg_per_cu, int steps, const f32x4* W4, const f32x4* X4, int rows) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k_probe<MODE><<<grid, 256>>>(out, steps, 1e-6f, W4, X4, rows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<MODE><<<grid, 256>>>(out, steps, 1e-6f, W4, X4, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * steps * 64 * 4096.0;
    printf("mode %d  wg/cu %d  steps %4d  %8.1f us  %6.1f TFLOP/s  %.3f us per CU-step\n", MODE, wg_per_cu, steps, ms * 1e3, flop / ms / 1e9, ms * 1e3 / steps / wg_per_cu);
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    const int rows = 89046;
    f32x4 *W4, *X4;
    if (hipMalloc(&W4, 36 * 16384) != hipSuccess || hipMalloc(&X4, (size_t)rows * 512) != hipSuccess) return 1;
    hipMemset(W4, 0, 36 * 16384);
    hipMemset(X4, 0, (size_t)rows * 512);
    for (int steps : {108, 720})
        for (int w = 1; w <= 3; w += 2) {
            run<3>(out, w, steps / w, W4, X4, rows);
            run<4>(out, w, steps / w, W4, X4, rows);
            run<5>(out, w, steps / w, W4, X4, rows);
            run<6>(out, w, steps / w, W4, X4, rows);
        }
    hipFree(out);
    return 0;
}
