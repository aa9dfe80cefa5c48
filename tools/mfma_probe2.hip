// Probe: can the B operand of v_mfma_f32_32x32x2_f32 come straight from L2 (no LDS, no barriers)?  Every wave walks a 9 x 64 KB weight
// set (the 9 in-plane offsets of W[27][128][128]) in 16 KB steps, prefetching the next step's operands into registers.
//   NACC = 4: wave tile 32 x 128 (one dwordx4 per MFMA group of 4)     NACC = 2: wave tile 32 x 64 (one dwordx2 per group of 2)
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe2.hip -o tools/mfma_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ W, float* out, int units, float seed) {
    const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (tid >> 6);
    typedef float bvec __attribute__((ext_vector_type(NACC)));
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a[16];
    for (int s = 0; s < 16; ++s) a[s] = seed * (float)(lane + s);
    bvec bA[16], bB[16];
    const int colhalf = (NACC == 2) ? (wave & 1) * 16 : 0;  // in units of float4 columns... (NACC=2: 64 of 128 columns)
    auto load = [&](bvec(&b)[16], int step) {
        const int k = (step >> 2) % 9, kc = step & 3;
        const float* base = W + ((size_t)k * 128 + kc * 32 + h * 16) * 128;
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = *(const bvec*)(base + s * 128 + (NACC == 4 ? i32 * 4 : colhalf * 4 + i32 * 2));
    };
    auto mul = [&](bvec(&b)[16]) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s][n], acc[n], 0, 0, 0);
    };
    const int steps = units * 36;
    int st = (wave * 7) % 36;  // waves start at different offsets, as independent waves would
    load(bA, st);
    for (int i = 0; i < steps; i += 2) {
        load(bB, st + i + 1);
        mul(bA);
        load(bA, st + i + 2);
        mul(bB);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

__global__ __launch_bounds__(256, 2) void k_probe_half(const float* __restrict__ W, const float* __restrict__ X, float* out, int units, float seed, int gather) {
    const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (tid >> 6);
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    f32x4 raA[4], raB[4], b0[8], b1[8];
    auto loadb = [&](f32x4(&b)[8], int step, int half) {
        const int k = (step >> 2) % 9, kc = step & 3;
        const f32x4* base = (const f32x4*)W + ((size_t)k * 128 + kc * 32 + h * 16 + half * 8) * 32 + i32;
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = base[s * 32];
    };
    auto loada = [&](f32x4(&a)[4], int step) {
        // gather: row pseudo-random in 64k rows (32 MB), else a fixed row per lane
        const unsigned row = gather ? ((unsigned)(wave * 32 + i32) * 2654435761u + (unsigned)(step >> 2) * 40503u) & 65535u : (unsigned)(wave * 32 + i32) & 65535u;
        const f32x4* p = (const f32x4*)X + (size_t)row * 32 + (step & 3) * 8 + h * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = p[j];
    };
    auto mul = [&](f32x4(&a)[4], f32x4(&b)[8], int half) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[half * 2 + (s >> 2)][s & 3], b[s][n], acc[n], 0, 0, 0);
    };
    const int steps = units * 36;
    int st = (wave * 7) % 36;
    loada(raA, st);
    loadb(b0, st, 0);
    auto step = [&](f32x4(&ca)[4], f32x4(&na)[4], int i) {
        loadb(b1, i, 1);
        loada(na, i + 1);
        mul(ca, b0, 0);
        loadb(b0, i + 1, 0);
        mul(ca, b1, 1);
    };
    for (int i = 0; i < steps; i += 2) {
        step(raA, raB, st + i);
        step(raB, raA, st + i + 1);
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

static void run_half(const float* W, const float* X, float* out, int wg_per_cu, int units, int gather) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k_probe_half<<<grid, 256>>>(W, X, out, units, 1e-6f, gather);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe_half<<<grid, 256>>>(W, X, out, units, 1e-6f, gather);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * units * 36 * 16.0 * 4 * 4096.0;
    printf("half-step B + A %s  wg/cu %d  units %d  %8.1f us  %6.1f TFLOP/s\n", gather ? "gather" : "fixed ", wg_per_cu, units, ms * 1e3, flop / ms / 1e9);
}

template <int NACC>
static void run(const float* W, float* out, int wg_per_cu, int units) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    k_probe<NACC><<<grid, 256>>>(W, out, units, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<NACC><<<grid, 256>>>(W, out, units, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * units * 36 * 16.0 * NACC * 4096.0;
    printf("NACC %d  wg/cu %d  units %d  %8.1f us  %6.1f TFLOP/s\n", NACC, wg_per_cu, units, ms * 1e3, flop / ms / 1e9);
}

int main() {
    float *W, *out;
    if (hipMalloc(&W, 27 * 128 * 128 * 4 + 65536) != hipSuccess || hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    hipMemset(W, 0, 27 * 128 * 128 * 4 + 65536);
    float* X;
    if (hipMalloc(&X, 65536 * 512) != hipSuccess) return 1;
    hipMemset(X, 0, 65536 * 512);
    for (int g = 0; g < 2; ++g)
        for (int w = 1; w <= 2; ++w) run_half(W, X, out, w, 2, g);
    for (int w = 1; w <= 3; ++w) run<4>(W, out, w, 2);
    for (int w = 1; w <= 4; ++w) run<2>(W, out, w, 4);
    hipFree(W);
    hipFree(out);
    return 0;
}
