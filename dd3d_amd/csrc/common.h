// Shared helpers for the dd3d MI355X (gfx950) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DD3D_API extern "C" __attribute__((visibility("default")))

// return-code convention of include/dd3d.h: 0 ok, <0 argument error, >0 hipError_t
#define DD3D_EARG (-1)
#define DD3D_ESIZE (-2)
#define DD3D_EWS (-3)

#define DD3D_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

#define DD3D_HIP(call)                               \
    do {                                             \
        hipError_t e__ = (call);                     \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- device helpers -------------------------------------------------------------------------

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence drains EVERY
// counter (s_waitcnt vmcnt(0)): global stores issued before it and register prefetches in flight across it are waited for, which
// serialises a tile loop on HBM latency.  Use this where the barrier protects nothing but LDS data.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t hash_coord(int4 c) {
    // murmur3-style mix of the four int32 coordinates
    uint32_t h = 0x9747b28cu;
    uint32_t k[4] = {(uint32_t)c.x, (uint32_t)c.y, (uint32_t)c.z, (uint32_t)c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = k[i] * 0xcc9e2d51u;
        v = (v << 15) | (v >> 17);
        v *= 0x1b873593u;
        h ^= v;
        h = (h << 13) | (h >> 19);
        h = h * 5u + 0xe6546b64u;
    }
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ bool coord_eq(int4 a, int4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

__device__ __forceinline__ int floor_div_i(int a, int s) {
    int q = a / s;
    if ((a % s != 0) && ((a < 0) != (s < 0))) --q;
    return q;
}

// Look a coordinate up in a linear-probing table whose slots hold voxel rows of `keys` (or -1 = empty).
__device__ __forceinline__ int table_lookup(const int* __restrict__ tab, uint32_t mask, const int4* __restrict__ keys, int4 q) {
    uint32_t h = hash_coord(q) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        int r = tab[h];
        if (r < 0) return -1;
        if (coord_eq(keys[r], q)) return r;
        h = (h + 1) & mask;
    }
    return -1;
}

// wave-level sums via DPP-free shuffles (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns the exclusive prefix, sets total.
__device__ __forceinline__ int block_excl_scan_256(int v, int* total, int* s_wave /*[4+1]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int base = 0;
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = s_wave[i];
        if (i < w) base += x;
        tot += x;
    }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

// Row chunks that never straddle a group: chunk slot `c` -> (group g, rows [r0,r1)); returns false for an unused slot.
// group_row_off[G+1] lives in device memory; every group g owns ceil(rows_g / CH) consecutive slots.
__device__ __forceinline__ bool resolve_chunk(int c, const int* __restrict__ group_row_off, int G, int CH, int* g_out, int* r0, int* r1) {
    int first = 0;
    for (int g = 0; g < G; ++g) {
        int a = group_row_off[g], b = group_row_off[g + 1];
        int n = (b - a + CH - 1) / CH;
        if (c < first + n) {
            *g_out = g;
            *r0 = a + (c - first) * CH;
            *r1 = min(*r0 + CH, b);
            return true;
        }
        first += n;
    }
    return false;
}

// ---- stream-K kernel gradient of the 128-channel convolution (conv_bww_sk.hip), called by dd3d_conv_bwd_weight (conv.hip) ----
#define DD3D_BWW_WIN 512          // rows per window
#define DD3D_BWW_MAX_TILES 3072   // G * 27 tiles whose step prefix fits the workgroup's LDS table ...
#define DD3D_BWW_MAX_GROUPS 113   // ... i.e. at most this many groups (more: the row-chunk kernel)
#define DD3D_BWW_MAX_NWG 1024     // persistent workgroups (partials: two [128 x 128] slabs each)
int64_t conv128_bww_sk_ws_bytes(int64_t M_bound, int G);
int conv128_bww_sk(const float* X, const float* dY, const int32_t* nbr, int64_t ld, int64_t M_bound, const int32_t* group_row_off, int G, float* dW, void* ws,
                   int nwg, hipStream_t st);
