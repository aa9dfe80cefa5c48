// Bandwidth-bound feature kernels: voxel mean (TensorField.sparse()), instance norm (N7), and the fused
// affine + ReLU + max-pool (N7 affine, N8, N9) with their backward passes.  C (channels) is 128 for the norm/pool kernels:
// one row = 512 B = 32 lanes x float4, so every row access is a fully coalesced 512-B segment.
#include "common.h"
#include <stdlib.h>

namespace {

// Row-chunk size of the pooling backward: a function of the TRUE row count (group_row_off[G], read on the device), not of the launch
// bound, so a launch over a bound sums in the same order as an exact launch.  About 1024 chunks, at least 64 rows, a multiple of 8.
__host__ __device__ __forceinline__ int arp_chunk(int M) {
    int ch = (((M > 0 ? M : 1) + 1023) / 1024 + 7) / 8 * 8;
    return ch < 64 ? 64 : ch;
}
inline int64_t arp_max_slots(int64_t bound, int G) { return (ceil_div64(bound, 64) < 1024 ? ceil_div64(bound, 64) : 1024) + G; }

// ---- voxel mean -----------------------------------------------------------------------------
__global__ void k_voxel_mean_fwd(const float* __restrict__ feats, int C, const int* __restrict__ vox_off, const int* __restrict__ vox_pts, int64_t M_bound,
                                 const int* __restrict__ m_dev, float* __restrict__ F0) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t M = m_dev ? min((int64_t)*m_dev, M_bound) : M_bound;
    if (t >= M * C) return;
    int64_t v = t / C;
    int c = (int)(t - v * C);
    int a = vox_off[v], b = vox_off[v + 1];
    float s = 0.f;
    for (int j = a; j < b; ++j) s += feats[(int64_t)vox_pts[j] * C + c];  // ascending point order: deterministic
    F0[t] = s / (float)(b - a);
}

__global__ void k_voxel_mean_bwd(const float* __restrict__ dF0, int C, const int* __restrict__ inv, const int* __restrict__ vox_off, int64_t N,
                                 float* __restrict__ dfeats) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= N * C) return;
    int64_t p = t / C;
    int c = (int)(t - p * C);
    int v = inv[p];
    dfeats[t] = dF0[(int64_t)v * C + c] / (float)(vox_off[v + 1] - vox_off[v]);
}

// ---- instance norm: grid (cloud, channel quarter); a workgroup = 32 row-lanes x 8 float4 channel-lanes (128-B row segments) ------
__device__ __forceinline__ f32x4 quarter_colsum(f32x4 v, f32x4* s_red /*[32*8]*/) {
    const int c = threadIdx.x & 7, rl = threadIdx.x >> 3;
    s_red[rl * 8 + c] = v;
    __syncthreads();
    f32x4 s = s_red[c];
#pragma unroll
    for (int j = 1; j < 32; ++j) s += s_red[j * 8 + c];
    __syncthreads();
    return s;
}

// 8 row-lanes x 32 float4 channel-lanes (whole 512-B rows), used by the pooling backward
__device__ __forceinline__ f32x4 block_colsum(f32x4 v, f32x4* s_red /*[8*32]*/) {
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    s_red[rl * 32 + c4] = v;
    __syncthreads();
    f32x4 s = s_red[c4];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += s_red[j * 32 + c4];
    __syncthreads();
    return s;
}

// One workgroup per cloud: RL row-lanes x 32 float4 channel-lanes, i.e. whole 512-B rows.  Every thread keeps up to CACHE of its
// rows in registers, so a cloud of <= RL*CACHE rows is read from HBM exactly once (mean, variance and the normalised write all
// work on the registers); longer clouds re-read their tail (L2).  Sums run over a thread's rows in ascending order, then over the
// row-lanes in lane order: deterministic.
// Column sum over the RL row-lanes of the workgroup (thread = row-lane * CL + channel-lane).  The 64 / CL row-lanes that share a wave
// are folded with shuffles first, so LDS carries one partial per (wave, channel-lane) and every thread reads RL * CL / 64 of them
// (an all-to-all through LDS made these kernels LDS-bandwidth-bound).  Fixed order: shuffle tree, then waves ascending.
template <int RL, int CL = 32>
__device__ __forceinline__ f32x4 cloud_colsum(f32x4 v, f32x4* s_red /*[RL*CL/64 * CL]*/) {
    constexpr int NW = RL * CL / 64;
    const int cl = threadIdx.x % CL, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = CL; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], o, 64);
    if (lane < CL) s_red[w * CL + lane] = v;
    __syncthreads();
    f32x4 s = s_red[cl];
#pragma unroll 2
    for (int j = 1; j < NW; ++j) s += s_red[j * CL + cl];
    __syncthreads();
    return s;
}

// NV column sums at once (one LDS exchange and two barriers instead of NV of each): same fold order as cloud_colsum per quantity.
template <int RL, int CL, int NV>
__device__ __forceinline__ void cloud_colsums(f32x4 (&v)[NV], f32x4* s_red /*[NV][RL*CL/64 * CL]*/) {
    constexpr int NW = RL * CL / 64;
    const int cl = threadIdx.x % CL, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
#pragma unroll
        for (int o = CL; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[q][j] += __shfl_xor(v[q][j], o, 64);
        if (lane < CL) s_red[(q * NW + w) * CL + lane] = v[q];
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        f32x4 s = s_red[(q * NW) * CL + cl];
#pragma unroll 2
        for (int j = 1; j < NW; ++j) s += s_red[(q * NW + j) * CL + cl];
        v[q] = s;
    }
    lds_barrier();
}

template <int RL, int CACHE>
__global__ __launch_bounds__(RL * 32) void k_inorm_fwd(const f32x4* __restrict__ Y, const int* __restrict__ seg_off, f32x4* __restrict__ xhat,
                                                       f32x4* __restrict__ inv_std) {
    __shared__ f32x4 s_red[RL * 32];
    const int b = blockIdx.x, c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = seg_off[b], r1 = seg_off[b + 1];
    const int n = r1 - r0;
    if (n <= 0) {
        if (rl == 0) inv_std[(int64_t)b * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[CACHE];
    f32x4 s = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {  // clamped row index: branch-free, all CACHE loads in flight together
        int r = r0 + rl + i * RL;
        v[i] = Y[(int64_t)min(r, r1 - 1) * 32 + c4];
        if (r >= r1) v[i] = zero;
        s += v[i];
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) s += Y[(int64_t)r * 32 + c4];
    const f32x4 mean = cloud_colsum<RL>(s, s_red) / (float)n;
    f32x4 q = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        f32x4 d = v[i] - mean;
        if (r < r1) q += d * d;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        f32x4 d = Y[(int64_t)r * 32 + c4] - mean;
        q += d * d;
    }
    const f32x4 var = cloud_colsum<RL>(q, s_red) / (float)n;
    f32x4 is;
#pragma unroll
    for (int j = 0; j < 4; ++j) is[j] = 1.0f / sqrtf(var[j] + 1e-8f);
    if (rl == 0) inv_std[(int64_t)b * 32 + c4] = is;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        if (r < r1) xhat[(int64_t)r * 32 + c4] = (v[i] - mean) * is;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) xhat[(int64_t)r * 32 + c4] = (Y[(int64_t)r * 32 + c4] - mean) * is;
}

// dY = (g - xhat * mean(g*xhat) - mean(g)) * inv_std     (MinkowskiInstanceNormFunction.backward)
template <int RL, int CACHE>
__global__ __launch_bounds__(RL * 32) void k_inorm_bwd(const f32x4* __restrict__ g, const f32x4* __restrict__ xhat, const f32x4* __restrict__ inv_std,
                                                       const int* __restrict__ seg_off, f32x4* __restrict__ dY) {
    __shared__ f32x4 s_red[RL * 32];
    const int b = blockIdx.x, c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = seg_off[b], r1 = seg_off[b + 1];
    const int n = r1 - r0;
    if (n <= 0) return;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 is = inv_std[(int64_t)b * 32 + c4];
    f32x4 vg[CACHE], vx[CACHE];
    f32x4 s1 = zero, s2 = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        vg[i] = g[(int64_t)min(r, r1 - 1) * 32 + c4];
        vx[i] = xhat[(int64_t)min(r, r1 - 1) * 32 + c4];
        if (r >= r1) vg[i] = vx[i] = zero;
        s1 += vg[i];
        s2 += vg[i] * vx[i];
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        f32x4 gg = g[(int64_t)r * 32 + c4];
        s1 += gg;
        s2 += gg * xhat[(int64_t)r * 32 + c4];
    }
    const f32x4 m1 = cloud_colsum<RL>(s1, s_red) / (float)n;
    const f32x4 m2 = cloud_colsum<RL>(s2, s_red) / (float)n;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        if (r < r1) dY[(int64_t)r * 32 + c4] = (vg[i] - vx[i] * m2 - m1) * is;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        int64_t o = (int64_t)r * 32 + c4;
        dY[o] = (g[o] - xhat[o] * m2 - m1) * is;
    }
}

// ---- fused affine + ReLU + max-pool (k=2, s=2) ------------------------------------------------
// one pooled row x one float4 channel lane of k_arp_fwd (shared with the fused instance-norm kernel)
// NORM: ``xhat`` holds the norm's INPUT rows and a row is normalised as it is read, (row - mu) * is -- the same two roundings that
// produce a stored xhat row, so both forms give the same bits
template <bool NORM = false>
__device__ __forceinline__ void arp_row(const f32x4* __restrict__ xhat, const int* __restrict__ kids, int64_t ld, int Kp, int64_t o, int c4, f32x4 ww,
                                        f32x4 bb, int use_affine_relu, f32x4* __restrict__ out, int4* __restrict__ arg,
                                        f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, f32x4 is = (f32x4){1.f, 1.f, 1.f, 1.f}) {
    const float ninf = -__builtin_inff();
    f32x4 best = {ninf, ninf, ninf, ninf};
    int4 a = make_int4(-1, -1, -1, -1);
    // all child indices first, then all child rows (absent children read row 0 and are masked): 8 gathers in flight instead of a
    // serial index -> row -> compare chain per child
    int ch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = kids[(int64_t)min(k, Kp - 1) * ld + o];
    f32x4 rows[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[k] = xhat[(int64_t)max(ch[k], 0) * 32 + c4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = ch[k];
        if (k >= Kp || i < 0) continue;
        f32x4 v = rows[k];
        if (NORM) v = (v - mu) * is;
        if (use_affine_relu) {
            v = v * ww + bb;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (v[0] > best[0]) { best[0] = v[0]; a.x = i; }
        if (v[1] > best[1]) { best[1] = v[1]; a.y = i; }
        if (v[2] > best[2]) { best[2] = v[2]; a.z = i; }
        if (v[3] > best[3]) { best[3] = v[3]; a.w = i; }
    }
    out[o * 32 + c4] = best;
    arg[o * 32 + c4] = a;
}

// out[o][c] = max over existing children i of relu(xhat[i][c]*w[c]+b[c]); first maximum in offset order wins; arg = child row
__global__ __launch_bounds__(256) void k_arp_fwd(const f32x4* __restrict__ xhat, const f32x4* __restrict__ w, const f32x4* __restrict__ bias,
                                                 int use_affine_relu, const int* __restrict__ kids, int64_t ld, int Kp, int64_t M_bound,
                                                 const int* __restrict__ m_dev, f32x4* __restrict__ out, int4* __restrict__ arg) {
    const int c4 = threadIdx.x & 31;
    int64_t o = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int64_t M_out = m_dev ? min((int64_t)*m_dev, M_bound) : M_bound;
    if (o >= M_out) return;
    f32x4 ww = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
    if (use_affine_relu) {
        ww = w[c4];
        bb = bias[c4];
    }
    arp_row(xhat, kids, ld, Kp, o, c4, ww, bb, use_affine_relu, out, arg);
}

// Instance norm forward + affine + ReLU + max-pool in one launch per level (one workgroup per cloud).
template <int RL, int CACHE>
__global__ __launch_bounds__(RL * 32) void k_inorm_arp_fwd(const f32x4* __restrict__ Y, const int* __restrict__ seg_off, f32x4* __restrict__ xhat,
                                                           f32x4* __restrict__ inv_std, const f32x4* __restrict__ w, const f32x4* __restrict__ bias,
                                                           const int* __restrict__ kids, int64_t ld, int Kp, const int* __restrict__ seg_out,
                                                           f32x4* __restrict__ pooled, int4* __restrict__ arg) {
    __shared__ f32x4 s_red[RL * 32];
    const int b = blockIdx.x, c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = seg_off[b], r1 = seg_off[b + 1];
    const int n = r1 - r0;
    if (n <= 0) {  // a cloud without voxels has no pooled rows either
        if (rl == 0) inv_std[(int64_t)b * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[CACHE];
    f32x4 s = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {  // clamped row index: branch-free, all CACHE loads in flight together
        int r = r0 + rl + i * RL;
        v[i] = Y[(int64_t)min(r, r1 - 1) * 32 + c4];
        if (r >= r1) v[i] = zero;
        s += v[i];
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) s += Y[(int64_t)r * 32 + c4];
    const f32x4 mean = cloud_colsum<RL>(s, s_red) / (float)n;
    f32x4 q = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        f32x4 d = v[i] - mean;
        if (r < r1) q += d * d;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        f32x4 d = Y[(int64_t)r * 32 + c4] - mean;
        q += d * d;
    }
    const f32x4 var = cloud_colsum<RL>(q, s_red) / (float)n;
    f32x4 is;
#pragma unroll
    for (int j = 0; j < 4; ++j) is[j] = 1.0f / sqrtf(var[j] + 1e-8f);
    if (rl == 0) inv_std[(int64_t)b * 32 + c4] = is;
    const f32x4 ww = w[c4], bb = bias[c4];
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        int r = r0 + rl + i * RL;
        if (r < r1) xhat[(int64_t)r * 32 + c4] = (v[i] - mean) * is;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) xhat[(int64_t)r * 32 + c4] = (Y[(int64_t)r * 32 + c4] - mean) * is;
    // affine + ReLU + max-pool of THIS cloud's pooled rows: their children are rows of this cloud, just written by this workgroup
    // (the barrier makes them visible to it), so they come back from L2 / L1 instead of a second pass over HBM in another launch
    __syncthreads();
    for (int64_t o = seg_out[b] + rl; o < seg_out[b + 1]; o += RL) arp_row(xhat, kids, ld, Kp, o, c4, ww, bb, 1, pooled, arg);
}


// ---- the same level WITHOUT the normalised rows in memory (dd3d_inorm_arp_fwd_y) ----
// The per-cloud mean is kept beside inv_std; the pooled rows normalise their children as they gather them from Y (rows this workgroup
// has just read: L1 / L2), and the backward kernels do the same.  Saves a full [rows, 128] write per level.
// A cloud is a chain of dependent memory round trips (row range -> rows -> two reductions -> child indices -> child rows), so what
// counts is how many clouds a CU has in flight: 256-thread workgroups (8 row-lanes x 32 channel-lanes) instead of 512-thread ones, the
// first pooled row's child indices loaded before the statistics, the next row's before the current one is reduced.
// Bit-compatibility with k_inorm_fwd<16, 8>: a thread works as TWO of that kernel's 16 row-lanes (wave w, half h: lanes 4w + h and
// 4w + 2 + h), keeps their sums apart, and the fold is that kernel's: the halves of a wave by one shuffle, the eight pair sums through
// LDS in ascending order.  VC rows per virtual lane stay in registers (16 VC rows of the cloud); longer clouds re-read their tail.
template <int VC>
__device__ __forceinline__ void cloud_stats16(const f32x4* __restrict__ Y, int r0, int r1, int la, int c4, int w, int lane, f32x4* s_red,
                                              f32x4* mean_o, f32x4* is_o) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int n = r1 - r0;
    f32x4 va[VC], vb[VC];
    f32x4 sa = zero, sb = zero;
#pragma unroll
    for (int i = 0; i < VC; ++i) {  // clamped row index: branch-free, all loads in flight together
        const int ra = r0 + la + i * 16, rb = ra + 2;
        va[i] = Y[(int64_t)min(ra, r1 - 1) * 32 + c4];
        vb[i] = Y[(int64_t)min(rb, r1 - 1) * 32 + c4];
    }
#pragma unroll
    for (int i = 0; i < VC; ++i) {
        const int ra = r0 + la + i * 16, rb = ra + 2;
        if (ra >= r1) va[i] = zero;
        if (rb >= r1) vb[i] = zero;
        sa += va[i];
        sb += vb[i];
    }
    for (int r = r0 + la + VC * 16; r < r1; r += 16) sa += Y[(int64_t)r * 32 + c4];
    for (int r = r0 + la + 2 + VC * 16; r < r1; r += 16) sb += Y[(int64_t)r * 32 + c4];
    auto fold = [&](f32x4 a, f32x4 b2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] += __shfl_xor(a[j], 32, 64);
            b2[j] += __shfl_xor(b2[j], 32, 64);
        }
        if (lane < 32) {
            s_red[(2 * w) * 32 + lane] = a;
            s_red[(2 * w + 1) * 32 + lane] = b2;
        }
        __syncthreads();
        f32x4 s = s_red[c4];
#pragma unroll 2
        for (int j = 1; j < 8; ++j) s += s_red[j * 32 + c4];
        __syncthreads();
        return s;
    };
    const f32x4 mean = fold(sa, sb) / (float)n;
    f32x4 qa = zero, qb = zero;
#pragma unroll
    for (int i = 0; i < VC; ++i) {
        const int ra = r0 + la + i * 16, rb = ra + 2;
        f32x4 d = va[i] - mean;
        if (ra < r1) qa += d * d;
        d = vb[i] - mean;
        if (rb < r1) qb += d * d;
    }
    for (int r = r0 + la + VC * 16; r < r1; r += 16) {
        f32x4 d = Y[(int64_t)r * 32 + c4] - mean;
        qa += d * d;
    }
    for (int r = r0 + la + 2 + VC * 16; r < r1; r += 16) {
        f32x4 d = Y[(int64_t)r * 32 + c4] - mean;
        qb += d * d;
    }
    const f32x4 var = fold(qa, qb) / (float)n;
    f32x4 is;
#pragma unroll
    for (int j = 0; j < 4; ++j) is[j] = 1.0f / sqrtf(var[j] + 1e-8f);
    *mean_o = mean;
    *is_o = is;
}

template <int VCMAX>
__global__ __launch_bounds__(256) void k_inorm_arp_fwd_y(const f32x4* __restrict__ Y, const int* __restrict__ seg_off, f32x4* __restrict__ mean_out,
                                                         f32x4* __restrict__ inv_std, const f32x4* __restrict__ w, const f32x4* __restrict__ bias,
                                                         const int* __restrict__ kids, int64_t ld, int Kp, const int* __restrict__ seg_out,
                                                         f32x4* __restrict__ pooled, int4* __restrict__ arg) {
    __shared__ f32x4 s_red[8 * 32];
    const int b = blockIdx.x, c4 = threadIdx.x & 31, rl = threadIdx.x >> 5, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = seg_off[b], r1 = seg_off[b + 1];
    const int64_t o0 = seg_out[b], o1 = seg_out[b + 1];
    const int n = r1 - r0;
    if (n <= 0) {  // a cloud without voxels has no pooled rows either
        if (rl == 0) {
            inv_std[(int64_t)b * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mean_out[(int64_t)b * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    // child indices of this thread's first pooled row: in flight while the statistics run
    int64_t o = o0 + rl;
    int ch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ch[k] = kids[(int64_t)min(k, Kp - 1) * ld + min(o, o1 - 1)];
    const f32x4 ww = w[c4], bb = bias[c4];
    f32x4 mean, is;
    const int la = 4 * wv + (lane >> 5);
    if (n <= 32) cloud_stats16<2>(Y, r0, r1, la, c4, wv, lane, s_red, &mean, &is);
    else if (VCMAX <= 4 || n <= 64) cloud_stats16<4>(Y, r0, r1, la, c4, wv, lane, s_red, &mean, &is);
    else cloud_stats16<VCMAX>(Y, r0, r1, la, c4, wv, lane, s_red, &mean, &is);
    if (rl == 0) {
        inv_std[(int64_t)b * 32 + c4] = is;
        mean_out[(int64_t)b * 32 + c4] = mean;
    }
    // affine + ReLU + max-pool of this cloud's pooled rows (arp_row<true>, with the child indices one row ahead)
    const float ninf = -__builtin_inff();
    for (; o < o1; o += 8) {
        f32x4 rows[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rows[k] = Y[(int64_t)max(ch[k], 0) * 32 + c4];
        int cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = ch[k];
        const int64_t on = o + 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) ch[k] = kids[(int64_t)min(k, Kp - 1) * ld + min(on, o1 - 1)];
        f32x4 best = {ninf, ninf, ninf, ninf};
        int4 a = make_int4(-1, -1, -1, -1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = cur[k];
            if (k >= Kp || i < 0) continue;
            f32x4 v = (rows[k] - mean) * is;
            v = v * ww + bb;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            if (v[0] > best[0]) { best[0] = v[0]; a.x = i; }
            if (v[1] > best[1]) { best[1] = v[1]; a.y = i; }
            if (v[2] > best[2]) { best[2] = v[2]; a.z = i; }
            if (v[3] > best[3]) { best[3] = v[3]; a.w = i; }
        }
        pooled[o * 32 + c4] = best;
        arg[o * 32 + c4] = a;
    }
}


// Backward of pool -> ReLU -> affine for every INPUT row i (each has exactly one parent):
//   gy = (arg[parent[i]] == i) ? g_pool[parent[i]] : 0 ; ga = (xhat*w+b > 0) ? gy : 0 ; g_xhat = ga * w
//   per row-chunk partial sums of ga*xhat (-> d weight) and ga (-> d bias); optional ga_out and an optional extra term
//   g_xhat += ga_prev * gw_extra[group]   (second-order term of the matching loss through the affine weight gradient)
__global__ __launch_bounds__(256) void k_arp_bwd(const f32x4* __restrict__ g_pool, const int4* __restrict__ arg, const int* __restrict__ parent,
                                                 const f32x4* __restrict__ xhat, const f32x4* __restrict__ w, const f32x4* __restrict__ bias,
                                                 int use_affine_relu, const int* __restrict__ group_row_off, int G,
                                                 f32x4* __restrict__ g_xhat, f32x4* __restrict__ ga_out, const f32x4* __restrict__ ga_prev,
                                                 const f32x4* __restrict__ gw_extra, f32x4* __restrict__ partial /*[slots][2][32]*/,
                                                 int* __restrict__ slab_valid) {
    __shared__ f32x4 s_red[256];
    const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
    int g, r0, r1;
    if (!resolve_chunk(blockIdx.x, group_row_off, G, arp_chunk(group_row_off[G]), &g, &r0, &r1)) return;
    f32x4 ww = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
    if (use_affine_relu) {
        ww = w[c4];
        bb = bias[c4];
    }
    f32x4 ex = {0.f, 0.f, 0.f, 0.f};
    if (gw_extra) ex = gw_extra[(int64_t)g * 32 + c4];
    f32x4 sw = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    for (int i = r0 + rl; i < r1; i += 8) {
        int p = parent[i];
        int64_t po = (int64_t)p * 32 + c4, io = (int64_t)i * 32 + c4;
        int4 a = arg[po];
        f32x4 gp = g_pool[po];
        f32x4 xh = xhat[io];
        f32x4 ga;
        ga[0] = a.x == i ? gp[0] : 0.f;
        ga[1] = a.y == i ? gp[1] : 0.f;
        ga[2] = a.z == i ? gp[2] : 0.f;
        ga[3] = a.w == i ? gp[3] : 0.f;
        if (use_affine_relu) {
            f32x4 y = xh * ww + bb;
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] = y[j] > 0.f ? ga[j] : 0.f;
        }
        sw += ga * xh;
        sb += ga;
        f32x4 gx = ga * ww;
        if (ga_prev) gx += ga_prev[io] * ex;
        g_xhat[io] = gx;
        if (ga_out) ga_out[io] = ga;
    }
    f32x4 tw = block_colsum(sw, s_red);
    f32x4 tb = block_colsum(sb, s_red);
    if (partial && rl == 0) {
        partial[((int64_t)blockIdx.x * 2 + 0) * 32 + c4] = tw;
        partial[((int64_t)blockIdx.x * 2 + 1) * 32 + c4] = tb;
        if (c4 == 0) slab_valid[blockIdx.x] = 1;
    }
}

// out[g][j][c4] = sum over the chunk slots of group g (every slot of a group is written by k_arp_bwd); 4 slot-lanes per output sum
// slots s = lane, lane+4, ... in order and are then combined in lane order: deterministic.  j in {0: d weight, 1: d bias}
__global__ __launch_bounds__(256) void k_arp_reduce(const f32x4* __restrict__ partial, const int* __restrict__ group_row_off, int G,
                                                    f32x4* __restrict__ gw_w, f32x4* __restrict__ gw_b) {
    __shared__ f32x4 s_red[256];
    const int CH = arp_chunk(group_row_off[G]);
    const int g = blockIdx.x, c4 = threadIdx.x & 31, j = (threadIdx.x >> 5) & 1, sl = threadIdx.x >> 6;
    int first = 0;
    for (int q = 0; q < g; ++q) first += (group_row_off[q + 1] - group_row_off[q] + CH - 1) / CH;
    int n = (group_row_off[g + 1] - group_row_off[g] + CH - 1) / CH;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = first + sl; c < first + n; c += 4) s += partial[((int64_t)c * 2 + j) * 32 + c4];
    s_red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0) {
        int t = threadIdx.x;
        f32x4 tot = s_red[t] + s_red[t + 64] + s_red[t + 128] + s_red[t + 192];
        (j == 0 ? gw_w : gw_b)[(int64_t)g * 32 + c4] = tot;
    }
}

// One cloud x one channel slice.  WAVE = false: the slice belongs to a workgroup of RL row-lanes x CL float4 channel-lanes (column sums through
// LDS, two barriers).  WAVE = true: the slice belongs to ONE wave of 8 row-lanes x 8 channel-lanes (column sums by three shuffle steps, no
// barrier, no LDS): a small cloud is a short dependent chain -- row range -> parent -> pooled gradient -> sums -> write -- and four
// independent waves per cloud run four such chains where a 256-thread workgroup per slice ran one (level 2 of cfg2: 29 rows per cloud).
template <int RL, int CACHE, int CL, bool WAVE>
__device__ __forceinline__ void arp_inorm_bwd_slice(const f32x4* __restrict__ g_pool, const int4* __restrict__ arg, const int* __restrict__ parent,
                                                    const f32x4* __restrict__ xhat, const f32x4* __restrict__ inv_std, const f32x4* __restrict__ w,
                                                    const f32x4* __restrict__ bias, const int* __restrict__ cloud_group, f32x4* __restrict__ dY,
                                                    f32x4* __restrict__ ga_out, const f32x4* __restrict__ ga_prev, const f32x4* __restrict__ gw_extra,
                                                    f32x4* __restrict__ part, f32x4* s_red, int b, int r0, int r1, int c4, int rl,
                                                    const f32x4* __restrict__ mean) {
    const int n = r1 - r0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 ww = w[c4], bb = bias[c4];
    const f32x4 is = inv_std[(int64_t)b * 32 + c4];  // issued with the first loads, not after the reductions (one round trip less)
    // mean given: ``xhat`` is the norm's INPUT and a row is normalised as it is read (the bits a stored xhat row would have)
    const bool from_y = mean != nullptr;
    f32x4 mu = zero;
    if (from_y) mu = mean[(int64_t)b * 32 + c4];
    f32x4 ex = zero;
    if (gw_extra) ex = gw_extra[(int64_t)(cloud_group ? cloud_group[b] : 0) * 32 + c4];
    // ga (ReLU-masked pooled gradient) and g_xhat of row i
    auto row = [&](int i, f32x4& xh, f32x4& ga, f32x4& gx) {
        const int64_t po = (int64_t)parent[i] * 32 + c4, io = (int64_t)i * 32 + c4;
        const int4 a = arg[po];
        const f32x4 gp = g_pool[po];
        xh = xhat[io];
        if (from_y) xh = (xh - mu) * is;
        ga[0] = a.x == i ? gp[0] : 0.f;
        ga[1] = a.y == i ? gp[1] : 0.f;
        ga[2] = a.z == i ? gp[2] : 0.f;
        ga[3] = a.w == i ? gp[3] : 0.f;
        const f32x4 y = xh * ww + bb;
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = y[j] > 0.f ? ga[j] : 0.f;
        gx = ga * ww;
        if (ga_prev) gx += ga_prev[io] * ex;
    };
    f32x4 vg[CACHE], vx[CACHE];
    f32x4 sw = zero, sb = zero, s1 = zero, s2 = zero;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {  // clamped row index: no branch around the loads, the CACHE rows' gathers overlap
        const int r = r0 + rl + i * RL;
        f32x4 ga;
        row(min(r, r1 - 1), vx[i], ga, vg[i]);
        if (r >= r1) ga = vg[i] = vx[i] = zero;
        if (ga_out && r < r1) ga_out[(int64_t)r * 32 + c4] = ga;
        sw += ga * vx[i];
        sb += ga;
        s1 += vg[i];
        s2 += vg[i] * vx[i];
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        f32x4 xh, ga, gx;
        row(r, xh, ga, gx);
        if (ga_out) ga_out[(int64_t)r * 32 + c4] = ga;
        sw += ga * xh;
        sb += ga;
        s1 += gx;
        s2 += gx * xh;
    }
    f32x4 v[4] = {s1, s2, sw, sb};
    if (WAVE) {  // fold the 8 row-lanes of the wave (lane = row-lane * 8 + channel-lane): fixed butterfly order
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[q][j] += __shfl_xor(v[q][j], o, 64);
    } else if (part) {  // all four column sums in one exchange (part is a kernel argument: the branch is uniform)
        cloud_colsums<RL, CL, 4>(v, s_red);
    } else {
        f32x4 v2[2] = {s1, s2};
        cloud_colsums<RL, CL, 2>(v2, s_red);
        v[0] = v2[0];
        v[1] = v2[1];
    }
    const f32x4 m1 = v[0] / (float)n, m2 = v[1] / (float)n;
    if (part && rl == 0) {
        part[((int64_t)b * 2 + 0) * 32 + c4] = v[2];
        part[((int64_t)b * 2 + 1) * 32 + c4] = v[3];
    }
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        const int r = r0 + rl + i * RL;
        if (r < r1) dY[(int64_t)r * 32 + c4] = (vg[i] - vx[i] * m2 - m1) * is;
    }
    for (int r = r0 + rl + CACHE * RL; r < r1; r += RL) {
        f32x4 xh, ga, gx;
        row(r, xh, ga, gx);
        dY[(int64_t)r * 32 + c4] = (gx - xh * m2 - m1) * is;
    }
}

// Backward of pool -> ReLU -> affine -> instance norm in ONE pass over a cloud (k_arp_bwd followed by k_inorm_bwd, without the
// round trip of g_xhat through HBM): the thread's rows of g_xhat and xhat stay in registers between the statistics and the final
// write.  Per-cloud partial sums of ga*xhat / ga go to part[b][2][32]; k_cloud_group_sum folds them per group in cloud order.
// grid (cloud, 32 / CL): a workgroup takes CL float4 channel-lanes (CL * 16 B of every row), so several workgroups per CU overlap
// their gather latencies (CL = 8: 128-B pieces of a row, one cache line).  Clouds of at most WAVE_ROWS rows are taken by workgroup
// (cloud, 0) alone, one wave per channel slice (see arp_inorm_bwd_slice); the choice depends on the cloud's row count only, so a bound-sized
// launch and an exact one sum in the same order.
constexpr int ARP_WAVE_CACHE = 6;    // rows per lane kept in registers in wave mode (8 row-lanes: 48 rows; longer clouds re-read their tail from L2)
constexpr int ARP_WAVE_ROWS = 128;   // clouds up to this many rows go wave by wave (measured at cfg2: 65.6 -> 30.8 us at level 2, 101 -> 88 us at level 1)
constexpr int ARP_WAVE_ROWS_FEW = 64;      // ... in a launch over FEW clouds (the synthetic prefix of the second-order pass, a shard's batch) only
constexpr int ARP_FEW_CLOUDS = 1024;       // up to this many rows: one workgroup per cloud leaves such a launch short of waves, four per cloud
                                           // fill the chip (measured, round 3: second-order launches of 100 clouds 83 -> 61 and 48 -> 36 us,
                                           // a 2-class shard's step 0.463 -> 0.436 ms; the merged launch of 2660 clouds is faster wave by wave)
template <int RL, int CACHE, int CL>
__global__ __launch_bounds__(RL * CL) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_arp_inorm_bwd(const f32x4* __restrict__ g_pool, const int4* __restrict__ arg, const int* __restrict__ parent,
                                                           const f32x4* __restrict__ xhat, const f32x4* __restrict__ inv_std,
                                                           const f32x4* __restrict__ w, const f32x4* __restrict__ bias,
                                                           const int* __restrict__ seg_off, const int* __restrict__ cloud_group,
                                                           f32x4* __restrict__ dY, f32x4* __restrict__ ga_out_all, int ga_clouds,
                                                           const f32x4* __restrict__ ga_prev, const f32x4* __restrict__ gw_extra,
                                                           f32x4* __restrict__ part /*[B][2][32] or null*/, int wave_rows,
                                                           const f32x4* __restrict__ mean /*[B][32] or null: xhat is the norm's input*/) {
    static_assert(RL * CL == 256 && CL == 8, "four slices of eight float4 channel-lanes");
    __shared__ f32x4 s_red[4 * (RL * CL / 64) * CL];
    const int b = blockIdx.x;
    const int r0 = seg_off[b], r1 = seg_off[b + 1];
    const int n = r1 - r0;
    f32x4* ga_out = (ga_clouds <= 0 || b < ga_clouds) ? ga_out_all : nullptr;  // only the leading clouds keep their masked gradient
    if (n <= 0) {
        const int c4 = blockIdx.y * CL + threadIdx.x % CL, rl = threadIdx.x / CL;
        if (part && rl == 0) {
            part[((int64_t)b * 2 + 0) * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
            part[((int64_t)b * 2 + 1) * 32 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    if (n <= wave_rows) {  // small cloud: one wave per channel slice, all four slices in workgroup (cloud, 0)
        if (blockIdx.y) return;
        const int lane = threadIdx.x & 63;
        arp_inorm_bwd_slice<8, ARP_WAVE_CACHE, 8, true>(g_pool, arg, parent, xhat, inv_std, w, bias, cloud_group, dY, ga_out, ga_prev, gw_extra, part, s_red, b,
                                                           r0, r1, (threadIdx.x >> 6) * 8 + (lane & 7), lane >> 3, mean);
        return;
    }
    arp_inorm_bwd_slice<RL, CACHE, CL, false>(g_pool, arg, parent, xhat, inv_std, w, bias, cloud_group, dY, ga_out, ga_prev, gw_extra, part, s_red, b, r0, r1,
                                              blockIdx.y * CL + threadIdx.x % CL, threadIdx.x / CL, mean);
}

// out_w/out_b[g][c4] = sum over the clouds of group g of part[b][0/1][c4]; 16 cloud-lanes take clouds in order (a short dependent
// chain of L2 reads each), combined in lane order: deterministic.  grid (group, 4 channel slices) of 256-thread workgroups: a
// 1024-thread workgroup needs a quarter of a CU's wave slots at once and waited ~70 us for them next to the 128-channel kernel gradient.
__global__ __launch_bounds__(256) void k_cloud_group_sum(const f32x4* __restrict__ part, const int* __restrict__ group_cloud_off, f32x4* __restrict__ gw_w,
                                                         f32x4* __restrict__ gw_b) {
    __shared__ f32x4 s_red[256];
    const int g = blockIdx.x, c4 = blockIdx.y * 8 + (threadIdx.x & 7), j = (threadIdx.x >> 3) & 1, sl = threadIdx.x >> 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = group_cloud_off[g] + sl;
    const int b1 = group_cloud_off[g + 1];
    for (; b < b1; b += 16 * 8) {  // eight of this lane's clouds in flight (clamped index, no branch around the loads), added in cloud order
        f32x4 x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = part[((int64_t)min(b + 16 * q, b1 - 1) * 2 + j) * 32 + c4];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b + 16 * q < b1) s += x[q];
    }
    s_red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0) {
        f32x4 t = s_red[threadIdx.x];
#pragma unroll 4
        for (int q = 1; q < 16; ++q) t += s_red[threadIdx.x + 16 * q];
        (j == 0 ? gw_w : gw_b)[(int64_t)g * 32 + c4] = t;
    }
}

// g_xhat[i] = ga[i] * gw[group(i)]  (outer-backward seed through an affine weight gradient)
__global__ void k_scale_rows_by_group(const f32x4* __restrict__ ga, const f32x4* __restrict__ gw, const int* __restrict__ group_row_off, int G,
                                      int64_t M, f32x4* __restrict__ out) {
    const int c4 = threadIdx.x & 31;
    int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (i >= M) return;
    int g = 0;
    while (g + 1 < G && i >= group_row_off[g + 1]) ++g;
    out[i * 32 + c4] = ga[i * 32 + c4] * gw[(int64_t)g * 32 + c4];
}

}  // namespace

DD3D_API int dd3d_voxel_mean_fwd(const float* feats, int64_t N, int C, const int32_t* vox_off, const int32_t* vox_pts, int64_t M, const int32_t* m_dev,
                                 float* F0, void* stream) {
    if (N < 0 || M < 0 || C <= 0) return DD3D_EARG;
    if (M == 0) return 0;
    if (!feats || !vox_off || !vox_pts || !F0) return DD3D_EARG;
    k_voxel_mean_fwd<<<dim3((unsigned)ceil_div64(M * C, 256)), 256, 0, (hipStream_t)stream>>>(feats, C, vox_off, vox_pts, M, m_dev, F0);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_voxel_mean_bwd(const float* dF0, int64_t M, int C, const int32_t* inv, const int32_t* vox_off, int64_t N, float* dfeats,
                                 void* stream) {
    if (N < 0 || M < 0 || C <= 0) return DD3D_EARG;
    if (N == 0) return 0;
    if (!dF0 || !inv || !vox_off || !dfeats) return DD3D_EARG;
    k_voxel_mean_bwd<<<dim3((unsigned)ceil_div64(N * C, 256)), 256, 0, (hipStream_t)stream>>>(dF0, C, inv, vox_off, N, dfeats);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_inorm_fwd(const float* Y, const int32_t* seg_off, int B, int C, float* xhat, float* inv_std, void* stream) {
    if (B < 0 || C != 128) return DD3D_EARG;
    if (B == 0) return 0;
    if (!Y || !seg_off || !xhat || !inv_std) return DD3D_EARG;
    k_inorm_fwd<16, 8><<<dim3(B), 512, 0, (hipStream_t)stream>>>((const f32x4*)Y, seg_off, (f32x4*)xhat, (f32x4*)inv_std);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_inorm_bwd(const float* g, const float* xhat, const float* inv_std, const int32_t* seg_off, int B, int C, float* dY, void* stream) {
    if (B < 0 || C != 128) return DD3D_EARG;
    if (B == 0) return 0;
    if (!g || !xhat || !inv_std || !seg_off || !dY) return DD3D_EARG;
    k_inorm_bwd<16, 8><<<dim3(B), 512, 0, (hipStream_t)stream>>>((const f32x4*)g, (const f32x4*)xhat, (const f32x4*)inv_std, seg_off, (f32x4*)dY);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// out[M_out,128], arg[M_out,128] (child row that won).  w/b may be null with use_affine_relu = 0 (plain max-pool).
DD3D_API int dd3d_affine_relu_pool_fwd(const float* xhat, const float* w, const float* b, int use_affine_relu, int C, const int32_t* kids, int64_t ld,
                                       int Kp, int64_t M_out, const int32_t* m_dev, float* out, int32_t* arg, void* stream) {
    if (M_out < 0 || C != 128 || Kp <= 0 || Kp > 8 || ld < M_out || (use_affine_relu && (!w || !b))) return DD3D_EARG;
    if (M_out == 0) return 0;
    if (!xhat || !kids || !out || !arg) return DD3D_EARG;
    k_arp_fwd<<<dim3((unsigned)ceil_div64(M_out, 8)), 256, 0, (hipStream_t)stream>>>((const f32x4*)xhat, (const f32x4*)w, (const f32x4*)b, use_affine_relu,
                                                                                    kids, ld, Kp, M_out, m_dev, (f32x4*)out, (int4*)arg);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_inorm_fwd followed by dd3d_affine_relu_pool_fwd (affine + ReLU on) in one launch: same xhat / inv_std / pooled / arg, bit for bit.
// seg_out[B+1]: the clouds' row ranges at the pooled level (a cloud's pooled rows only have children in that cloud).
DD3D_API int dd3d_inorm_arp_fwd(const float* Y, const int32_t* seg_off, int B, int C, float* xhat, float* inv_std, const float* w, const float* b,
                                const int32_t* kids, int64_t ld, int Kp, const int32_t* seg_out, float* pooled, int32_t* arg, void* stream) {
    if (B < 0 || C != 128 || Kp <= 0 || Kp > 8) return DD3D_EARG;
    if (B == 0) return 0;
    if (!Y || !seg_off || !xhat || !inv_std || !w || !b || !kids || !seg_out || !pooled || !arg) return DD3D_EARG;
    k_inorm_arp_fwd<16, 8><<<dim3(B), 512, 0, (hipStream_t)stream>>>((const f32x4*)Y, seg_off, (f32x4*)xhat, (f32x4*)inv_std, (const f32x4*)w, (const f32x4*)b,
                                                                     kids, ld, Kp, seg_out, (f32x4*)pooled, (int4*)arg);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_inorm_arp_fwd WITHOUT the normalised rows in memory: mean[B,128] and inv_std[B,128] are all a later pass needs next to Y itself
// (dd3d_arp_inorm_bwd_y normalises a row as it reads it).  pooled / arg / inv_std: the same bits as dd3d_inorm_arp_fwd.
DD3D_API int dd3d_inorm_arp_fwd_y(const float* Y, const int32_t* seg_off, int B, int C, float* mean, float* inv_std, const float* w, const float* b,
                                  const int32_t* kids, int64_t ld, int Kp, const int32_t* seg_out, float* pooled, int32_t* arg, void* stream) {
    if (B < 0 || C != 128 || Kp <= 0 || Kp > 8) return DD3D_EARG;
    if (B == 0) return 0;
    if (!Y || !seg_off || !mean || !inv_std || !w || !b || !kids || !seg_out || !pooled || !arg) return DD3D_EARG;
    static const int vc = getenv("DD3D_NORM_VC") ? atoi(getenv("DD3D_NORM_VC")) : 6;  // rows per virtual row-lane kept in registers (measurements)
    if (vc <= 4)
        k_inorm_arp_fwd_y<4><<<dim3(B), 256, 0, (hipStream_t)stream>>>((const f32x4*)Y, seg_off, (f32x4*)mean, (f32x4*)inv_std, (const f32x4*)w, (const f32x4*)b,
                                                                       kids, ld, Kp, seg_out, (f32x4*)pooled, (int4*)arg);
    else if (vc <= 6)
        k_inorm_arp_fwd_y<6><<<dim3(B), 256, 0, (hipStream_t)stream>>>((const f32x4*)Y, seg_off, (f32x4*)mean, (f32x4*)inv_std, (const f32x4*)w, (const f32x4*)b,
                                                                       kids, ld, Kp, seg_out, (f32x4*)pooled, (int4*)arg);
    else
        k_inorm_arp_fwd_y<8><<<dim3(B), 256, 0, (hipStream_t)stream>>>((const f32x4*)Y, seg_off, (f32x4*)mean, (f32x4*)inv_std, (const f32x4*)w, (const f32x4*)b,
                                                                       kids, ld, Kp, seg_out, (f32x4*)pooled, (int4*)arg);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int64_t dd3d_arp_bwd_ws_bytes(int64_t M_in, int G) {
    int64_t slots = arp_max_slots(M_in, G);
    return align_up64(slots * 2 * 128 * 4, 256) + align_up64(slots * 4, 256);
}

// g_xhat[M_in,128]; gw_w/gw_b [G,128] (may be null); ga_out (may be null) saves relu-masked pooled gradient;
// ga_prev + gw_extra[G,128] (both or neither) add ga_prev * gw_extra[group] to g_xhat.
DD3D_API int dd3d_affine_relu_pool_bwd(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* w, const float* b,
                                       int use_affine_relu, int C, int64_t M_in, const int32_t* group_row_off, int G, float* g_xhat, float* ga_out,
                                       const float* ga_prev, const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream) {
    if (M_in < 0 || C != 128 || G <= 0 || !group_row_off || (use_affine_relu && (!w || !b)) || ((ga_prev == nullptr) != (gw_extra == nullptr)) ||
        ((gw_w == nullptr) != (gw_b == nullptr)))
        return DD3D_EARG;
    if (ws_bytes < dd3d_arp_bwd_ws_bytes(M_in, G) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    int64_t slots = arp_max_slots(M_in, G);
    f32x4* partial = (f32x4*)ws;
    int* valid = (int*)((char*)ws + align_up64(slots * 2 * 128 * 4, 256));
    DD3D_HIP(hipMemsetAsync(valid, 0, slots * 4, st));
    if (M_in > 0) {
        if (!g_pool || !arg || !parent || !xhat || !g_xhat) return DD3D_EARG;
        k_arp_bwd<<<dim3((unsigned)slots), 256, 0, st>>>((const f32x4*)g_pool, (const int4*)arg, parent, (const f32x4*)xhat, (const f32x4*)w,
                                                         (const f32x4*)b, use_affine_relu, group_row_off, G, (f32x4*)g_xhat, (f32x4*)ga_out,
                                                         (const f32x4*)ga_prev, (const f32x4*)gw_extra, gw_w ? partial : nullptr, valid);
    }
    if (gw_w) k_arp_reduce<<<dim3(G), 256, 0, st>>>(partial, group_row_off, G, (f32x4*)gw_w, (f32x4*)gw_b);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_scale_rows_by_group(const float* ga, const float* gw, const int32_t* group_row_off, int G, int64_t M, int C, float* out, void* stream) {
    if (M < 0 || C != 128 || G <= 0 || !group_row_off) return DD3D_EARG;
    if (M == 0) return 0;
    if (!ga || !gw || !out) return DD3D_EARG;
    k_scale_rows_by_group<<<dim3((unsigned)ceil_div64(M, 8)), 256, 0, (hipStream_t)stream>>>((const f32x4*)ga, (const f32x4*)gw, group_row_off, G, M,
                                                                                            (f32x4*)out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int64_t dd3d_arp_inorm_bwd_ws_bytes(int B) { return align_up64((int64_t)(B > 0 ? B : 1) * 2 * 128 * 4, 256); }

// dY[M_in,128] = backward of (instance norm -> affine -> ReLU -> max pool) w.r.t. the norm's input, per cloud, in one pass.
// gw_w/gw_b[G,128] (nullable pair): per-group affine gradients; ga_out (nullable): ReLU-masked pooled gradient;
// ga_prev + gw_extra[G,128] (nullable pair): adds ga_prev * gw_extra[group] to the gradient w.r.t. xhat first.
static int arp_inorm_bwd_launch(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* mean, const float* inv_std,
                                const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                                const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                                const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream) {
    if (B < 0 || C != 128 || G <= 0 || !w || !b || ((ga_prev == nullptr) != (gw_extra == nullptr)) || ((gw_w == nullptr) != (gw_b == nullptr)) ||
        (gw_w && !group_cloud_off) || (G > 1 && gw_extra && !cloud_group))
        return DD3D_EARG;
    if (gw_w && (ws_bytes < dd3d_arp_inorm_bwd_ws_bytes(B) || !ws)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    // (a function of the cloud count alone, which a bound-sized launch and an exact one share: the same mode per cloud, the same sums)
    static const int wave_env = getenv("DD3D_ARP_WAVE_ROWS") ? atoi(getenv("DD3D_ARP_WAVE_ROWS")) : -1;  // 0: every cloud in workgroup mode (measurements)
    const int wave_rows = wave_env >= 0 ? wave_env : (B < ARP_FEW_CLOUDS ? ARP_WAVE_ROWS_FEW : ARP_WAVE_ROWS);
    if (B > 0) {
        if (!g_pool || !arg || !parent || !xhat || !inv_std || !seg_off || !dY) return DD3D_EARG;
        // 32 row-lanes x 8 float4 channel-lanes per workgroup, four workgroups per cloud (measured against 512- and 1024-thread shapes:
        // more, smaller workgroups per CU overlap their gather chains best)
        k_arp_inorm_bwd<32, 4, 8><<<dim3(B, 4), 256, 0, st>>>((const f32x4*)g_pool, (const int4*)arg, parent, (const f32x4*)xhat, (const f32x4*)inv_std,
                                                       (const f32x4*)w, (const f32x4*)b, seg_off, cloud_group, (f32x4*)dY, (f32x4*)ga_out, ga_clouds,
                                                       (const f32x4*)ga_prev, (const f32x4*)gw_extra, gw_w ? (f32x4*)ws : nullptr, wave_rows,
                                                       (const f32x4*)mean);
    }
    if (gw_w) k_cloud_group_sum<<<dim3(G, 4), 256, 0, st>>>((const f32x4*)ws, group_cloud_off, (f32x4*)gw_w, (f32x4*)gw_b);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_arp_inorm_bwd(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* inv_std,
                                const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                                const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                                const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream) {
    return arp_inorm_bwd_launch(g_pool, arg, parent, xhat, nullptr, inv_std, w, b, C, seg_off, B, cloud_group, group_cloud_off, G, dY, ga_out, ga_clouds,
                                ga_prev, gw_extra, gw_w, gw_b, ws, ws_bytes, stream);
}

// dd3d_arp_inorm_bwd over the norm's INPUT rows Y and the per-cloud mean / inv_std of dd3d_inorm_arp_fwd_y instead of stored xhat rows:
// every output has the same bits.
DD3D_API int dd3d_arp_inorm_bwd_y(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* Y, const float* mean, const float* inv_std,
                                  const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                                  const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                                  const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream) {
    if (B > 0 && !mean) return DD3D_EARG;
    return arp_inorm_bwd_launch(g_pool, arg, parent, Y, mean, inv_std, w, b, C, seg_off, B, cloud_group, group_cloud_off, G, dY, ga_out, ga_clouds,
                                ga_prev, gw_extra, gw_w, gw_b, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// dd3d_arp_inorm_bwd for callers that cannot allocate (code under HIP-graph capture with a frozen memory pool, tight serving loops,
// harnesses that forbid hipMalloc): the per-cloud partial sums part[B][2][32] float4 live in a file-scope device array instead of a
// caller workspace.  The array's NAME is a host-side stand-in, not a device address: its address on each device is taken once with
// hipGetSymbolAddress (with that device current) and kept per device; host code only ever uses that pointer.
// ---------------------------------------------------------------------------------------------
#define DD3D_ARP_STATIC_MAX_CLOUDS 8192
static __device__ __align__(256) float scratch[(size_t)DD3D_ARP_STATIC_MAX_CLOUDS * 2 * 128];  // 8 MiB per GPU, held until the process ends

#define DD3D_MAX_DEVICES 64
static void* g_scratch_ptr[DD3D_MAX_DEVICES];  // device address of `scratch` on every GPU this process has called from (0 = not taken yet)

DD3D_API int dd3d_arp_inorm_bwd_static_max_clouds(void) { return DD3D_ARP_STATIC_MAX_CLOUDS; }

DD3D_API int dd3d_arp_inorm_bwd_static(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* inv_std,
                                       const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                                       const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                                       const float* gw_extra, float* gw_w, float* gw_b, void* stream) {
    if (B > DD3D_ARP_STATIC_MAX_CLOUDS) return DD3D_ESIZE;  // more clouds than the static scratch holds: use dd3d_arp_inorm_bwd with a workspace
    int dev = -1;
    DD3D_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= DD3D_MAX_DEVICES) return DD3D_ESIZE;
    void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {  // first call on this device (it is the current one): no allocation, the array is part of the loaded code object
        DD3D_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return (int)hipErrorInvalidSymbol;
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);  // racing first calls store the same address
    }
    // every slot part[b], b < B, is written by the kernel before k_cloud_group_sum reads it (empty clouds write zeros): nothing to clear
    return dd3d_arp_inorm_bwd(g_pool, arg, parent, xhat, inv_std, w, b, C, seg_off, B, cloud_group, group_cloud_off, G, dY, ga_out, ga_clouds, ga_prev,
                              gw_extra, gw_w, gw_b, p, (int64_t)sizeof(float) * DD3D_ARP_STATIC_MAX_CLOUDS * 2 * 128, stream);
}
