// Average pooling of the reference's UNUSED model variants (models/MEConv.py:20-39,97-130: MEConv / MEConvImage build
// MinkowskiAvgPooling(kernel_size=2, stride=2) after every block and, with full_minkowski=True, MinkowskiGlobalAvgPooling before a
// MinkowskiLinear).  Not on the matching step's hot path (MEConvExp pools with the maximum); bandwidth-bound one-pass kernels.
//   local : out[v] = mean of the children of v that exist (MinkowskiEngine's average pooling divides by the number of inputs
//           actually mapped to the output, not by the kernel volume); every input row has exactly one parent, so the backward is a gather
//   global: out[b] = mean over the rows of cloud b
// Every index read from a map is range-checked before it becomes an address.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_avg_pool_fwd(const f32x4* __restrict__ X, const int* __restrict__ kids, int64_t ld, int K, int M_in, int M_bound,
                                                      const int* __restrict__ m_dev, int C4, f32x4* __restrict__ out, int* __restrict__ cnt) {
    const int M_out = m_dev ? min(*m_dev, M_bound) : M_bound;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int v = (int)(idx / C4), c4 = (int)(idx - (int64_t)v * C4);
    if (v >= M_out) return;
    int kid[8];
    f32x4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) kid[k] = k < K ? kids[(int64_t)k * ld + v] : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // the eight gathers go out together (clamped address, masked afterwards)
        const bool ok = (unsigned)kid[k] < (unsigned)M_in;
        x[k] = X[(int64_t)(ok ? kid[k] : 0) * C4 + c4];
        if (!ok) kid[k] = -1;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)  // offset order: deterministic
        if (kid[k] >= 0) {
            s += x[k];
            ++n;
        }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    out[(int64_t)v * C4 + c4] = n ? s / (float)n : zero;
    if (c4 == 0) cnt[v] = n;
}

__global__ __launch_bounds__(256) void k_avg_pool_bwd(const f32x4* __restrict__ g, const int* __restrict__ cnt, const int* __restrict__ parent, int M_in,
                                                      int M_out, int C4, f32x4* __restrict__ dX) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(idx / C4), c4 = (int)(idx - (int64_t)i * C4);
    if (i >= M_in) return;
    const int p = parent[i];
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)p < (unsigned)M_out) {
        const int n = cnt[p];
        if (n > 0) r = g[(int64_t)p * C4 + c4] / (float)n;
    }
    dX[(int64_t)i * C4 + c4] = r;
}

// one workgroup per cloud: 256 / C4 row-lanes walk the cloud's rows, folded over the row-lanes in lane order (deterministic)
__global__ __launch_bounds__(256) void k_global_avg_pool_fwd(const f32x4* __restrict__ X, const int* __restrict__ seg_off, int C4, int64_t M,
                                                             f32x4* __restrict__ out) {
    __shared__ f32x4 s_red[256];
    const int b = blockIdx.x, c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
    const int64_t r0 = max(seg_off[b], 0), r1 = min((int64_t)seg_off[b + 1], M);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0 + rl; r < r1; r += RL) s += X[r * C4 + c4];
    s_red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        f32x4 t = s_red[c4];
        for (int q = 1; q < RL; ++q) t += s_red[q * C4 + c4];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        out[(int64_t)b * C4 + c4] = r1 > r0 ? t / (float)(r1 - r0) : zero;
    }
}

__global__ __launch_bounds__(256) void k_global_avg_pool_bwd(const f32x4* __restrict__ g, const int* __restrict__ seg_off, int C4, int64_t M,
                                                             f32x4* __restrict__ dX) {
    const int b = blockIdx.x, c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
    const int64_t r0 = max(seg_off[b], 0), r1 = min((int64_t)seg_off[b + 1], M);
    if (r1 <= r0) return;
    const f32x4 v = g[(int64_t)b * C4 + c4] / (float)(r1 - r0);
    for (int64_t r = r0 + rl; r < r1; r += RL) dX[r * C4 + c4] = v;
}

inline bool channels_ok(int C) { return C > 0 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0; }

}  // namespace

// See include/dd3d.h.
DD3D_API int dd3d_avg_pool_fwd(const float* X, int64_t M_in, int C, const int32_t* kids, int64_t ld, int K, int64_t M_out, const int32_t* m_dev, float* out,
                               int32_t* cnt, void* stream) {
    if (M_in < 0 || M_out < 0 || !channels_ok(C) || K <= 0 || K > 8 || ld < M_out || M_in >= (1ll << 31) || M_out >= (1ll << 31)) return DD3D_EARG;
    if (M_out == 0) return 0;
    if (!X || !kids || !out || !cnt) return DD3D_EARG;
    const int C4 = C / 4;
    k_avg_pool_fwd<<<dim3((unsigned)ceil_div64(M_out * C4, 256)), 256, 0, (hipStream_t)stream>>>((const f32x4*)X, kids, ld, K, (int)M_in, (int)M_out, m_dev, C4,
                                                                                                (f32x4*)out, cnt);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_avg_pool_bwd(const float* g, const int32_t* cnt, const int32_t* parent, int64_t M_in, int64_t M_out, int C, float* dX, void* stream) {
    if (M_in < 0 || M_out < 0 || !channels_ok(C) || M_in >= (1ll << 31) || M_out >= (1ll << 31)) return DD3D_EARG;
    if (M_in == 0) return 0;
    if (!g || !cnt || !parent || !dX) return DD3D_EARG;
    const int C4 = C / 4;
    k_avg_pool_bwd<<<dim3((unsigned)ceil_div64(M_in * C4, 256)), 256, 0, (hipStream_t)stream>>>((const f32x4*)g, cnt, parent, (int)M_in, (int)M_out, C4,
                                                                                               (f32x4*)dX);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_global_avg_pool_fwd(const float* X, int64_t M, const int32_t* seg_off, int B, int C, float* out, void* stream) {
    if (B < 0 || M < 0 || !channels_ok(C)) return DD3D_EARG;
    if (B == 0) return 0;
    if (!X || !seg_off || !out) return DD3D_EARG;
    k_global_avg_pool_fwd<<<dim3((unsigned)B), 256, 0, (hipStream_t)stream>>>((const f32x4*)X, seg_off, C / 4, M, (f32x4*)out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_global_avg_pool_bwd(const float* g, int64_t M, const int32_t* seg_off, int B, int C, float* dX, void* stream) {
    if (B < 0 || M < 0 || !channels_ok(C)) return DD3D_EARG;
    if (B == 0 || M == 0) return 0;
    if (!g || !seg_off || !dX) return DD3D_EARG;
    k_global_avg_pool_bwd<<<dim3((unsigned)B), 256, 0, (hipStream_t)stream>>>((const f32x4*)g, seg_off, C / 4, M, (f32x4*)dX);
    DD3D_CHECK_LAUNCH();
    return 0;
}
