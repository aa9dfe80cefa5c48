// Rasteriser of the evaluation block: synthetic point clouds -> 28x28 occupancy images (mnist_distill.py:163-168 populate_img, called
// at :179 before the ConvNet is trained on the distilled set).  One thread per point: img[cloud, 0, trunc(p[1]), trunc(p[2])] = 1.
// The reference indexes through `pt.type(torch.LongTensor)`: truncation toward zero, and a negative index counts from the end as in
// Python; an index outside [-H, H) x [-W, W) is Python's IndexError -> bit 0 of *err.  Several points of a cloud may hit one pixel:
// they all store the same 1.0f, so the plain stores need no ordering.  Bandwidth-trivial (12 B read per point).
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void k_populate_img(const float* __restrict__ xyz, int64_t total, int64_t pts, int H, int W, float* __restrict__ img,
                                                      int* __restrict__ err) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int y = (int)xyz[t * 3 + 1], x = (int)xyz[t * 3 + 2];  // float -> int conversion truncates toward zero, like .type(LongTensor)
    if (y < 0) y += H;
    if (x < 0) x += W;
    if (y < 0 || y >= H || x < 0 || x >= W) {
        atomicOr(err, 1);
        return;
    }
    img[((t / pts) * H + y) * W + x] = 1.f;
}
}  // namespace

DD3D_API int dd3d_populate_img(const float* xyz, int64_t n_clouds, int64_t pts_per_cloud, int H, int W, float* img, int32_t* err, void* stream) {
    if (n_clouds < 0 || pts_per_cloud < 0 || H <= 0 || W <= 0) return DD3D_EARG;
    const int64_t total = n_clouds * pts_per_cloud;
    if (total == 0) return 0;
    if (!xyz || !img || !err) return DD3D_EARG;
    if (total >= (1ll << 31) * 256) return DD3D_ESIZE;
    k_populate_img<<<dim3((unsigned)ceil_div64(total, 256)), 256, 0, (hipStream_t)stream>>>(xyz, total, pts_per_cloud, H, W, img, err);
    DD3D_CHECK_LAUNCH();
    return 0;
}
