/* libdd3d_cpu.so -- the C++/OpenMP CPU leg of the oracle (SURVEY.md section 8b "mirrored by libdd3d_cpu.so", 8d(ii)).
 * TEST INFRASTRUCTURE AND CPU BASELINE ONLY: nothing under dd3d_amd/ loads it.  Host pointers, no streams, no workspaces.
 * Every function returns 0, -1 (bad arguments), -2 (a voxel outside the dense extent: MinkowskiEngine's dense() raises there) or
 * -3 (a coordinate beyond the 16 bits per axis of this leg's packed voxel keys).  _f64 for checking, _f32 for the timed baseline.
 * Array layouts are those of include/dd3d.h, so a test hands the arrays of a HIP call over unchanged:
 *   coordinate rows int32 [M][4] = (cloud, c1, c2, c3), first-occurrence order, clouds ascending;  seg_off int32 [B+1];
 *   neighbour tables nbr[27][ld], k = (d1+1) + 3 (d2+1) + 9 (d3+1), -1 = absent;  children tables kids[8][ld], k = e1 + 2 e2 + 4 e3;
 *   features row-major [M][C];  kernels [27][Cin][Cout];  gradient lists flat in parameters() order (models/MEConv.py:221-230). */
#ifndef DD3D_CPU_H
#define DD3D_CPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dd3d_cpu_net {
    int J, E0, E1, E2;                                  /* classifier outputs, dense extent (models/MEConv.py:215,250) */
    const void *W0, *w1, *b1, *W1, *w5, *b5, *Wc, *bc; /* the eight parameters of MEConvExp, in the arithmetic type of the call */
} dd3d_cpu_net;

/* ---- one class of one matching step (mnist_distill.py:336-365); the timed baseline of bench.py ---- */
int64_t dd3d_cpu_num_params(const dd3d_cpu_net* n);
void dd3d_cpu_set_threads(int t);
int dd3d_cpu_max_threads(void);
void dd3d_cpu_prof_dump(void);
int dd3d_cpu_class_step_f64(const dd3d_cpu_net* n, const double* rcoords, const double* rfeats, const int* roff, int Br, const double* scoords,
                            const double* sfeats, const int* soff, int Bs, int label, int metric, double* loss, double* grad, double* gw_real,
                            double* gw_syn);
int dd3d_cpu_class_step_f32(const dd3d_cpu_net* n, const double* rcoords, const float* rfeats, const int* roff, int Br, const double* scoords,
                            const float* sfeats, const int* soff, int Bs, int label, int metric, float* loss, float* grad, float* gw_real,
                            float* gw_syn);

/* ---- coordinate maps (int32, bit-exact); mirrors of the libdd3d_hip.so entry points named on the right ---- */
/* coords [N][3]: spatial coordinates already divided by the voxel size, as doubles; Cq / inv sized for N rows */
int dd3d_cpu_quantize_unique(const double* coords, const int32_t* pt_off, int B, int32_t* Cq, int32_t* inv, int32_t* seg_off,
                             int64_t* M);                                                   /* dd3d_floor_coords + dd3d_quantize_unique + dd3d_segment_offsets */
int dd3d_cpu_stride_coords(const int32_t* Cin, int64_t M_in, int B, int stride, int32_t* Cout, int32_t* parent, int32_t* seg_off,
                           int64_t* M_out);                                                 /* dd3d_stride_coords + dd3d_unique_rows */
int dd3d_cpu_kernel_map(const int32_t* Cq, int64_t M, int tensor_stride, int32_t* nbr /* [27][M] */);   /* dd3d_kernel_map, kernel 3, stride 1 */
int dd3d_cpu_pool_map(const int32_t* Cin, int64_t M_in, const int32_t* Cout, int64_t M_out, int tensor_stride_in,
                      int32_t* kids /* [8][M_out] */);                                      /* dd3d_kernel_map, kernel 2, stride 2 */
int dd3d_cpu_dense_sites(const int32_t* Cq, int64_t M, int tensor_stride, int E0, int E1, int E2, int32_t* site);   /* dd3d_dense_sites, one group */

/* ---- feature operators (T = double / float) ---- */
#define DD3D_CPU_DECL(T, SFX)                                                                                                                      \
    int dd3d_cpu_voxel_mean_fwd_##SFX(const T* feats, int64_t N, int Cf, const int32_t* inv, int64_t M, T* F0);           /* dd3d_voxel_mean_fwd */ \
    int dd3d_cpu_voxel_mean_bwd_##SFX(const T* dF0, int64_t M, int Cf, const int32_t* inv, int64_t N, T* dfeats);         /* dd3d_voxel_mean_bwd */ \
    int dd3d_cpu_conv_fwd_##SFX(const T* X, int Cin, const T* W, int Cout, const int32_t* nbr, int64_t ld, int64_t M, T* Y);        /* dd3d_conv_fwd */ \
    int dd3d_cpu_conv_bwd_data_##SFX(const T* dY, int Cout, const T* W, int Cin, const int32_t* nbr, int64_t ld, int64_t M, T* dX); /* dd3d_conv_bwd_data */ \
    int dd3d_cpu_conv_bwd_weight_##SFX(const T* X, int Cin, const T* dY, int Cout, const int32_t* nbr, int64_t ld, int64_t M, T* dW); /* dd3d_conv_bwd_weight, one group */ \
    int dd3d_cpu_inorm_fwd_##SFX(const T* Y, const int32_t* seg_off, int B, int C, T* xhat, T* inv_std);                  /* dd3d_inorm_fwd */      \
    int dd3d_cpu_inorm_bwd_##SFX(const T* g, const T* xhat, const T* inv_std, const int32_t* seg_off, int B, int C, T* dY); /* dd3d_inorm_bwd */    \
    int dd3d_cpu_affine_relu_pool_fwd_##SFX(const T* xhat, const T* w, const T* b, int C, const int32_t* kids, int64_t ld, int64_t M_out, T* out,  \
                                            int32_t* arg);                                                                /* dd3d_affine_relu_pool_fwd */ \
    int dd3d_cpu_affine_relu_pool_bwd_##SFX(const T* g_pool, const int32_t* arg, const int32_t* parent, const T* xhat, const T* w, const T* b,     \
                                            int C, int64_t M_in, T* ga, T* g_xhat);                                       /* dd3d_affine_relu_pool_bwd */ \
    int dd3d_cpu_sparse_linear_fwd_##SFX(const T* F, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int C, int S, const T* Wc,     \
                                         const T* bias, int J, T* logits);                                                /* dd3d_sparse_linear_fwd */ \
    int dd3d_cpu_sparse_linear_bwd_x_##SFX(const T* dl, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int C, int S, const T* Wc,  \
                                           int J, T* dF);                                                                 /* dd3d_sparse_linear_bwd_x */ \
    int dd3d_cpu_sparse_linear_bwd_w_##SFX(const T* dl, const T* F, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int C, int S,   \
                                           int J, T* gW);                                                                 /* dd3d_sparse_linear_bwd_w, one group */ \
    int dd3d_cpu_softmax_ce_##SFX(const T* logits, const int32_t* labels, int B, int J, T* loss, T* dlogits);             /* dd3d_softmax_ce */     \
    /* distillation_loss.py:83-130 over the eight MEConvExp-shaped tensors (settings.Pyt False); metric 0 ours, 1 modmse, 2 mse, 3 cos;         \
     * d_gw_syn (nullable): the gradient with respect to all eight entries of gw_syn */                                                             \
    int dd3d_cpu_match_loss_##SFX(const T* gw_syn, const T* gw_real, int J, int S, int metric, T* loss, T* d_gw_syn);     /* dd3d_match_loss_fwd + _bwd */
DD3D_CPU_DECL(double, f64)
DD3D_CPU_DECL(float, f32)

#ifdef __cplusplus
}
#endif
#endif
