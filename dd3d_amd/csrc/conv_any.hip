// Sparse convolution for ANY channel widths up to 128 and any number of kernel offsets -- the shapes that only the reference's UNUSED
// model variants name (models/MEConv.py:25-39,102-130: 1 -> 128, 128 -> 64 -> 32 -> 16 at kernel size 3, 16 -> 1 and 128 -> 128 at kernel
// size 1) and whatever a research variant of the network asks for.  The hot-path shapes ({3,128} -> 128, K = 27) keep their own kernels
// (conv.hip, conv_bww_sk.hip); these are their general-purpose siblings: exact-fp32 MFMA 32x32x2 on zero-padded 32-wide channel blocks,
// every wave on its own (no workgroup barrier anywhere), deterministic (fixed summation orders, no float atomics).
//
//   forward / input gradient (k_conv_any): a wave owns 32 output rows and all <= 4 blocks of 32 output channels; per offset with a
//     neighbour it stages the 32 gathered input rows in its own LDS region (coalesced along the channels) and contracts them with the
//     offset's [CA x CB] weight slice read through the caches.  The input gradient of a stride-1 convolution is the same launch with the
//     offsets mirrored (k <-> K-1-k: the offsets of an odd kernel are centrally symmetric) and the weight slice read transposed.
//   kernel gradient (k_conv_any_bww + k_conv_any_reduce): dW[k] = sum_o X[nbr[k][o]]^T dY[o].  The rows are cut into ANY_NCH chunks (a
//     function of the TRUE row count, so a bound-sized launch sums in the same order); a wave owns one (chunk, offset, 32-channel block of
//     X) and all blocks of dY's channels, two rows per MFMA; the chunk partials are added in chunk order.
#include "common.h"

namespace {

constexpr int ANY_MAXC = 128;  // widest channel count (four 32-wide blocks of accumulators per wave)
constexpr int ANY_NCH = 32;    // row chunks of the kernel gradient

__device__ __forceinline__ void wave_lds_sync() {  // LDS is in order per wave: wait for the wave's own writes, and keep the compiler from moving accesses across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Y[o][b] = sum_k sum_a X[nbr[ks][o]][a] * W[k][a * wa + b * wb],  ks = flip ? K-1-k : k;  nbr == null: every row is its own only neighbour (K = 1)
__global__ __launch_bounds__(256) void k_conv_any(const float* __restrict__ X, int CA, const float* __restrict__ W, int CB, int wa, int wb, int flip,
                                                  const int* __restrict__ nbr, int64_t ld, int K, int64_t M_bound, const int* __restrict__ m_dev,
                                                  float* __restrict__ Y) {
    extern __shared__ float s_a[];  // [4 waves][32 rows][LDA]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i32 = lane & 31, h = lane >> 5;
    const int64_t M = m_dev ? min((int64_t)*m_dev, M_bound) : M_bound;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * 32;
    if (row0 >= M) return;  // wave-uniform, and the kernel has no workgroup barrier
    const int CAp = (CA + 1) & ~1, LDA = CAp + 1;  // contraction in pairs; an odd row pitch keeps the 32 rows on 32 banks
    float* As = s_a + (size_t)w * 32 * LDA;
    const int nblk = (CB + 31) >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int64_t myrow = row0 + i32;
    for (int k = 0; k < K; ++k) {
        const int ks = flip ? K - 1 - k : k;
        int nb = -1;
        if (myrow < M) nb = nbr ? nbr[(int64_t)ks * ld + myrow] : (int)myrow;
        if (!__ballot(nb >= 0)) continue;  // none of the wave's rows has this neighbour
        {   // stage the 32 gathered rows, coalesced along the channels: element idx = r * CAp + c, 64 elements per pass (no division in the loop)
            int r = lane / CAp, c = lane - r * CAp;
            const int dr = 64 / CAp, dc = 64 - dr * CAp;
            for (int idx = lane; idx < 32 * CAp; idx += 64) {
                const int src = __shfl(nb, r, 64);
                As[r * LDA + c] = (src >= 0 && c < CA) ? X[(int64_t)src * CA + c] : 0.f;
                r += dr;
                c += dc;
                if (c >= CAp) {
                    c -= CAp;
                    ++r;
                }
            }
        }
        wave_lds_sync();
        const float* Wk = W + (int64_t)k * CA * CB;
        for (int s0 = 0; s0 < CAp / 2; s0 += 4) {  // four k-pairs at a time: their weight loads are in flight together, then 4 x nblk MFMAs
            float av[4], bv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = min(2 * (s0 + q) + h, CAp - 1);
                av[q] = (2 * (s0 + q) < CAp) ? As[i32 * LDA + a] : 0.f;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n >= nblk) break;
                    const int b = n * 32 + i32;
                    const bool ok = 2 * (s0 + q) + h < CA && b < CB;
                    bv[q][n] = ok ? Wk[(int64_t)(2 * (s0 + q) + h) * wa + (int64_t)b * wb] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (2 * (s0 + q) >= CAp) break;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n >= nblk) break;
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q][n], acc[n], 0, 0, 0);
                }
            }
        }
        wave_lds_sync();  // the next offset overwrites the rows just read
    }
    // C/D layout of a 32x32 block: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (n >= nblk) break;
        const int b = n * 32 + i32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M && b < CB) Y[row * CB + b] = acc[n][r];
        }
    }
}

// part[c][k][a][b] = sum over the rows o of chunk c of X[nbr[k][o]][a] * dY[o][b]
__global__ __launch_bounds__(256) void k_conv_any_bww(const float* __restrict__ X, int CA, const float* __restrict__ dY, int CB, const int* __restrict__ nbr,
                                                      int64_t ld, int K, int64_t M_bound, const int* __restrict__ m_dev, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i32 = lane & 31, h = lane >> 5;
    const int64_t M = m_dev ? min((int64_t)*m_dev, M_bound) : M_bound;
    const int nA = (CA + 31) >> 5, nblk = (CB + 31) >> 5, k = blockIdx.y;
    const int u = blockIdx.x * 4 + w;  // (chunk, block of X's channels)
    if (u >= ANY_NCH * nA) return;
    const int c = u / nA, ia = u - c * nA;
    const int64_t chunk = ((M + ANY_NCH - 1) / ANY_NCH + 1) & ~(int64_t)1;  // even: a pair of rows never straddles two chunks
    const int64_t o0 = min((int64_t)c * chunk, M), o1 = min(o0 + chunk, M);
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int a = ia * 32 + i32;
    for (int64_t ob = o0; ob < o1; ob += 128) {  // 128 rows per pass: their table entries in two coalesced loads, then eight row pairs at a time
        int nbv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t oo = ob + q * 64 + lane;
            nbv[q] = -1;
            if (oo < o1) nbv[q] = nbr ? nbr[(int64_t)k * ld + oo] : (int)oo;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!__ballot(nbv[q] >= 0)) continue;
            for (int j0 = 0; j0 < 32; j0 += 8) {  // pair j of this half: rows ob + 64 q + 2 j + h
                float av[8], bv[8][4];
                int srcs[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    srcs[j] = __shfl(nbv[q], 2 * (j0 + j) + h, 64);
                    const int64_t oo = ob + q * 64 + 2 * (j0 + j) + h;
                    av[j] = (srcs[j] >= 0 && a < CA) ? X[(int64_t)srcs[j] * CA + a] : 0.f;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (n >= nblk) break;
                        const int b = n * 32 + i32;
                        bv[j][n] = (srcs[j] >= 0 && b < CB) ? dY[oo * CB + b] : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (!__ballot(srcs[j] >= 0)) continue;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (n >= nblk) break;
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j][n], acc[n], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* out = part + ((int64_t)c * K + k) * CA * CB;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (n >= nblk) break;
        const int b = n * 32 + i32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ar = ia * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ar < CA && b < CB) out[(int64_t)ar * CB + b] = acc[n][r];
        }
    }
}

__global__ __launch_bounds__(256) void k_conv_any_reduce(const float* __restrict__ part, int64_t elems, float* __restrict__ dW) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float v[ANY_NCH];
#pragma unroll
    for (int c = 0; c < ANY_NCH; ++c) v[c] = part[(int64_t)c * elems + e];  // all loads in flight, added in chunk order
    float t = v[0];
#pragma unroll
    for (int c = 1; c < ANY_NCH; ++c) t += v[c];
    dW[e] = t;
}

int launch_conv_any(const float* X, int CA, const float* W, int CB, int wa, int wb, int flip, const int32_t* nbr, int64_t ld, int K, int64_t M,
                    const int32_t* m_dev, float* Y, hipStream_t st) {
    const int CAp = (CA + 1) & ~1;
    const size_t smem = (size_t)4 * 32 * (CAp + 1) * sizeof(float);
    if (smem > 64 * 1024) DD3D_HIP(hipFuncSetAttribute((const void*)k_conv_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    k_conv_any<<<dim3((unsigned)ceil_div64(M, 128)), 256, smem, st>>>(X, CA, W, CB, wa, wb, flip, nbr, ld, K, M, m_dev, Y);
    DD3D_CHECK_LAUNCH();
    return 0;
}

bool any_shape_ok(int Cin, int Cout, int K) { return Cin >= 1 && Cin <= ANY_MAXC && Cout >= 1 && Cout <= ANY_MAXC && K >= 1 && K <= 343; }

}  // namespace

DD3D_API int dd3d_conv_any_max_channels(void) { return ANY_MAXC; }

// See include/dd3d.h.
DD3D_API int dd3d_conv_any_fwd(const float* X, int64_t M_in, int Cin, const float* W, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                               const int32_t* m_dev, float* Y, void* stream) {
    if (M_out < 0 || M_in < 0 || !any_shape_ok(Cin, Cout, K) || (nbr && ld < M_out) || (!nbr && (K != 1 || M_in < M_out))) return DD3D_EARG;
    if (M_out == 0) return 0;
    if (!X || !W || !Y) return DD3D_EARG;
    return launch_conv_any(X, Cin, W, Cout, Cout, 1, 0, nbr, ld, K, M_out, m_dev, Y, (hipStream_t)stream);
}

DD3D_API int dd3d_conv_any_bwd_data(const float* dY, int64_t M, const int32_t* m_dev, int Cout, const float* W, int Cin, const int32_t* nbr, int64_t ld,
                                    int K, float* dX, void* stream) {
    if (M < 0 || !any_shape_ok(Cin, Cout, K) || !(K & 1) || (nbr && ld < M) || (!nbr && K != 1)) return DD3D_EARG;
    if (M == 0) return 0;
    if (!dY || !W || !dX) return DD3D_EARG;
    // dX[i][ci] = sum_k sum_co dY[nbr[K-1-k][i]][co] * W[k][ci][co]: contraction index co (pitch 1), output index ci (pitch Cout)
    return launch_conv_any(dY, Cout, W, Cin, 1, Cout, 1, nbr, ld, K, M, m_dev, dX, (hipStream_t)stream);
}

DD3D_API int64_t dd3d_conv_any_bwd_weight_ws_bytes(int K, int Cin, int Cout) {
    if (!any_shape_ok(Cin, Cout, K)) return 0;
    return align_up64((int64_t)ANY_NCH * K * Cin * Cout * (int64_t)sizeof(float), 256);
}

DD3D_API int dd3d_conv_any_bwd_weight(const float* X, int Cin, const float* dY, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                                      const int32_t* m_dev, float* dW, void* ws, int64_t ws_bytes, void* stream) {
    if (M_out < 0 || !any_shape_ok(Cin, Cout, K) || (nbr && ld < M_out) || (!nbr && K != 1) || !dW) return DD3D_EARG;
    if (ws_bytes < dd3d_conv_any_bwd_weight_ws_bytes(K, Cin, Cout) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    const int64_t elems = (int64_t)K * Cin * Cout;
    if (M_out == 0) {
        DD3D_HIP(hipMemsetAsync(dW, 0, elems * sizeof(float), st));
        return 0;
    }
    if (!X || !dY) return DD3D_EARG;
    const int nA = (Cin + 31) / 32;
    k_conv_any_bww<<<dim3((unsigned)((ANY_NCH * nA + 3) / 4), (unsigned)K), 256, 0, st>>>(X, Cin, dY, Cout, nbr, ld, K, M_out, m_dev, (float*)ws);
    k_conv_any_reduce<<<dim3((unsigned)ceil_div64(elems, 256)), 256, 0, st>>>((const float*)ws, elems, dW);
    DD3D_CHECK_LAUNCH();
    return 0;
}
