// (a) Sparse-aware classifier: SparseTensor.dense() + nn.Linear(100352,10) (models/MEConv.py:250-256) without ever
//     materialising the [B,100352] dense tensor.  A voxel v of cloud b with dense site s = site[v] touches the C weight columns
//     {c*S + s}; all three products below are closed under differentiation, which is what the second-order part of the
//     matching loss needs (the reference gets it from torch autograd through index_put / addmm):
//        fwd   : out[b][j]          = sum_{v in b} sum_c F[v][c] * Wl[g(b)][j][c][s]
//        bwd_x : dF[v][c]           = sum_j dl[b(v)][j] * Wl[g][j][c][s]
//        bwd_w : dWl[g][j][c][s]    = sum_{b in g, v = vox_at[b][s]} dl[b][j] * F[v][c]
//     Wl is addressed with explicit element strides so that both the parameter layout [J, C*S] and a site-major copy work.
// (b) The gradient distance of distillation_loss.py:26-130 as one segmented reduction over all layers + its backward.
// (c) softmax cross-entropy (mean over a group's clouds) with first and second-order backward, and the SGD/clamp update of
//     mnist_distill.py:311-317,369 for the batched step.
#include "common.h"
#include "../../include/dd3d.h"

namespace {

struct WStride {
    int64_t g, j, c, s;
};

// Forward in two short launches instead of one long per-cloud loop (a cloud's voxels were walked serially, one dependent
// site -> weight-row round trip each): (1) SIXTEEN lanes per voxel (four voxels per wave) -- a lane holds 8 of the voxel's 128 features
// and multiplies them into its 32-B piece of each output's 512-B weight row; the 16 partial sums meet in four DPP adds (no LDS
// traffic), part[v][j]; (2) one thread per (cloud, output) adds the cloud's voxels in row order (+ bias): deterministic.
__device__ __forceinline__ float row16_sum(float v) {
    // sum over the 16 lanes of a DPP row; every lane ends with the total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

__global__ __launch_bounds__(256) void k_sl_fwd_part(const float* __restrict__ F, const int* __restrict__ site, const int4* __restrict__ coords,
                                                     const int* __restrict__ m_ptr, const int* __restrict__ cloud_group, const float* __restrict__ Wl,
                                                     WStride ws, int J, float* __restrict__ part) {
    const int l16 = threadIdx.x & 15;
    const int M = *m_ptr;
    const bool unit_c = ws.c == 1;
    // grid-stride over the voxels: the launch is sized for the chip, not for the bound (a bound-sized grid of small workgroups
    // would spend its time dispatching empty workgroups).  The trip count is uniform over a wave (the DPP adds need all 16 lanes of a row).
    for (int v0 = blockIdx.x * 16; v0 < M; v0 += gridDim.x * 16) {
        const int v = v0 + (threadIdx.x >> 4);
        const bool live = v < M;
        const int vc = live ? v : M - 1;
        const int s = site[vc];
        const int g = cloud_group ? cloud_group[coords[vc].x] : 0;
        const f32x4* Fv = (const f32x4*)(F + (int64_t)vc * 128 + l16 * 8);
        const f32x4 f0 = Fv[0], f1 = Fv[1];
        const float* Ws = Wl + g * ws.g + (int64_t)max(s, 0) * ws.s;
        float* out = part + (int64_t)vc * J;
        for (int j0 = 0; j0 < J; j0 += 10) {
            float acc[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const float* Wj = Ws + (int64_t)min(j0 + u, J - 1) * ws.j;  // clamped: the weight rows are loaded without a branch
                f32x4 w0, w1;
                if (unit_c) {
                    w0 = *(const f32x4*)(Wj + l16 * 8);
                    w1 = *(const f32x4*)(Wj + l16 * 8 + 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w0[e] = Wj[(int64_t)(l16 * 8 + e) * ws.c];
                        w1[e] = Wj[(int64_t)(l16 * 8 + 4 + e) * ws.c];
                    }
                }
                acc[u] = f0[0] * w0[0] + f0[1] * w0[1] + f0[2] * w0[2] + f0[3] * w0[3] + f1[0] * w1[0] + f1[1] * w1[1] + f1[2] * w1[2] + f1[3] * w1[3];
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const float t = row16_sum(acc[u]);
                if (live && l16 == u && j0 + u < J) out[j0 + u] = s < 0 ? 0.f : t;
            }
        }
    }
}

__global__ void k_sl_fwd_sum(const float* __restrict__ part, const int* __restrict__ seg_off, int B, int J, const float* __restrict__ bias,
                             float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * J) return;
    const int b = t / J, j = t - b * J;
    float s = 0.f;
    int v = seg_off[b];
    const int v1 = seg_off[b + 1];
    for (; v < v1; v += 16) {  // sixteen loads in flight (clamped index, masked add), added in row order: same sum, bit for bit, as one by one
        float x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = part[(int64_t)min(v + q, v1 - 1) * J + j];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (v + q < v1) s += x[q];
    }
    out[t] = bias ? s + bias[j] : s;
}

// dF[v][c] = sum_j dl[b(v)][j] * Wl[g][j][c][s(v)]: every voxel is independent -> grid over voxels (8 per workgroup, 32 float4 channel
// lanes each: with a site-major operand a lane reads one 16-B piece of each output's weight row)
__global__ __launch_bounds__(256) void k_sl_bwd_x(const float* __restrict__ dl, const int* __restrict__ site, const int4* __restrict__ coords,
                                                  const int* __restrict__ m_ptr, const int* __restrict__ cloud_group, const float* __restrict__ Wl,
                                                  WStride ws, int J, int accumulate, float* __restrict__ dF) {
    const int c4 = threadIdx.x & 31;
    const int M = *m_ptr;
    const bool unit_c = ws.c == 1;
    for (int v = blockIdx.x * 8 + (threadIdx.x >> 5); v < M; v += gridDim.x * 8) {  // grid-stride: see k_sl_fwd_part
        const int s = site[v];
        const int b = coords[v].x;
        const int g = cloud_group ? cloud_group[b] : 0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s >= 0) {
            const float* Wg = Wl + g * ws.g + (int64_t)(c4 * 4) * ws.c + s * ws.s;
            const float* d = dl + (int64_t)b * J;
            for (int j0 = 0; j0 < J; j0 += 10) {
                f32x4 w[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const float* Wj = Wg + (int64_t)min(j0 + u, J - 1) * ws.j;
                    if (unit_c) {
                        w[u] = *(const f32x4*)Wj;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[u][e] = Wj[(int64_t)e * ws.c];
                    }
                }
#pragma unroll
                for (int u = 0; u < 10; ++u)
                    if (j0 + u < J) acc += d[j0 + u] * w[u];
            }
        }
        f32x4* o = (f32x4*)(dF + (int64_t)v * 128) + c4;
        *o = accumulate ? *o + acc : acc;
    }
}

// grid (S, G): one workgroup per (site, group).  The clouds of the group that have a voxel at the site (about a fifth of them for
// stroke digits) are first COMPACTED into an LDS list, in cloud order (one coalesced read of the table + a ballot scan: one memory
// round trip); 8 row-lanes x 32 float4 channel-lanes then walk the list, row-lane r taking entries r, r + 8, ... with four row
// gathers in flight, so a (site, group) costs two or three dependent round trips instead of two per four clouds.  The row-lanes are
// combined in lane order: deterministic.  `zero_empty`: a site no cloud of the group occupies is written as zeros here (layouts with
// the channels innermost: whole 512-byte rows); otherwise such sites are skipped and the caller zero-fills the buffer first.
__global__ __launch_bounds__(256) void k_sl_bwd_w(const float* __restrict__ dl, const float* __restrict__ F, const int* __restrict__ vox_at, int S, int B,
                                                  const int* __restrict__ group_cloud_off, int J, float* __restrict__ dWl, WStride ws, int zero_empty) {
    constexpr int LIST = 1024;  // clouds per compaction pass
    __shared__ int s_v[LIST], s_b[LIST], s_wc[16];
    __shared__ f32x4 s_red[4][8][32];
    const int s = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c4 = tid & 31, rl = tid >> 5;
    const int b0 = group_cloud_off[g], b1 = group_cloud_off[g + 1];
    const int* va = vox_at + (int64_t)s * B;
    const f32x4* F4 = (const f32x4*)F;
    float* out = dWl + g * ws.g + s * ws.s;
    const bool unit_c = ws.c == 1;
    bool any = false;
    for (int j0 = 0; j0 < J; j0 += 16) {
        f32x4 acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int base = b0; base < b1; base += LIST) {
            // ---- the occupied clouds of [base, base + LIST), in cloud order ----
            int v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = base + q * 256 + tid;
                v[q] = b < b1 ? va[b] : -1;
            }
            __syncthreads();  // the previous pass's list is no longer read
            unsigned long long bal[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bal[q] = __ballot(v[q] >= 0);
                if (lane == 0) s_wc[q * 4 + wid] = __popcll(bal[q]);
            }
            __syncthreads();
            int n = 0, pos0[4];
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int c = s_wc[x];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (x == q * 4 + wid) pos0[q] = n;
                n += c;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] >= 0) {
                    const int pos = pos0[q] + __popcll(bal[q] & ((1ull << lane) - 1ull));
                    s_v[pos] = v[q];
                    s_b[pos] = base + q * 256 + tid;
                }
            __syncthreads();
            any |= n > 0;
            // ---- row-lane rl: entries rl, rl + 8, ...; four gathers in flight; added in list order ----
            for (int i = rl; i < n; i += 32) {
                int vv[4], bb[4];
                f32x4 ff[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int iq = min(i + 8 * q, n - 1);
                    vv[q] = s_v[iq], bb[q] = s_b[iq];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) ff[q] = F4[(int64_t)vv[q] * 32 + c4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i + 8 * q >= n) continue;
                    const float* d = dl + (int64_t)bb[q] * J + j0;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (j0 + u < J) acc[u] += d[u] * ff[q];
                }
            }
        }
        if (!any && !zero_empty) return;  // (uniform) nothing at this site: the caller's zero fill stands
        // ---- combine the 8 row-lanes: the two of a wave by a shuffle, the four waves through LDS, eight outputs per round ----
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
            if (j0 + h8 * 8 >= J) break;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x4 a = acc[h8 * 8 + u], o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __shfl_xor(a[e], 32, 64);
                if (lane < 32) s_red[wid][u][c4] = a + o;  // row-lane 2w + row-lane 2w+1
            }
            __syncthreads();
            {  // 8 outputs x 32 channel-lanes = one item per thread
                const int u = tid >> 5, c = tid & 31;
                if (j0 + h8 * 8 + u < J) {
                    const f32x4 t = s_red[0][u][c] + s_red[1][u][c] + s_red[2][u][c] + s_red[3][u][c];
                    float* o = out + (int64_t)(j0 + h8 * 8 + u) * ws.j;
                    if (unit_c) {
                        *(f32x4*)(o + c * 4) = t;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[(int64_t)(c * 4 + e) * ws.c] = t[e];
                    }
                }
            }
            __syncthreads();
        }
    }
}

// site-major copy of a parameter-layout operand: out[g][s][j][c] = in[g][j][c*S + s]  (and back with inverse = 1)
__global__ __launch_bounds__(256) void k_classifier_layout(const float* __restrict__ in, int J, int C, int S, int inverse, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int j = blockIdx.z % J, g = blockIdx.z / J;
    const int s0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t gbase = (int64_t)g * J * C * S;
    if (!inverse) {
        for (int q = ty; q < 32; q += 8) {
            int c = c0 + q, sx = s0 + tx;
            tile[q][tx] = (c < C && sx < S) ? in[gbase + ((int64_t)j * C + c) * S + sx] : 0.f;
        }
        __syncthreads();
        for (int q = ty; q < 32; q += 8) {
            int sx = s0 + q, c = c0 + tx;
            if (c < C && sx < S) out[gbase + ((int64_t)sx * J + j) * C + c] = tile[tx][q];
        }
    } else {
        for (int q = ty; q < 32; q += 8) {
            int sx = s0 + q, c = c0 + tx;
            tile[q][tx] = (c < C && sx < S) ? in[gbase + ((int64_t)sx * J + j) * C + c] : 0.f;
        }
        __syncthreads();
        for (int q = ty; q < 32; q += 8) {
            int c = c0 + q, sx = s0 + tx;
            if (c < C && sx < S) out[gbase + ((int64_t)j * C + c) * S + sx] = tile[tx][q];
        }
    }
}

// ---- softmax cross-entropy over [B,J] logits, mean over the clouds of each group --------------------------------
// per cloud: p = softmax(z), dlogits = (p - onehot)/B_g, cloud_loss = (lse - z[y])/B_g ; then one thread per group sums in order
__global__ void k_softmax_ce(const float* __restrict__ logits, const int* __restrict__ labels, const int* __restrict__ cloud_group,
                             const int* __restrict__ group_cloud_off, int B, int J, float* __restrict__ prob, float* __restrict__ dlogits,
                             float* __restrict__ cloud_loss) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int g = cloud_group ? cloud_group[b] : 0;
    float inv = 1.0f / (float)max(group_cloud_off[g + 1] - group_cloud_off[g], 1);
    const float* z = logits + (int64_t)b * J;
    float m = z[0];
    for (int j = 1; j < J; ++j) m = fmaxf(m, z[j]);
    float se = 0.f;
    for (int j = 0; j < J; ++j) se += expf(z[j] - m);
    float lse = m + logf(se);
    int y = labels[b];
    for (int j = 0; j < J; ++j) {
        float p = expf(z[j] - lse);
        if (prob) prob[(int64_t)b * J + j] = p;
        dlogits[(int64_t)b * J + j] = (p - (j == y ? 1.f : 0.f)) * inv;
    }
    cloud_loss[b] = (lse - z[y]) * inv;
}

// one wave per (group, column): lanes take clouds b0+lane, b0+lane+64, ... in order, then a fixed shuffle tree -> deterministic
__global__ void k_group_sum(const float* __restrict__ v, const int* __restrict__ group_cloud_off, int G, float* __restrict__ out) {
    int g = blockIdx.x;
    float s = 0.f;
    for (int b = group_cloud_off[g] + threadIdx.x; b < group_cloud_off[g + 1]; b += 64) s += v[b];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[g] = s;
}

// out[g][j] = sum over the clouds b of group g of v[b][j]: one wave per (g, j), same fixed order as k_group_sum
__global__ void k_group_sum_rows(const float* __restrict__ v, const int* __restrict__ group_cloud_off, int G, int J, float* __restrict__ out) {
    int g = blockIdx.x / J, j = blockIdx.x - g * J;
    float s = 0.f;
    for (int b = group_cloud_off[g] + threadIdx.x; b < group_cloud_off[g + 1]; b += 64) s += v[(int64_t)b * J + j];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// Classifier head in ONE launch (after k_sl_fwd_part): logits = sum of the voxel partials (+ bias), softmax cross entropy with its
// logit gradient, and the two per-group sums -- mean CE per group and the bias gradient sum_b dlogits[b][:] -- that k_sl_fwd_sum,
// k_softmax_ce, k_group_sum and k_group_sum_rows compute as four dependent launches (~5 us of dispatch latency each on the critical path
// of a step).  Workgroup (x, g) takes clouds [64 x, 64 x + 64) of group g; the last workgroup of a group to finish (a counter per group,
// left at zero again for the next launch) adds the group's rows up in exactly the order of k_group_sum / k_group_sum_rows, so the results
// are the same bits.
__global__ __launch_bounds__(256) void k_cls_head(const float* __restrict__ part, const int* __restrict__ seg_off, int J, const float* __restrict__ bias,
                                                  const int* __restrict__ labels, const int* __restrict__ group_cloud_off, float* __restrict__ logits,
                                                  float* __restrict__ prob, float* __restrict__ dlogits, float* __restrict__ cloud_loss,
                                                  float* __restrict__ loss_g, float* __restrict__ bias_grad /*[G][J] or null*/, int* __restrict__ counter) {
    __shared__ int s_last;
    __shared__ float s_z[64 * 16];  // the workgroup's logits when J <= 16 (the softmax reads them back from LDS instead of through L2)
    const int g = blockIdx.y, tid = threadIdx.x;
    const int b0 = group_cloud_off[g], b1 = group_cloud_off[g + 1];
    const int c0 = b0 + blockIdx.x * 64, c1 = min(c0 + 64, b1);
    if (c0 >= b1 && !(blockIdx.x == 0 && b1 == b0)) return;  // (an empty group: its first workgroup writes the zeros)
    for (int t = tid; t < (c1 - c0) * J; t += 256) {         // k_sl_fwd_sum: one (cloud, column) per thread, rows added in order
        const int b = c0 + t / J, j = t % J;
        float s = 0.f;
        int v = seg_off[b];
        const int v1 = seg_off[b + 1];
        // sixteen rows in flight, clamped index and masked add (a cloud has ~12 stride-4 voxels: one round trip instead of a batch of
        // eight and a one-by-one tail); the adds run in row order as before
        for (; v < v1; v += 16) {
            float x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = part[(int64_t)min(v + q, v1 - 1) * J + j];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (v + q < v1) s += x[q];
        }
        const float zz = bias ? s + bias[j] : s;
        logits[(int64_t)b * J + j] = zz;
        if (J <= 16) s_z[t] = zz;
    }
    __threadfence_block();
    __syncthreads();
    if (tid < c1 - c0) {  // k_softmax_ce: one cloud per thread
        const int b = c0 + tid;
        const float inv = 1.0f / (float)max(b1 - b0, 1);
        const float* z = J <= 16 ? s_z + tid * J : logits + (int64_t)b * J;
        float m = z[0];
        for (int j = 1; j < J; ++j) m = fmaxf(m, z[j]);
        float se = 0.f;
        for (int j = 0; j < J; ++j) se += expf(z[j] - m);
        const float lse = m + logf(se);
        const int y = labels[b];
        for (int j = 0; j < J; ++j) {
            const float p = expf(z[j] - lse);
            if (prob) prob[(int64_t)b * J + j] = p;
            dlogits[(int64_t)b * J + j] = (p - (j == y ? 1.f : 0.f)) * inv;
        }
        cloud_loss[b] = (lse - z[y]) * inv;
    }
    // the group's last workgroup sums its rows
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int nwg = max((b1 - b0 + 63) / 64, 1);
        const int done = atomicAdd(&counter[g], 1) + 1;
        s_last = done == nwg;
        if (s_last) counter[g] = 0;  // every workgroup of the group has passed its atomic: ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const int lane = tid & 63, w = tid >> 6;
    for (int col = w; col <= J; col += 4) {  // column J = the cloud losses
        if (col < J && !bias_grad) continue;
        if (col == J && !loss_g) continue;
        float s = 0.f;
        for (int b = b0 + lane; b < b1; b += 64)
            s += col < J ? dlogits[(int64_t)b * J + col] : cloud_loss[b];  // (written by other workgroups: read behind the fence above)
        s = wave_sum(s);
        if (lane == 0) {
            if (col < J) bias_grad[(int64_t)g * J + col] = s;
            else loss_g[g] = s;
        }
    }
}

// per_class[g] = ((a[g] + b[g]) + c[g]) + d[g] (b, c, d nullable) and total = sum_g per_class[g]: the partial matching losses of a step and their
// sum in ONE launch (torch: two additions and a reduction, three dependent graph nodes at the very end of a step).  One wave; lane l adds
// groups l, l + 64, ... in order, then the fixed shuffle tree of wave_sum: deterministic.
__global__ void k_loss_total(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, const float* __restrict__ d, int G,
                             float* __restrict__ per_class, float* __restrict__ total) {
    float s = 0.f;
    for (int g = threadIdx.x; g < G; g += 64) {
        float v = a[g];
        if (b) v += b[g];
        if (c) v += c[g];
        if (d) v += d[g];
        if (per_class) per_class[g] = v;
        s += v;
    }
    s = wave_sum(s);
    if (threadIdx.x == 0 && total) *total = s;
}

// second order: dlogits = (softmax(z) - onehot)/B_g ; given u = dL/d(dlogits) -> dz = p * (u - <p,u>) / B_g
__global__ void k_softmax_ce_bwd2(const float* __restrict__ prob, const float* __restrict__ u, const float* __restrict__ u_group /*[G][J] or null*/,
                                  const int* __restrict__ group_cloud_off, int G, int J, float* __restrict__ dz) {
    int g = blockIdx.y;
    int b = group_cloud_off[g] + blockIdx.x * blockDim.x + threadIdx.x;
    int b1 = group_cloud_off[g + 1];
    if (b >= b1) return;
    float inv = 1.0f / (float)(b1 - group_cloud_off[g]);
    const float* p = prob + (int64_t)b * J;
    const float* uu = u + (int64_t)b * J;
    const float* ug = u_group ? u_group + (int64_t)g * J : nullptr;   // per-group term of u (the gradient through gw[bias] = sum dlogits)
    float d = 0.f;
    for (int j = 0; j < J; ++j) d += p[j] * (uu[j] + (ug ? ug[j] : 0.f));
    for (int j = 0; j < J; ++j) dz[(int64_t)b * J + j] = p[j] * (uu[j] + (ug ? ug[j] : 0.f) - d) * inv;
}

// ---- gradient distance ------------------------------------------------------------------------------------------
constexpr int GD_ROW_CHUNK = 8192;  // elements of one contiguous row per work item
constexpr int GD_COL_CHUNK = 64;    // logical columns per work item in column-major segments (128 rows per item)
constexpr int GD_U = 8;             // elements of a thread whose loads are in flight together

struct GPlan {
    int n, metric;
    int item_off[DD3D_MAX_SEG + 1];  // work items per segment (prefix)
    int row_off[DD3D_MAX_SEG + 1];   // logical rows per segment (prefix)
    int part_off[DD3D_MAX_SEG + 1];  // partial records (row x chunk) per segment (prefix)
    int nchunk[DD3D_MAX_SEG];
    dd3d_gseg seg[DD3D_MAX_SEG];
};

__device__ __forceinline__ int find_seg(const GPlan& p, int item) {
    int s = 0;
    while (s + 1 < p.n && item >= p.item_off[s + 1]) ++s;
    return s;
}

// partial[(part_off[s] + row*nchunk + chunk)] = (dot, ss, rr, sqdiff) over the item's elements
__global__ __launch_bounds__(256) void k_gd_stats(const GPlan p, float4* __restrict__ partial) {
    __shared__ float s_red[4 * 4];
    __shared__ float4 s_half[128];
    const int item = blockIdx.x, tid = threadIdx.x;
    const int s = find_seg(p, item);
    dd3d_gseg sg = p.seg[s];
    const int64_t goff = (int64_t)blockIdx.y * sg.rows * sg.cols;  // group g: every segment tensor is [G, rows*cols]
    sg.syn += goff;
    sg.real += goff;
    partial += (int64_t)blockIdx.y * p.part_off[p.n];
    const int local = item - p.item_off[s];
    const int nch = p.nchunk[s];
    if (!sg.colmajor) {
        int row = local / nch, ch = local - row * nch;
        int64_t e0 = (int64_t)ch * GD_ROW_CHUNK, e1 = min(e0 + GD_ROW_CHUNK, sg.cols);
        const float* a = sg.syn + row * sg.cols;
        const float* b = sg.real + row * sg.cols;
        float dot = 0.f, ss = 0.f, rr = 0.f, sq = 0.f;
        // GD_U elements of the thread at a time, their loads issued together (one element per iteration left the loop a chain of 32
        // dependent round trips: the compiler does not unroll it).  Same order of additions; the padding adds +0.
        for (int64_t e = e0 + tid; e < e1; e += 256 * GD_U) {
            float x[GD_U], y[GD_U];
#pragma unroll
            for (int j = 0; j < GD_U; ++j) {
                const int64_t ej = e + 256 * j, ec = min(ej, e1 - 1);
                x[j] = a[ec]; y[j] = b[ec];
                if (ej >= e1) x[j] = y[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < GD_U; ++j) {
                const float d = x[j] - y[j];
                dot += x[j] * y[j]; ss += x[j] * x[j]; rr += y[j] * y[j]; sq += d * d;
            }
        }
        dot = wave_sum(dot); ss = wave_sum(ss); rr = wave_sum(rr); sq = wave_sum(sq);
        if ((tid & 63) == 0) {
            int w = tid >> 6;
            s_red[w * 4 + 0] = dot; s_red[w * 4 + 1] = ss; s_red[w * 4 + 2] = rr; s_red[w * 4 + 3] = sq;
        }
        __syncthreads();
        if (tid == 0) {
            float4 o;
            o.x = s_red[0] + s_red[4] + s_red[8] + s_red[12];
            o.y = s_red[1] + s_red[5] + s_red[9] + s_red[13];
            o.z = s_red[2] + s_red[6] + s_red[10] + s_red[14];
            o.w = s_red[3] + s_red[7] + s_red[11] + s_red[15];
            partial[p.part_off[s] + (int64_t)row * nch + ch] = o;
        }
    } else {
        // logical row r = memory-contiguous index; logical column q strides by `rows`
        int rtile = local / nch, ch = local - rtile * nch;
        int r = rtile * 128 + (tid & 127), half = tid >> 7;
        int64_t q0 = (int64_t)ch * GD_COL_CHUNK, q1 = min(q0 + GD_COL_CHUNK, sg.cols);
        float dot = 0.f, ss = 0.f, rr = 0.f, sq = 0.f;
        if (r < sg.rows)
            for (int64_t q = q0 + half; q < q1; q += 2 * GD_U) {
                float x[GD_U], y[GD_U];
#pragma unroll
                for (int j = 0; j < GD_U; ++j) {
                    const int64_t qj = q + 2 * j, qc = min(qj, q1 - 1);
                    x[j] = sg.syn[qc * sg.rows + r]; y[j] = sg.real[qc * sg.rows + r];
                    if (qj >= q1) x[j] = y[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < GD_U; ++j) {
                    const float d = x[j] - y[j];
                    dot += x[j] * y[j]; ss += x[j] * x[j]; rr += y[j] * y[j]; sq += d * d;
                }
            }
        if (half == 1) s_half[tid & 127] = make_float4(dot, ss, rr, sq);
        __syncthreads();
        if (half == 0 && r < sg.rows) {
            float4 t = s_half[tid];
            partial[p.part_off[s] + (int64_t)r * nch + ch] = make_float4(dot + t.x, ss + t.y, rr + t.z, sq + t.w);
        }
    }
}

__device__ __forceinline__ float row_term(int metric, float4 st) {
    if (metric == 0) return 1.0f - st.x / (sqrtf(st.z) * sqrtf(st.y) + 0.000001f);
    return st.w;
}

// one block per group: per-row stats in chunk order, then the loss as a fixed pairwise tree over the rows -> fully deterministic
__global__ __launch_bounds__(256) void k_gd_finalize(const GPlan p, const float4* __restrict__ partial, float4* __restrict__ stats,
                                                     float* __restrict__ loss) {
    __shared__ float s_term[256];
    __shared__ float4 s_tot[256];
    const int tid = threadIdx.x;
    const int total_rows = p.row_off[p.n];
    partial += (int64_t)blockIdx.x * p.part_off[p.n];
    stats += (int64_t)blockIdx.x * (total_rows + 1);
    loss += blockIdx.x;
    float run = 0.f;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < total_rows; base += 256) {
        int gr = base + tid;
        float term = 0.f;
        float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < total_rows) {
            int s = 0;
            while (s + 1 < p.n && gr >= p.row_off[s + 1]) ++s;
            int row = gr - p.row_off[s];
            int nch = p.nchunk[s];
            const float4* pp = partial + p.part_off[s] + (int64_t)row * nch;
            int c = 0;
            for (; c + 4 <= nch; c += 4) {  // four loads in flight, added in chunk order
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = pp[c + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) { st.x += v[q].x; st.y += v[q].y; st.z += v[q].z; st.w += v[q].w; }
            }
            for (; c < nch; ++c) {
                float4 v = pp[c];
                st.x += v.x; st.y += v.y; st.z += v.z; st.w += v.w;
            }
            stats[gr] = st;
            term = row_term(p.metric, st);
        }
        s_term[tid] = term;
        s_tot[tid] = st;
        __syncthreads();
        // fixed pairwise tree over the 256 slots (rows beyond total_rows hold zeros): deterministic, 8 short steps instead of a
        // 256-long serial chain on one thread
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                s_term[tid] += s_term[tid + off];
                float4 a = s_tot[tid], b = s_tot[tid + off];
                s_tot[tid] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
            __syncthreads();
        }
        if (tid == 0) {
            run += s_term[0];
            tot.x += s_tot[0].x; tot.y += s_tot[0].y; tot.z += s_tot[0].z; tot.w += s_tot[0].w;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float L;
        if (p.metric == 0 || p.metric == 1) L = run;
        else if (p.metric == 2) L = tot.w;
        else L = 1.0f - tot.x / (sqrtf(tot.z) * sqrtf(tot.y) + 0.000001f);
        loss[0] = L;
        stats[total_rows] = tot;  // global record used by the 'cos' backward
    }
}

// d loss / d syn for every segment with a dsyn buffer
__global__ __launch_bounds__(256) void k_gd_bwd(const GPlan p, const float4* __restrict__ stats, const float* __restrict__ gl) {
    const int item = blockIdx.x, tid = threadIdx.x;
    const int s = find_seg(p, item);
    dd3d_gseg sg = p.seg[s];
    if (!sg.dsyn) return;
    const int64_t goff = (int64_t)blockIdx.y * sg.rows * sg.cols;
    sg.syn += goff;
    sg.real += goff;
    sg.dsyn += goff;
    const int local = item - p.item_off[s];
    const int nch = p.nchunk[s];
    const float up = gl ? gl[0] : 1.0f;
    const int total_rows = p.row_off[p.n];
    stats += (int64_t)blockIdx.y * (total_rows + 1);
    const int metric = p.metric;
    auto coef = [&](float4 st, float* ca, float* cb) {
        // d/ds [1 - <r,s>/(|r||s|+eps)] = -r/den + <r,s>*|r|/(|s|*den^2) * s
        float nr = sqrtf(st.z), ns = sqrtf(st.y), den = nr * ns + 0.000001f;
        *ca = -up / den;
        *cb = ns > 0.f ? up * st.x * nr / (ns * den * den) : 0.f;
    };
    if (!sg.colmajor) {
        int row = local / nch, ch = local - row * nch;
        int64_t e0 = (int64_t)ch * GD_ROW_CHUNK, e1 = min(e0 + GD_ROW_CHUNK, sg.cols);
        const float* a = sg.syn + row * sg.cols;
        const float* b = sg.real + row * sg.cols;
        float* d = sg.dsyn + row * sg.cols;
        float ca = 0.f, cb = 0.f;
        if (metric == 0) coef(stats[p.row_off[s] + row], &ca, &cb);
        else if (metric == 3) coef(stats[total_rows], &ca, &cb);
        for (int64_t e = e0 + tid; e < e1; e += 256 * GD_U) {
            float x[GD_U], y[GD_U];
#pragma unroll
            for (int j = 0; j < GD_U; ++j) {
                const int64_t ec = min(e + 256 * j, e1 - 1);
                x[j] = a[ec]; y[j] = b[ec];
            }
#pragma unroll
            for (int j = 0; j < GD_U; ++j)
                if (e + 256 * j < e1) d[e + 256 * j] = (metric == 1 || metric == 2) ? 2.0f * up * (x[j] - y[j]) : ca * y[j] + cb * x[j];
        }
    } else {
        int rtile = local / nch, ch = local - rtile * nch;
        int r = rtile * 128 + (tid & 127), half = tid >> 7;
        if (r >= sg.rows) return;
        int64_t q0 = (int64_t)ch * GD_COL_CHUNK, q1 = min(q0 + GD_COL_CHUNK, sg.cols);
        float ca = 0.f, cb = 0.f;
        if (metric == 0) coef(stats[p.row_off[s] + r], &ca, &cb);
        else if (metric == 3) coef(stats[total_rows], &ca, &cb);
        for (int64_t q = q0 + half; q < q1; q += 2 * GD_U) {
            float x[GD_U], y[GD_U];
#pragma unroll
            for (int j = 0; j < GD_U; ++j) {
                const int64_t ec = min(q + 2 * j, q1 - 1) * sg.rows + r;
                x[j] = sg.syn[ec]; y[j] = sg.real[ec];
            }
#pragma unroll
            for (int j = 0; j < GD_U; ++j)
                if (q + 2 * j < q1) sg.dsyn[(q + 2 * j) * sg.rows + r] = (metric == 1 || metric == 2) ? 2.0f * up * (x[j] - y[j]) : ca * y[j] + cb * x[j];
        }
    }
}

// k_gd_stats + k_gd_finalize (+ k_gd_bwd) in ONE launch for SMALL lists: at most GD_SMALL_PASSES block-wide passes in all -- a row-major row
// of at most GD_SMALL_COLS columns is one pass, a column-major segment one pass per (128 rows, 64 columns) -- and at most 256 rows: an
// instance-norm weight's gradient ([1,128] per class), the first layer's kernel gradient ([27,3,128]: 128 rows of 81), the terms that sit
// alone on the tail of a step.  Such a list is three dependent launches of a handful of workgroups each, i.e. ~20 us of dispatch latency for
// microseconds of work.  One workgroup per group walks the list with EXACTLY the arithmetic of the three kernels (the same per-thread element
// order, wave sums, four-wave combine, half-block combine, chunk order, 256-slot pairwise tree, coefficients): every output has the same bits.
constexpr int GD_SMALL_PASSES = 8;    // (a workgroup takes its passes one after the other: longer lists are faster as three wide launches --
constexpr int GD_SMALL_COLS = 2048;   //  measured: the classifier's twelve rows of up to 6272 columns in one launch: +58 us per step)
__global__ __launch_bounds__(256) void k_gd_small(const GPlan p, float4* __restrict__ stats, float* __restrict__ loss, const float* __restrict__ gl, int bwd) {
    __shared__ float s_red[4 * 4];
    __shared__ float4 s_half[128];
    __shared__ float s_term[256];
    __shared__ float4 s_tot[256];
    const int tid = threadIdx.x, g = blockIdx.x;
    const int total_rows = p.row_off[p.n];
    stats += (int64_t)g * (total_rows + 1);
    s_term[tid] = 0.f;
    s_tot[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int s = 0; s < p.n; ++s) {
        const dd3d_gseg sg = p.seg[s];
        const int64_t goff = (int64_t)g * sg.rows * sg.cols;
        const float* syn = sg.syn + goff;
        const float* real = sg.real + goff;
        if (!sg.colmajor) {
            for (int row = 0; row < sg.rows; ++row) {
                const int gr = p.row_off[s] + row;
                const int64_t e1 = sg.cols;
                const float* a = syn + row * sg.cols;
                const float* b = real + row * sg.cols;
                float dot = 0.f, ss = 0.f, rr = 0.f, sq = 0.f;
                for (int64_t e = tid; e < e1; e += 256 * GD_U) {
                    float x[GD_U], y[GD_U];
#pragma unroll
                    for (int j = 0; j < GD_U; ++j) {
                        const int64_t ej = e + 256 * j, ec = min(ej, e1 - 1);
                        x[j] = a[ec]; y[j] = b[ec];
                        if (ej >= e1) x[j] = y[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < GD_U; ++j) {
                        const float d = x[j] - y[j];
                        dot += x[j] * y[j]; ss += x[j] * x[j]; rr += y[j] * y[j]; sq += d * d;
                    }
                }
                dot = wave_sum(dot); ss = wave_sum(ss); rr = wave_sum(rr); sq = wave_sum(sq);
                if ((tid & 63) == 0) {
                    const int w = tid >> 6;
                    s_red[w * 4 + 0] = dot; s_red[w * 4 + 1] = ss; s_red[w * 4 + 2] = rr; s_red[w * 4 + 3] = sq;
                }
                __syncthreads();
                if (tid == 0) {
                    float4 st = make_float4(0.f, 0.f, 0.f, 0.f);  // (k_gd_finalize adds the row's one chunk record into zeros)
                    st.x += s_red[0] + s_red[4] + s_red[8] + s_red[12];
                    st.y += s_red[1] + s_red[5] + s_red[9] + s_red[13];
                    st.z += s_red[2] + s_red[6] + s_red[10] + s_red[14];
                    st.w += s_red[3] + s_red[7] + s_red[11] + s_red[15];
                    stats[gr] = st;
                    s_term[gr] = row_term(p.metric, st);
                    s_tot[gr] = st;
                }
                __syncthreads();
            }
        } else {
            const int nch = p.nchunk[s], half = tid >> 7;
            for (int rt = 0; rt * 128 < sg.rows; ++rt) {
                const int r = rt * 128 + (tid & 127);
                float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int ch = 0; ch < nch; ++ch) {
                    const int64_t q0 = (int64_t)ch * GD_COL_CHUNK, q1 = min(q0 + GD_COL_CHUNK, sg.cols);
                    float dot = 0.f, ss = 0.f, rr = 0.f, sq = 0.f;
                    if (r < sg.rows)
                        for (int64_t q = q0 + half; q < q1; q += 2 * GD_U) {
                            float x[GD_U], y[GD_U];
#pragma unroll
                            for (int j = 0; j < GD_U; ++j) {
                                const int64_t qj = q + 2 * j, qc = min(qj, q1 - 1);
                                x[j] = syn[qc * sg.rows + r]; y[j] = real[qc * sg.rows + r];
                                if (qj >= q1) x[j] = y[j] = 0.f;
                            }
#pragma unroll
                            for (int j = 0; j < GD_U; ++j) {
                                const float d = x[j] - y[j];
                                dot += x[j] * y[j]; ss += x[j] * x[j]; rr += y[j] * y[j]; sq += d * d;
                            }
                        }
                    if (half == 1) s_half[tid & 127] = make_float4(dot, ss, rr, sq);
                    __syncthreads();
                    if (half == 0 && r < sg.rows) {
                        const float4 t = s_half[tid];
                        const float4 rec = make_float4(dot + t.x, ss + t.y, rr + t.z, sq + t.w);  // the (row, chunk) record of k_gd_stats ...
                        st.x += rec.x; st.y += rec.y; st.z += rec.z; st.w += rec.w;               // ... added in chunk order, as k_gd_finalize does
                    }
                    __syncthreads();
                }
                if (half == 0 && r < sg.rows) {
                    const int gr = p.row_off[s] + r;
                    stats[gr] = st;
                    s_term[gr] = row_term(p.metric, st);
                    s_tot[gr] = st;
                }
            }
            __syncthreads();
        }
    }
    for (int off = 128; off > 0; off >>= 1) {  // the pairwise tree of k_gd_finalize over the 256 slots (rows beyond total_rows hold zeros)
        if (tid < off) {
            s_term[tid] += s_term[tid + off];
            float4 x = s_tot[tid], y = s_tot[tid + off];
            s_tot[tid] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        }
        __syncthreads();
    }
    if (tid == 0) {
        float run = 0.f;
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        run += s_term[0];
        tot.x += s_tot[0].x; tot.y += s_tot[0].y; tot.z += s_tot[0].z; tot.w += s_tot[0].w;
        float L;
        if (p.metric == 0 || p.metric == 1) L = run;
        else if (p.metric == 2) L = tot.w;
        else L = 1.0f - tot.x / (sqrtf(tot.z) * sqrtf(tot.y) + 0.000001f);
        loss[g] = L;
        stats[total_rows] = tot;
    }
    if (!bwd) return;
    __syncthreads();  // (workgroup-scope fence: the stats written above are read back below)
    const float up = gl ? gl[0] : 1.0f;
    const int metric = p.metric;
    auto coef = [&](float4 st, float* ca, float* cb) {
        const float nr = sqrtf(st.z), ns = sqrtf(st.y), den = nr * ns + 0.000001f;
        *ca = -up / den;
        *cb = ns > 0.f ? up * st.x * nr / (ns * den * den) : 0.f;
    };
    for (int s = 0; s < p.n; ++s) {
        const dd3d_gseg sg = p.seg[s];
        if (!sg.dsyn) continue;
        const int64_t goff = (int64_t)g * sg.rows * sg.cols;
        const float* syn = sg.syn + goff;
        const float* real = sg.real + goff;
        float* dsyn = sg.dsyn + goff;
        if (!sg.colmajor) {
            for (int row = 0; row < sg.rows; ++row) {
                const int64_t e1 = sg.cols;
                const float* a = syn + row * sg.cols;
                const float* b = real + row * sg.cols;
                float* d = dsyn + row * sg.cols;
                float ca = 0.f, cb = 0.f;
                if (metric == 0) coef(stats[p.row_off[s] + row], &ca, &cb);
                else if (metric == 3) coef(stats[total_rows], &ca, &cb);
                for (int64_t e = tid; e < e1; e += 256 * GD_U) {
                    float x[GD_U], y[GD_U];
#pragma unroll
                    for (int j = 0; j < GD_U; ++j) {
                        const int64_t ec = min(e + 256 * j, e1 - 1);
                        x[j] = a[ec]; y[j] = b[ec];
                    }
#pragma unroll
                    for (int j = 0; j < GD_U; ++j)
                        if (e + 256 * j < e1) d[e + 256 * j] = (metric == 1 || metric == 2) ? 2.0f * up * (x[j] - y[j]) : ca * y[j] + cb * x[j];
                }
            }
        } else {
            const int nch = p.nchunk[s], half = tid >> 7;
            for (int rt = 0; rt * 128 < sg.rows; ++rt) {
                const int r = rt * 128 + (tid & 127);
                if (r >= sg.rows) continue;
                float ca = 0.f, cb = 0.f;
                if (metric == 0) coef(stats[p.row_off[s] + r], &ca, &cb);
                else if (metric == 3) coef(stats[total_rows], &ca, &cb);
                for (int ch = 0; ch < nch; ++ch) {
                    const int64_t q0 = (int64_t)ch * GD_COL_CHUNK, q1 = min(q0 + GD_COL_CHUNK, sg.cols);
                    for (int64_t q = q0 + half; q < q1; q += 2 * GD_U) {
                        float x[GD_U], y[GD_U];
#pragma unroll
                        for (int j = 0; j < GD_U; ++j) {
                            const int64_t ec = min(q + 2 * j, q1 - 1) * sg.rows + r;
                            x[j] = syn[ec]; y[j] = real[ec];
                        }
#pragma unroll
                        for (int j = 0; j < GD_U; ++j)
                            if (q + 2 * j < q1) dsyn[(q + 2 * j) * sg.rows + r] = (metric == 1 || metric == 2) ? 2.0f * up * (x[j] - y[j]) : ca * y[j] + cb * x[j];
                    }
                }
            }
        }
    }
}

// SGD with momentum on the synthetic xyz + the clamp of the NEXT iteration's prologue (mnist_distill.py:311-317,369)
__global__ void k_sgd_clamp(float* __restrict__ x, const float* __restrict__ grad, float* __restrict__ mom, int64_t n, float lr, float momentum,
                            int first_step, int do_clamp, float lo, float hi) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = grad[i];
    float m = first_step ? g : momentum * mom[i] + g;
    mom[i] = m;
    float v = x[i] - lr * m;
    if (do_clamp) {
        v = v < lo ? lo : v;
        v = v > hi ? hi : v;
        if (i % 3 == 0) v = 0.f;
    }
    x[i] = v;
}

// Prologue of an iteration in ONE launch (mnist_distill.py:311-317 + get_syn_tensor :241-244 + create_input_batch MEConv.py:260-267):
// clamp every synthetic point in place ([lo, hi], first axis := 0; comparisons, so a NaN stays a NaN as with the reference's masked
// assignments), and write the points of the clouds this process owns into the head of the merged batch: coordinates (columns 1..3 of
// [.,4] rows; column 0, the batch index, is constant and written once by the caller) = xyz / quantization_size, features = xyz.
// dest_cloud[c] = position of cloud c among the owned clouds, or -1.
__global__ void k_syn_head(float* __restrict__ xyz, int64_t n_points, int npts, const int* __restrict__ dest_cloud, int do_clamp, float lo, float hi,
                           int zero_first, float quant, float* __restrict__ coords, float* __restrict__ feats) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_points) return;
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float x = xyz[p * 3 + a];
        if (do_clamp) {
            x = x < lo ? lo : x;
            x = x > hi ? hi : x;
        }
        if (zero_first && a == 0) x = 0.f;
        v[a] = x;
        xyz[p * 3 + a] = x;
    }
    const int64_t c = p / npts;
    const int d = dest_cloud ? dest_cloud[c] : (int)c;
    if (d < 0) return;
    const int64_t r = (int64_t)d * npts + (p - c * npts);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        coords[r * 4 + 1 + a] = v[a] / quant;
        feats[r * 3 + a] = v[a];
    }
}

// Epilogue of an iteration in ONE launch: scatter the gradient of the owned clouds into the full-size gradient (zero elsewhere) and
// take the SGD-with-momentum step (optimizer_img.step(), mnist_distill.py:369); same arithmetic as k_sgd_clamp.
__global__ void k_sgd_scatter(float* __restrict__ x, float* __restrict__ mom, const float* __restrict__ dfeats, const int* __restrict__ dest_cloud,
                              int64_t n_points, int npts, float lr, float momentum, float* __restrict__ grad_out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_points) return;
    const int64_t c = p / npts;
    const int d = dest_cloud ? dest_cloud[c] : (int)c;
    const int64_t r = (int64_t)d * npts + (p - c * npts);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float g = d >= 0 ? dfeats[r * 3 + a] : 0.f;
        if (grad_out) grad_out[p * 3 + a] = g;
        if (d < 0) continue;  // another owner's cloud (another rank, another chunk of classes): neither moved nor its momentum touched
        const float m = momentum * mom[p * 3 + a] + g;
        mom[p * 3 + a] = m;
        x[p * 3 + a] = x[p * 3 + a] - lr * m;
    }
}

int build_plan(const dd3d_gseg* segs, int n, int metric, GPlan* p) {
    if (n <= 0 || n > DD3D_MAX_SEG || metric < 0 || metric > 3 || !segs) return DD3D_EARG;
    p->n = n;
    p->metric = metric;
    p->item_off[0] = p->row_off[0] = p->part_off[0] = 0;
    for (int s = 0; s < n; ++s) {
        const dd3d_gseg& g = segs[s];
        if (g.rows <= 0 || g.cols <= 0 || !g.syn || !g.real) return DD3D_EARG;
        int64_t nch, items;
        if (!g.colmajor) {
            nch = ceil_div64(g.cols, GD_ROW_CHUNK);
            items = g.rows * nch;
        } else {
            nch = ceil_div64(g.cols, GD_COL_CHUNK);
            items = ceil_div64(g.rows, 128) * nch;
        }
        if (items > (1 << 24) || g.rows > (1 << 22)) return DD3D_ESIZE;
        p->nchunk[s] = (int)nch;
        p->item_off[s + 1] = p->item_off[s] + (int)items;
        p->row_off[s + 1] = p->row_off[s] + (int)g.rows;
        p->part_off[s + 1] = p->part_off[s] + (int)(g.rows * nch);
        p->seg[s] = g;
    }
    return 0;
}

}  // namespace

static WStride mk_stride(const int64_t* st) { return WStride{st[0], st[1], st[2], st[3]}; }

// strides[4] = element strides of Wl over (group, j, channel, site).  coords = int32 [M,4] rows of the voxels (column 0 = cloud),
// M_bound rows at most; the true count is seg_off[B] (read on the device).  ws: dd3d_sparse_linear_ws_bytes(M_bound, J).
DD3D_API int64_t dd3d_sparse_linear_ws_bytes(int64_t M_bound, int J) { return align_up64((M_bound > 0 ? M_bound : 1) * (int64_t)J * 4, 256); }

DD3D_API int dd3d_sparse_linear_fwd(const float* F, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                                    const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, const float* bias, float* out, void* ws,
                                    int64_t ws_bytes, void* stream) {
    if (B < 0 || C != 128 || J <= 0 || M_bound < 0 || !strides) return DD3D_EARG;
    if (B == 0) return 0;
    if (!F || !site || !coords || !seg_off || !Wl || !out) return DD3D_EARG;
    if (!ws || ws_bytes < dd3d_sparse_linear_ws_bytes(M_bound, J)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (M_bound > 0)
        k_sl_fwd_part<<<dim3((unsigned)(ceil_div64(M_bound, 16) < 2048 ? ceil_div64(M_bound, 16) : 2048)), 256, 0, st>>>(F, site, (const int4*)coords, seg_off + B, cloud_group, Wl, mk_stride(strides),
                                                                             J, (float*)ws);
    k_sl_fwd_sum<<<dim3((unsigned)ceil_div64((int64_t)B * J, 256)), 256, 0, st>>>((const float*)ws, seg_off, B, J, bias, out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_sparse_linear_fwd + dd3d_softmax_ce + dd3d_group_sum_rows(dlogits) in two launches (same bits): logits[B,J], prob[B,J] (nullable),
// dlogits[B,J], cloud_loss[B] (scratch), loss_g[G] (nullable), bias_grad[G,J] (nullable).  counters: G ints, zero before the FIRST call (the
// kernel leaves them zero).  max_per_group >= clouds of the largest group.
DD3D_API int dd3d_classifier_head(const float* F, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                                  const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, const float* bias, const int32_t* labels,
                                  const int32_t* group_cloud_off, int G, int max_per_group, float* logits, float* prob, float* dlogits,
                                  float* cloud_loss, float* loss_g, float* bias_grad, int32_t* counters, void* ws, int64_t ws_bytes, void* stream) {
    if (B < 0 || C != 128 || J <= 0 || M_bound < 0 || !strides || G <= 0 || max_per_group <= 0 || !group_cloud_off || !counters) return DD3D_EARG;
    if (B == 0) return 0;
    if (!F || !site || !coords || !seg_off || !Wl || !logits || !labels || !dlogits || !cloud_loss) return DD3D_EARG;
    if (!ws || ws_bytes < dd3d_sparse_linear_ws_bytes(M_bound, J)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (M_bound > 0)
        k_sl_fwd_part<<<dim3((unsigned)(ceil_div64(M_bound, 16) < 2048 ? ceil_div64(M_bound, 16) : 2048)), 256, 0, st>>>(F, site, (const int4*)coords, seg_off + B, cloud_group, Wl, mk_stride(strides),
                                                                             J, (float*)ws);
    k_cls_head<<<dim3((unsigned)((max_per_group + 63) / 64), (unsigned)G), 256, 0, st>>>((const float*)ws, seg_off, J, bias, labels, group_cloud_off, logits, prob,
                                                                                         dlogits, cloud_loss, loss_g, bias_grad, counters);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_sparse_linear_bwd_x(const float* dl, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                                      const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, int accumulate, float* dF,
                                      void* stream) {
    if (B < 0 || C != 128 || J <= 0 || M_bound < 0 || !strides) return DD3D_EARG;
    if (B == 0 || M_bound == 0) return 0;
    if (!dl || !site || !coords || !seg_off || !Wl || !dF) return DD3D_EARG;
    k_sl_bwd_x<<<dim3((unsigned)(ceil_div64(M_bound, 8) < 4096 ? ceil_div64(M_bound, 8) : 4096)), 256, 0, (hipStream_t)stream>>>(dl, site, (const int4*)coords, seg_off + B, cloud_group, Wl,
                                                                                       mk_stride(strides), J, accumulate, dF);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dWl must hold G * strides[0] floats (or J*C*S when G == 1); it is zero-filled here, then the occupied sites are written.
DD3D_API int dd3d_sparse_linear_bwd_w(const float* dl, const float* F, const int32_t* vox_at, int S, int B, const int32_t* group_cloud_off, int G, int C,
                                      int J, float* dWl, const int64_t* strides, int64_t dWl_elems, void* stream) {
    if (G <= 0 || B < 0 || C != 128 || J <= 0 || S <= 0 || !strides || !dWl || !group_cloud_off || dWl_elems < (int64_t)J * C * S) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    // channels innermost (the step's [G, J, S, C] layout): the kernel writes every (group, output, site) row itself, zeros included;
    // otherwise (the parameter's own [J, C * S]: 4-byte stores a site count apart) empty sites are skipped over a zero-filled buffer
    const int zero_empty = strides[2] == 1 && B > 0 && dl && F && vox_at && dWl_elems == (int64_t)G * J * C * S;
    if (!zero_empty) DD3D_HIP(hipMemsetAsync(dWl, 0, dWl_elems * 4, st));
    if (!dl || !F || !vox_at) return DD3D_EARG;
    if (B > 0) k_sl_bwd_w<<<dim3(S, G), 256, 0, st>>>(dl, F, vox_at, S, B, group_cloud_off, J, dWl, mk_stride(strides), zero_empty);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// cloud_loss[B] is scratch; loss_g[G] (may be null) receives the mean CE of every group
// out[g][s][j][c] = in[g][j][c*S+s] (inverse = 0) or the other way round (inverse = 1); G groups of [J, C*S]
DD3D_API int dd3d_classifier_layout(const float* in, int G, int J, int C, int S, int inverse, float* out, void* stream) {
    if (G <= 0 || J <= 0 || C <= 0 || S <= 0 || !in || !out || (int64_t)G * J > 65535) return DD3D_EARG;
    k_classifier_layout<<<dim3((S + 31) / 32, (C + 31) / 32, G * J), 256, 0, (hipStream_t)stream>>>(in, J, C, S, inverse, out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_softmax_ce(const float* logits, const int32_t* labels, const int32_t* cloud_group, const int32_t* group_cloud_off, int G, int B, int J,
                             float* prob, float* dlogits, float* cloud_loss, float* loss_g, void* stream) {
    if (G <= 0 || J <= 0 || B < 0 || !group_cloud_off) return DD3D_EARG;
    if (B == 0) return 0;
    if (!logits || !labels || !dlogits || !cloud_loss) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    k_softmax_ce<<<dim3((B + 63) / 64), 64, 0, st>>>(logits, labels, cloud_group, group_cloud_off, B, J, prob, dlogits, cloud_loss);
    if (loss_g) k_group_sum<<<dim3(G), 64, 0, st>>>(cloud_loss, group_cloud_off, G, loss_g);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_group_sum_rows(const float* v, const int32_t* group_cloud_off, int G, int J, float* out, void* stream) {
    if (G <= 0 || J <= 0 || !v || !group_cloud_off || !out) return DD3D_EARG;
    k_group_sum_rows<<<dim3(G * J), 64, 0, (hipStream_t)stream>>>(v, group_cloud_off, G, J, out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// per_class[G] (nullable) = a + b + c + d elementwise, left to right (b, c, d nullable), *total (nullable) = their sum over the groups; one launch.
DD3D_API int dd3d_loss_total(const float* a, const float* b, const float* c, const float* d, int G, float* per_class, float* total, void* stream) {
    if (G < 0 || (G > 0 && !a)) return DD3D_EARG;
    k_loss_total<<<1, 64, 0, (hipStream_t)stream>>>(a, b, c, d, G, per_class, total);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_softmax_ce_bwd2(const float* prob, const float* u, const float* u_group, const int32_t* group_cloud_off, int G, int J,
                                  int max_clouds_per_group, float* dz, void* stream) {
    if (G <= 0 || J <= 0 || max_clouds_per_group <= 0 || !prob || !u || !group_cloud_off || !dz) return DD3D_EARG;
    k_softmax_ce_bwd2<<<dim3((max_clouds_per_group + 63) / 64, G), 64, 0, (hipStream_t)stream>>>(prob, u, u_group, group_cloud_off, G, J, dz);
    DD3D_CHECK_LAUNCH();
    return 0;
}

constexpr int GD_SMALL_MIN_GROUPS = 8;  // one workgroup per group: with fewer groups the three launches spread the same work wider (measured on
                                        // one-, two- and five-class shards: 0.357 / 0.431 / 0.699 ms per step against 0.365 / 0.436 / 0.705 in
                                        // one launch; with ten groups the two forms take the same time)
static bool plan_is_small(const GPlan& p, int G) {
    static const bool off = getenv("DD3D_GD_SMALL") && !atoi(getenv("DD3D_GD_SMALL"));  // 0: always the three launches (measurements, tests)
    if (off || G < GD_SMALL_MIN_GROUPS || p.row_off[p.n] > 256) return false;
    int64_t passes = 0;
    for (int s = 0; s < p.n; ++s) {
        const dd3d_gseg& g = p.seg[s];
        if (g.colmajor) passes += ceil_div64(g.rows, 128) * p.nchunk[s];
        else if (p.nchunk[s] != 1 || g.cols > GD_SMALL_COLS) return false;
        else passes += g.rows;
    }
    return passes <= GD_SMALL_PASSES;
}

DD3D_API int64_t dd3d_match_loss_ws_bytes(const dd3d_gseg* segs, int n, int G) {
    GPlan p;
    if (G <= 0 || build_plan(segs, n, 0, &p)) return -1;
    return align_up64((int64_t)G * p.part_off[n] * 16, 256) + 256;
}

DD3D_API int64_t dd3d_match_loss_stats_floats(const dd3d_gseg* segs, int n, int G) {
    GPlan p;
    if (G <= 0 || build_plan(segs, n, 0, &p)) return -1;
    return (int64_t)G * ((int64_t)p.row_off[n] + 1) * 4;
}

// metric: 0 'ours' (distance_wb), 1 'modmse' (distance_mse), 2 'mse', 3 'cos'.  segs is a HOST array.
// G >= 1 independent groups: every segment tensor is [G, rows*cols] contiguous, loss_dev[G], one stats block per group.
DD3D_API int dd3d_match_loss_fwd(const dd3d_gseg* segs, int n, int G, int metric, float* loss_dev, float* stats_dev, void* ws, int64_t ws_bytes,
                                 void* stream) {
    GPlan p;
    int rc = build_plan(segs, n, metric, &p);
    if (rc) return rc;
    if (!loss_dev || !stats_dev || G <= 0 || G > 65535) return DD3D_EARG;
    if (!ws || ws_bytes < dd3d_match_loss_ws_bytes(segs, n, G)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (plan_is_small(p, G)) {  // one launch, the same bits
        k_gd_small<<<G, 256, 0, st>>>(p, (float4*)stats_dev, loss_dev, nullptr, 0);
        DD3D_CHECK_LAUNCH();
        return 0;
    }
    float4* partial = (float4*)ws;
    k_gd_stats<<<dim3(p.item_off[n], G), 256, 0, st>>>(p, partial);
    k_gd_finalize<<<G, 256, 0, st>>>(p, partial, (float4*)stats_dev, loss_dev);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_match_loss_fwd followed by dd3d_match_loss_bwd (same outputs, bit for bit): ONE launch for small lists (see k_gd_small), the three
// launches otherwise.
DD3D_API int dd3d_match_loss_fwd_bwd(const dd3d_gseg* segs, int n, int G, int metric, float* loss_dev, float* stats_dev, const float* gl_dev, void* ws,
                                     int64_t ws_bytes, void* stream) {
    GPlan p;
    int rc = build_plan(segs, n, metric, &p);
    if (rc) return rc;
    if (!loss_dev || !stats_dev || G <= 0 || G > 65535) return DD3D_EARG;
    if (plan_is_small(p, G)) {
        k_gd_small<<<G, 256, 0, (hipStream_t)stream>>>(p, (float4*)stats_dev, loss_dev, gl_dev, 1);
        DD3D_CHECK_LAUNCH();
        return 0;
    }
    rc = dd3d_match_loss_fwd(segs, n, G, metric, loss_dev, stats_dev, ws, ws_bytes, stream);
    if (rc) return rc;
    return dd3d_match_loss_bwd(segs, n, G, metric, stats_dev, gl_dev, ws, ws_bytes, stream);
}

// writes seg[i].dsyn (where non-null) = gl * d loss / d seg[i].syn ; gl_dev may be null (= 1)
DD3D_API int dd3d_match_loss_bwd(const dd3d_gseg* segs, int n, int G, int metric, const float* stats_dev, const float* gl_dev, void* ws,
                                 int64_t ws_bytes, void* stream) {
    GPlan p;
    int rc = build_plan(segs, n, metric, &p);
    if (rc) return rc;
    if (!stats_dev || G <= 0 || G > 65535) return DD3D_EARG;
    (void)ws; (void)ws_bytes;
    hipStream_t st = (hipStream_t)stream;
    k_gd_bwd<<<dim3(p.item_off[n], G), 256, 0, st>>>(p, (const float4*)stats_dev, gl_dev);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_sgd_momentum_clamp(float* x, const float* grad, float* mom, int64_t n, float lr, float momentum, int first_step, int do_clamp, float lo,
                                     float hi, void* stream) {
    if (n < 0) return DD3D_EARG;
    if (n == 0) return 0;
    if (!x || !grad || !mom) return DD3D_EARG;
    k_sgd_clamp<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(x, grad, mom, n, lr, momentum, first_step, do_clamp, lo, hi);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_syn_head(float* xyz, int64_t n_clouds, int npts, const int32_t* dest_cloud, int do_clamp, float lo, float hi, int zero_first,
                           float quant, float* coords, float* feats, void* stream) {
    if (n_clouds < 0 || npts <= 0 || quant == 0.f) return DD3D_EARG;
    const int64_t n = n_clouds * npts;
    if (n == 0) return 0;
    if (!xyz || !coords || !feats) return DD3D_EARG;
    k_syn_head<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(xyz, n, npts, dest_cloud, do_clamp, lo, hi, zero_first, quant, coords,
                                                                                    feats);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_sgd_scatter(float* xyz, float* mom, const float* dfeats, const int32_t* dest_cloud, int64_t n_clouds, int npts, float lr,
                              float momentum, float* grad_out, void* stream) {
    if (n_clouds < 0 || npts <= 0) return DD3D_EARG;
    const int64_t n = n_clouds * npts;
    if (n == 0) return 0;
    if (!xyz || !mom || !dfeats) return DD3D_EARG;
    k_sgd_scatter<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(xyz, mom, dfeats, dest_cloud, n, npts, lr, momentum, grad_out);
    DD3D_CHECK_LAUNCH();
    return 0;
}
