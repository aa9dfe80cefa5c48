// C++/OpenMP restatement of ONE CLASS of the gradient-matching step of MEConvExp (the CPU leg).
// TEST INFRASTRUCTURE AND CPU BASELINE ONLY: nothing under dd3d_amd/ loads this library; tests/ use it as a second, independent checker
// (it shares no code with oracle/minkowski_ref.py, which pins it) and bench.py's cpu_baseline leg times it on the host cores.
//
// What it follows (paths relative to /root/reference; MinkowskiEngine 0.5.4 semantics as in SURVEY.md Appendix A -- parity unpinned at
// that boundary, see oracle/minkowski_ref.py):
//   mnist_distill.py:336-360   real forward/backward -> gw_real, synthetic forward -> gw_syn (create_graph), match_loss
//   mnist_distill.py:365       loss.backward() into the synthetic points (only the torch-native pieces are differentiated twice:
//                              the instance-norm affine, dense()/Linear/cross entropy -- MinkowskiEngine's backward passes are opaque)
//   models/MEConv.py:221-256   (conv k3 -> instance norm -> ReLU -> max pool 2/2) x 2 -> dense() -> Linear
//   distillation_loss.py:26-130 distance_wb / distance_mse / match_loss (settings.Pyt = False)
// Templated on the arithmetic type: double for checking, float for the timed baseline.
#include <omp.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "dd3d_cpu.h"  // the exported interface (dd3d_cpu_net: [27,3,128] [128] [128] [27,128,128] [128] [128] [J,128*S] [J], order of parameters())

#define CPU_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int C = 128;

// DD3D_CPU_PROF=1: seconds per op family, printed by dd3d_cpu_prof_dump()
struct Prof {
    double t[16] = {0};
    const char* name[16] = {"maps", "conv_fwd", "conv_bwd_data", "conv_bwd_weight", "inorm", "pool", "linear", "loss", "other"};
} g_prof;
struct Tick {
    int i;
    double t0;
    explicit Tick(int i_) : i(i_), t0(omp_get_wtime()) {}
    ~Tick() { g_prof.t[i] += omp_get_wtime() - t0; }
};

inline int floor_div(int a, int s) {
    int q = a / s;
    if ((a % s != 0) && ((a < 0) != (s < 0))) --q;
    return q;
}

struct Key {
    static uint64_t pack(int b, int x, int y, int z) {  // 16 bits each, spatial coordinates offset into [0, 65536)
        return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(x + 32768) << 32) | ((uint64_t)(uint16_t)(y + 32768) << 16) | (uint64_t)(uint16_t)(z + 32768);
    }
};

struct Level {
    int M = 0;
    std::vector<int> Cd;  // [M][4]
    std::unordered_map<uint64_t, int> tab;
    std::vector<int> seg;  // [B+1] rows of every cloud (clouds ascend, rows in first-occurrence order: contiguous per cloud)
    int find(int b, int x, int y, int z) const {
        auto it = tab.find(Key::pack(b, x, y, z));
        return it == tab.end() ? -1 : it->second;
    }
};

// unique rows in first-occurrence order (MinkowskiEngine CPU insertion order); map[i] = row of input i
void unique_rows(const std::vector<int>& in, int n, Level& lv, std::vector<int>& map, int B) {
    lv.tab.reserve((size_t)n * 2);
    map.resize(n);
    lv.Cd.clear();
    for (int i = 0; i < n; ++i) {
        const int* c = &in[(size_t)i * 4];
        auto r = lv.tab.emplace(Key::pack(c[0], c[1], c[2], c[3]), lv.M);
        if (r.second) {
            lv.Cd.insert(lv.Cd.end(), c, c + 4);
            ++lv.M;
        }
        map[i] = r.first->second;
    }
    lv.seg.assign(B + 1, 0);
    for (int v = 0; v < lv.M; ++v) lv.seg[lv.Cd[(size_t)v * 4] + 1]++;
    for (int b = 0; b < B; ++b) lv.seg[b + 1] += lv.seg[b];
}

struct Maps {
    int B = 0, N = 0, S = 0;
    Level L0, L1, L2;
    std::vector<int> inv, cnt0, par1, par2, nbr0, nbr1, kid1, kid2, site;
    bool ok = true, range_ok = true;
};

// k = (d0+1) + 3 (d1+1) + 9 (d2+1), d0 = first spatial column; table [27][M]
void kernel_map3(const Level& lv, int ts, std::vector<int>& nbr) {
    const int M = lv.M;
    nbr.assign((size_t)27 * M, -1);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < M; ++o) {
        const int* c = &lv.Cd[(size_t)o * 4];
        for (int k = 0; k < 27; ++k) {
            const int d0 = (k % 3 - 1) * ts, d1 = ((k / 3) % 3 - 1) * ts, d2 = (k / 9 - 1) * ts;
            nbr[(size_t)k * M + o] = lv.find(c[0], c[1] + d0, c[2] + d1, c[3] + d2);
        }
    }
}

// children of every stride-2ts voxel at offsets {0,1} * ts, k = e0 + 2 e1 + 4 e2; table [8][M_out]
void pool_map(const Level& in, const Level& out, int ts, std::vector<int>& kid) {
    const int M = out.M;
    kid.assign((size_t)8 * M, -1);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < M; ++o) {
        const int* c = &out.Cd[(size_t)o * 4];
        for (int k = 0; k < 8; ++k) kid[(size_t)k * M + o] = in.find(c[0], c[1] + (k & 1) * ts, c[2] + ((k >> 1) & 1) * ts, c[3] + (k >> 2) * ts);
    }
}

void stride_up(const Level& in, int s, Level& out, std::vector<int>& parent, int B) {
    std::vector<int> q((size_t)in.M * 4);
    for (int v = 0; v < in.M; ++v) {
        q[(size_t)v * 4] = in.Cd[(size_t)v * 4];
        for (int a = 1; a < 4; ++a) q[(size_t)v * 4 + a] = floor_div(in.Cd[(size_t)v * 4 + a], s) * s;
    }
    unique_rows(q, in.M, out, parent, B);
}

// coords: [N][3] spatial coordinates already divided by the quantisation size (models/MEConv.py:262), off[B+1]: points of every cloud
void build_maps(const double* coords, const int* off, int B, const dd3d_cpu_net& net, Maps& m) {
    Tick tick_(0);
    m.B = B;
    m.N = off[B];
    std::vector<int> q((size_t)m.N * 4);
    for (int b = 0; b < B; ++b)
        for (int i = off[b]; i < off[b + 1]; ++i) {
            q[(size_t)i * 4] = b;
            for (int a = 0; a < 3; ++a) {
                const double f = std::floor(coords[(size_t)i * 3 + a]);
                if (!(f > -32000.0 && f < 32000.0) || B > 65535) m.range_ok = false;  // the packed keys hold 16 bits per coordinate
                q[(size_t)i * 4 + 1 + a] = (int)f;
            }
        }
    if (!m.range_ok) {
        m.ok = false;
        return;
    }
    unique_rows(q, m.N, m.L0, m.inv, B);
    m.cnt0.assign(m.L0.M, 0);
    for (int i = 0; i < m.N; ++i) m.cnt0[m.inv[i]]++;
    stride_up(m.L0, 2, m.L1, m.par1, B);
    stride_up(m.L1, 4, m.L2, m.par2, B);
    kernel_map3(m.L0, 1, m.nbr0);
    kernel_map3(m.L1, 2, m.nbr1);
    pool_map(m.L0, m.L1, 1, m.kid1);
    pool_map(m.L1, m.L2, 2, m.kid2);
    // dense(): flat site relative to the minimum stride-4 coordinate of the WHOLE batch (models/MEConv.py:251-254)
    int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
    for (int v = 0; v < m.L2.M; ++v)
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], m.L2.Cd[(size_t)v * 4 + 1 + a]);
    m.S = net.E0 * net.E1 * net.E2;
    m.site.resize(m.L2.M);
    for (int v = 0; v < m.L2.M; ++v) {
        const int* c = &m.L2.Cd[(size_t)v * 4];
        const int a = (c[1] - mn[0]) / 4, d = (c[2] - mn[1]) / 4, e = (c[3] - mn[2]) / 4;
        if (a < 0 || a >= net.E0 || d < 0 || d >= net.E1 || e < 0 || e >= net.E2) {
            m.ok = false;  // MinkowskiEngine's dense() raises
            m.site[v] = 0;
        } else {
            m.site[v] = (a * net.E1 + d) * net.E2 + e;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ feature ops
template <class T>
using Vec = std::vector<T>;

// Y[o] = sum_k X[nbr[k][o]] . W[k].  Register-blocked: four output rows x CB output channels of accumulators stay in vector registers
// while an offset's weight rows stream past once for the four rows (the rows of a tile are neighbours in space, so they mostly share
// their offsets); Co not a multiple of the block (the 3-channel input gradient of the first layer): a plain loop.
template <class T, int CB>
__attribute__((target_clones("avx512f", "avx2", "default"))) void conv_fwd_blocked(const T* X, int Ci, const T* W, int Co, const int* nbr, int M, T* Y) {
    constexpr int RT = 4;
#pragma omp parallel
    {
#pragma omp for schedule(dynamic, 16)
    for (int o0 = 0; o0 < M; o0 += RT) {
        const int nr = std::min(RT, M - o0);
        for (int c0 = 0; c0 < Co; c0 += CB) {
            T acc[RT][CB];
            for (int r = 0; r < RT; ++r)
                for (int c = 0; c < CB; ++c) acc[r][c] = 0;
            for (int k = 0; k < 27; ++k) {
                const T* xr[RT];
                bool any = false;
                for (int r = 0; r < RT; ++r) {
                    const int i = r < nr ? nbr[(size_t)k * M + o0 + r] : -1;
                    xr[r] = i >= 0 ? X + (size_t)i * Ci : nullptr;
                    any |= i >= 0;
                }
                if (!any) continue;
                const T* w = W + (size_t)k * Ci * Co + c0;
                for (int a = 0; a < Ci; ++a) {
                    const T* wr = w + (size_t)a * Co;
                    for (int r = 0; r < RT; ++r) {
                        if (!xr[r]) continue;
                        const T xv = xr[r][a];
#pragma omp simd
                        for (int c = 0; c < CB; ++c) acc[r][c] += xv * wr[c];
                    }
                }
            }
            for (int r = 0; r < nr; ++r)
                for (int c = 0; c < CB; ++c) Y[(size_t)(o0 + r) * Co + c0 + c] = acc[r][c];
        }
    }
    }
}

template <class T>
void conv_fwd(const T* X, int Ci, const T* W, int Co, const int* nbr, int M, T* Y) {
    Tick tick_(1);
    constexpr int CB = 256 / (int)sizeof(T);  // 64 floats / 32 doubles: 4 rows x CB = 16 zmm registers
    if (Co % CB == 0) return conv_fwd_blocked<T, CB>(X, Ci, W, Co, nbr, M, Y);
#pragma omp parallel for schedule(dynamic, 64)
    for (int o = 0; o < M; ++o) {
        T* y = Y + (size_t)o * Co;
        for (int c = 0; c < Co; ++c) y[c] = 0;
        for (int k = 0; k < 27; ++k) {
            const int i = nbr[(size_t)k * M + o];
            if (i < 0) continue;
            const T* x = X + (size_t)i * Ci;
            const T* w = W + (size_t)k * Ci * Co;
            for (int a = 0; a < Ci; ++a)
                for (int c = 0; c < Co; ++c) y[c] += x[a] * w[(size_t)a * Co + c];
        }
    }
}

// input gradient of a stride-1 convolution: dX[i] = sum_k g[nbr[26-k][i]] . W[k]^T (the offsets of an odd kernel are centrally symmetric:
// nbr[k][o] == i  <=>  nbr[26-k][i] == o), i.e. the forward convolution of g with the mirrored, transposed kernel Wm[k] = W[26-k]^T
template <class T>
void conv_bwd_data(const T* g, int Co, const T* W, int Ci, const int* nbr, int M, T* dX) {
    Tick tick_(2);
    Vec<T> Wm((size_t)27 * Co * Ci);
#pragma omp parallel for schedule(static) collapse(2)
    for (int k = 0; k < 27; ++k)
        for (int c = 0; c < Co; ++c) {
            T* dst = &Wm[((size_t)(26 - k) * Co + c) * Ci];
            const T* src = W + (size_t)k * Ci * Co + c;
            for (int a = 0; a < Ci; ++a) dst[a] = src[(size_t)a * Co];
        }
    conv_fwd(g, Co, Wm.data(), Ci, nbr, M, dX);
}

// dW[k][a][c] = sum_o X[nbr[k][o]][a] g[o][c].  The pairs of every offset are compacted first; a thread per (k, block of 16 input channels)
// then walks them four at a time (four products per load/store of the 8 KB block of dW it keeps in L1), pairs in row order -> deterministic
template <class T>
__attribute__((target_clones("avx512f", "avx2", "default"))) void conv_bwd_weight_units(const T* X, int Ci, const T* g, int Co, const int* pin, const int* pout,
                                                                                       const int* poff, T* dW) {
    const int AB = Ci >= 16 ? 16 : Ci, nab = (Ci + AB - 1) / AB;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int k = 0; k < 27; ++k)
        for (int ab = 0; ab < nab; ++ab) {
            const int a0 = ab * AB, a1 = std::min(a0 + AB, Ci);
            T* w = dW + (size_t)k * Ci * Co;
            for (int a = a0; a < a1; ++a)
                for (int c = 0; c < Co; ++c) w[(size_t)a * Co + c] = 0;
            int p = poff[k];
            const int p1 = poff[k + 1];
            for (; p + 4 <= p1; p += 4) {
                const T *x0 = X + (size_t)pin[p] * Ci, *x1 = X + (size_t)pin[p + 1] * Ci, *x2 = X + (size_t)pin[p + 2] * Ci, *x3 = X + (size_t)pin[p + 3] * Ci;
                const T* __restrict__ g0 = g + (size_t)pout[p] * Co;
                const T* __restrict__ g1 = g + (size_t)pout[p + 1] * Co;
                const T* __restrict__ g2 = g + (size_t)pout[p + 2] * Co;
                const T* __restrict__ g3 = g + (size_t)pout[p + 3] * Co;
                for (int a = a0; a < a1; ++a) {
                    const T v0 = x0[a], v1 = x1[a], v2 = x2[a], v3 = x3[a];
                    T* __restrict__ wr = w + (size_t)a * Co;
#pragma omp simd
                    for (int c = 0; c < Co; ++c) wr[c] += (v0 * g0[c] + v1 * g1[c]) + (v2 * g2[c] + v3 * g3[c]);
                }
            }
            for (; p < p1; ++p) {
                const T* x = X + (size_t)pin[p] * Ci;
                const T* __restrict__ gr = g + (size_t)pout[p] * Co;
                for (int a = a0; a < a1; ++a) {
                    const T xv = x[a];
                    T* __restrict__ wr = w + (size_t)a * Co;
#pragma omp simd
                    for (int c = 0; c < Co; ++c) wr[c] += xv * gr[c];
                }
            }
        }
}

template <class T>
void conv_bwd_weight(const T* X, int Ci, const T* g, int Co, const int* nbr, int M, T* dW) {
    Tick tick_(3);
    std::vector<int> pin, pout, poff(28, 0);
    pin.reserve((size_t)M * 8);
    pout.reserve((size_t)M * 8);
    for (int k = 0; k < 27; ++k) {
        for (int o = 0; o < M; ++o) {
            const int i = nbr[(size_t)k * M + o];
            if (i >= 0) {
                pin.push_back(i);
                pout.push_back(o);
            }
        }
        poff[k + 1] = (int)pin.size();
    }
    conv_bwd_weight_units(X, Ci, g, Co, pin.data(), pout.data(), poff.data(), dW);
}

// instance norm per cloud and channel: biased variance, eps = 1e-8 inside the sqrt
template <class T>
void inorm_fwd(const T* Y, const std::vector<int>& seg, int B, T* xhat, T* inv_std) {
    Tick tick_(4);
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
        const int r0 = seg[b], r1 = seg[b + 1], n = std::max(r1 - r0, 1);
        T mean[C], var[C];
        for (int c = 0; c < C; ++c) mean[c] = var[c] = 0;
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < C; ++c) mean[c] += Y[(size_t)r * C + c];
        for (int c = 0; c < C; ++c) mean[c] /= (T)n;
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < C; ++c) {
                const T d = Y[(size_t)r * C + c] - mean[c];
                var[c] += d * d;
            }
        for (int c = 0; c < C; ++c) inv_std[(size_t)b * C + c] = (T)1 / std::sqrt(var[c] / (T)n + (T)1e-8);
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < C; ++c) xhat[(size_t)r * C + c] = (Y[(size_t)r * C + c] - mean[c]) * inv_std[(size_t)b * C + c];
    }
}

template <class T>
void inorm_bwd(const T* g, const T* xhat, const T* inv_std, const std::vector<int>& seg, int B, T* dY) {
    Tick tick_(4);
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
        const int r0 = seg[b], r1 = seg[b + 1], n = std::max(r1 - r0, 1);
        T m1[C], m2[C];
        for (int c = 0; c < C; ++c) m1[c] = m2[c] = 0;
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < C; ++c) {
                m1[c] += g[(size_t)r * C + c];
                m2[c] += g[(size_t)r * C + c] * xhat[(size_t)r * C + c];
            }
        for (int c = 0; c < C; ++c) {
            m1[c] /= (T)n;
            m2[c] /= (T)n;
        }
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < C; ++c)
                dY[(size_t)r * C + c] = (g[(size_t)r * C + c] - xhat[(size_t)r * C + c] * m2[c] - m1[c]) * inv_std[(size_t)b * C + c];
    }
}

// relu(xhat * w + b) then max over the existing children (strict >: the first maximum in offset order wins); arg = winning child row
template <class T>
void affine_relu_pool_fwd(const T* xhat, const T* w, const T* bias, const int* kid, int M_out, T* out, int* arg) {
    Tick tick_(5);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < M_out; ++o)
        for (int c = 0; c < C; ++c) {
            T best = -INFINITY;
            int a = -1;
            for (int k = 0; k < 8; ++k) {
                const int i = kid[(size_t)k * M_out + o];
                if (i < 0) continue;
                T v = xhat[(size_t)i * C + c] * w[c] + bias[c];
                v = v > 0 ? v : 0;
                if (v > best) {
                    best = v;
                    a = i;
                }
            }
            out[(size_t)o * C + c] = best;
            arg[(size_t)o * C + c] = a;
        }
}

// backward of pool -> ReLU -> affine for every input row: ga = the ReLU-masked pooled gradient; g_xhat = ga * w (+ ga_prev * extra)
template <class T>
void affine_relu_pool_bwd(const T* g_pool, const int* arg, const int* parent, const T* xhat, const T* w, const T* bias, int M_in, T* ga, T* g_xhat) {
    Tick tick_(5);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M_in; ++i) {
        const int p = parent[i];
        for (int c = 0; c < C; ++c) {
            T g = arg[(size_t)p * C + c] == i ? g_pool[(size_t)p * C + c] : (T)0;
            if (!(xhat[(size_t)i * C + c] * w[c] + bias[c] > 0)) g = 0;
            ga[(size_t)i * C + c] = g;
            g_xhat[(size_t)i * C + c] = g * w[c];
        }
    }
}

template <class T>
void colsum2(const T* ga, const T* xhat, int M, T* gw, T* gb) {  // gw = sum ga * xhat, gb = sum ga, rows in order
    for (int c = 0; c < C; ++c) gw[c] = gb[c] = 0;
    for (int i = 0; i < M; ++i)
        for (int c = 0; c < C; ++c) {
            gw[c] += ga[(size_t)i * C + c] * xhat[(size_t)i * C + c];
            gb[c] += ga[(size_t)i * C + c];
        }
}

// dense(P).view(B,-1) @ Wc^T (+ bias) without the dense tensor: logits[b][j] = sum_{v in b} sum_c P[v][c] Wc[j][c S + site(v)]
template <class T>
void linear_fwd(const T* P, const Maps& m, const T* Wc, const T* bias, int J, T* out) {
    Tick tick_(6);
    const int S = m.S;
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < m.B; ++b)
        for (int j = 0; j < J; ++j) {
            T s = 0;
            for (int v = m.L2.seg[b]; v < m.L2.seg[b + 1]; ++v) {
                const T* w = Wc + (size_t)j * C * S + m.site[v];
                T t = 0;
                for (int c = 0; c < C; ++c) t += P[(size_t)v * C + c] * w[(size_t)c * S];
                s += t;
            }
            out[(size_t)b * J + j] = bias ? s + bias[j] : s;
        }
}

// dP[v][c] (+)= sum_j dl[b(v)][j] Wc[j][c S + site(v)]
template <class T>
void linear_bwd_x(const T* dl, const Maps& m, const T* Wc, int J, bool accumulate, T* dP) {
    Tick tick_(6);
    const int S = m.S;
#pragma omp parallel for schedule(dynamic, 16)
    for (int v = 0; v < m.L2.M; ++v) {
        const int b = m.L2.Cd[(size_t)v * 4];
        for (int c = 0; c < C; ++c) {
            T s = 0;
            for (int j = 0; j < J; ++j) s += dl[(size_t)b * J + j] * Wc[(size_t)j * C * S + (size_t)c * S + m.site[v]];
            dP[(size_t)v * C + c] = accumulate ? dP[(size_t)v * C + c] + s : s;
        }
    }
}

// gWc[j][c S + s] = sum over the voxels v at site s of dl[b(v)][j] P[v][c]  (voxels in row order: deterministic)
template <class T>
void linear_bwd_w(const T* dl, const T* P, const Maps& m, int J, T* gW) {
    Tick tick_(6);
    const int S = m.S;
    std::memset(gW, 0, sizeof(T) * (size_t)J * C * S);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < J; ++j)
        for (int v = 0; v < m.L2.M; ++v) {
            const T d = dl[(size_t)m.L2.Cd[(size_t)v * 4] * J + j];
            T* w = gW + (size_t)j * C * S + m.site[v];
            for (int c = 0; c < C; ++c) w[(size_t)c * S] += d * P[(size_t)v * C + c];
        }
}

// ------------------------------------------------------------------------------------------------------------------ one pass
template <class T>
struct Pass {
    Maps m;
    Vec<T> F0, Y0, xh0, is0, P0, Y1, xh1, is1, P1, logits, prob, dl, ga0, ga1;
    Vec<int> arg0, arg1;
    Vec<T> gw[8];  // order of network.parameters(): W0, w1, b1, W1, w5, b5, Wc, bc
    Vec<T> gP1, gx1, gY1, gP0, gx0, gY0;  // scratch of the first-order backward (kept between calls: see class_step)
};

template <class T>
struct Net {
    int J, S;
    const T *W0, *w1, *b1, *W1, *w5, *b5, *Wc, *bc;
};

// forward + first-order backward of the mean cross entropy w.r.t. the parameters (mnist_distill.py:343-349 / :354-359)
template <class T>
bool forward_backward(const Net<T>& n, const dd3d_cpu_net& cn, const double* coords, const T* feats, const int* off, int B, int label, Pass<T>& p) {
    Maps& m = p.m;
    build_maps(coords, off, B, cn, m);
    if (!m.ok) return false;
    const int M0 = m.L0.M, M1 = m.L1.M, M2 = m.L2.M, J = n.J;
    // TensorField.sparse(): mean of the features of the points of every voxel, points in order
    p.F0.assign((size_t)M0 * 3, 0);
    for (int i = 0; i < m.N; ++i)
        for (int a = 0; a < 3; ++a) p.F0[(size_t)m.inv[i] * 3 + a] += feats[(size_t)i * 3 + a];
    for (int v = 0; v < M0; ++v)
        for (int a = 0; a < 3; ++a) p.F0[(size_t)v * 3 + a] /= (T)m.cnt0[v];
    p.Y0.resize((size_t)M0 * C); p.xh0.resize((size_t)M0 * C); p.is0.resize((size_t)B * C);
    conv_fwd(p.F0.data(), 3, n.W0, C, m.nbr0.data(), M0, p.Y0.data());
    inorm_fwd(p.Y0.data(), m.L0.seg, B, p.xh0.data(), p.is0.data());
    p.P0.resize((size_t)M1 * C); p.arg0.resize((size_t)M1 * C);
    affine_relu_pool_fwd(p.xh0.data(), n.w1, n.b1, m.kid1.data(), M1, p.P0.data(), p.arg0.data());
    p.Y1.resize((size_t)M1 * C); p.xh1.resize((size_t)M1 * C); p.is1.resize((size_t)B * C);
    conv_fwd(p.P0.data(), C, n.W1, C, m.nbr1.data(), M1, p.Y1.data());
    inorm_fwd(p.Y1.data(), m.L1.seg, B, p.xh1.data(), p.is1.data());
    p.P1.resize((size_t)M2 * C); p.arg1.resize((size_t)M2 * C);
    affine_relu_pool_fwd(p.xh1.data(), n.w5, n.b5, m.kid2.data(), M2, p.P1.data(), p.arg1.data());
    p.logits.resize((size_t)B * J); p.prob.resize((size_t)B * J); p.dl.resize((size_t)B * J);
    linear_fwd(p.P1.data(), m, n.Wc, n.bc, J, p.logits.data());
    for (int b = 0; b < B; ++b) {  // mean cross entropy: dlogits = (softmax - onehot) / B
        const T* z = &p.logits[(size_t)b * J];
        T mx = z[0];
        for (int j = 1; j < J; ++j) mx = std::max(mx, z[j]);
        T se = 0;
        for (int j = 0; j < J; ++j) se += std::exp(z[j] - mx);
        const T lse = mx + std::log(se);
        for (int j = 0; j < J; ++j) {
            const T pr = std::exp(z[j] - lse);
            p.prob[(size_t)b * J + j] = pr;
            p.dl[(size_t)b * J + j] = (pr - (j == label ? (T)1 : (T)0)) / (T)B;
        }
    }
    // ---- backward
    p.gw[6].resize((size_t)J * C * m.S); p.gw[7].assign(J, 0);
    linear_bwd_w(p.dl.data(), p.P1.data(), m, J, p.gw[6].data());
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < J; ++j) p.gw[7][j] += p.dl[(size_t)b * J + j];
    Vec<T>&gP1 = p.gP1, &gx1 = p.gx1, &gY1 = p.gY1, &gP0 = p.gP0, &gx0 = p.gx0, &gY0 = p.gY0;  // (every element is written before it is read)
    gP1.resize((size_t)M2 * C); gx1.resize((size_t)M1 * C); gY1.resize((size_t)M1 * C); gP0.resize((size_t)M1 * C);
    gx0.resize((size_t)M0 * C); gY0.resize((size_t)M0 * C);
    linear_bwd_x(p.dl.data(), m, n.Wc, J, false, gP1.data());
    p.ga1.resize((size_t)M1 * C);
    affine_relu_pool_bwd(gP1.data(), p.arg1.data(), m.par2.data(), p.xh1.data(), n.w5, n.b5, M1, p.ga1.data(), gx1.data());
    p.gw[4].resize(C); p.gw[5].resize(C);
    colsum2(p.ga1.data(), p.xh1.data(), M1, p.gw[4].data(), p.gw[5].data());
    inorm_bwd(gx1.data(), p.xh1.data(), p.is1.data(), m.L1.seg, B, gY1.data());
    p.gw[3].resize((size_t)27 * C * C);
    conv_bwd_weight(p.P0.data(), C, gY1.data(), C, m.nbr1.data(), M1, p.gw[3].data());
    conv_bwd_data(gY1.data(), C, n.W1, C, m.nbr1.data(), M1, gP0.data());
    p.ga0.resize((size_t)M0 * C);
    affine_relu_pool_bwd(gP0.data(), p.arg0.data(), m.par1.data(), p.xh0.data(), n.w1, n.b1, M0, p.ga0.data(), gx0.data());
    p.gw[1].resize(C); p.gw[2].resize(C);
    colsum2(p.ga0.data(), p.xh0.data(), M0, p.gw[1].data(), p.gw[2].data());
    inorm_bwd(gx0.data(), p.xh0.data(), p.is0.data(), m.L0.seg, B, gY0.data());
    p.gw[0].resize((size_t)27 * 3 * C);
    conv_bwd_weight(p.F0.data(), 3, gY0.data(), C, m.nbr0.data(), M0, p.gw[0].data());
    return true;
}

// ------------------------------------------------------------------------------------------------------------------ gradient distance
// Row grouping of distillation_loss.py:27-47 with settings.Pyt False: a [K, Cin, Cout] kernel -> one row per OUTPUT channel (elements
// (cin, k)), 2-D tensors row by row, the 1-D bias one row.
struct RowView {
    int rows, cols;
    int64_t row_stride, e1, n1, e2;  // element (r, q) at r * row_stride + (q / n1) * e2 + (q % n1) * e1   (q = cin * K + k for kernels)
};
inline RowView view_of(int idx, int J, int S) {
    switch (idx) {
        case 0: return {C, 3 * 27, 1, (int64_t)3 * C, 27, C};      // W0[k][a][c]: row c, q = a * 27 + k -> k * 3 C + a * C
        case 3: return {C, C * 27, 1, (int64_t)C * C, 27, C};
        case 6: return {J, C * S, (int64_t)C * S, 1, C * S, 0};
        case 7: return {1, J, 0, 1, J, 0};
        default: return {1, C, 0, 1, C, 0};
    }
}
template <class T>
inline T at(const T* t, const RowView& v, int r, int q) { return t[r * v.row_stride + (q / v.n1) * v.e2 + (q % v.n1) * v.e1]; }
template <class T>
inline T& at(T* t, const RowView& v, int r, int q) { return t[r * v.row_stride + (q / v.n1) * v.e2 + (q % v.n1) * v.e1]; }

// match_loss(gw_syn, gw_real, metric) and its gradient with respect to the entries of gw_syn listed in `need` (D[i] sized like gw[i])
// (``need``: the entries of gw_syn that depend on the synthetic points in the reference's step -- the op-level entry point asks for all eight)
constexpr bool kNeedStep[8] = {false, true, false, false, true, false, true, true};
template <class T>
T match_loss(const Vec<T>* gs, const Vec<T>* gr, int metric, int J, int S, Vec<T>* D, const bool* need = kNeedStep) {
    Tick tick_(7);
    T total = 0;
    if (metric == 0 || metric == 1) {  // "ours" (distance_wb per tensor) / "modmse" (distance_mse per tensor)
        for (int i = 0; i < 8; ++i) {
            const RowView v = view_of(i, J, S);
            if (need[i]) D[i].assign(gs[i].size(), 0);
            const T *s = gs[i].data(), *r = gr[i].data();
            T acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc) if (v.rows > 1)
            for (int row = 0; row < v.rows; ++row) {
                // (sums over up to 128 * S elements are kept in double also in the fp32 variant: a running fp32 sum over 1e5 terms drifts by
                // more than the 1e-4 this leg is held to, and torch's own fp32 reductions are pairwise)
                if (metric == 1) {
                    double t = 0;
                    for (int q = 0; q < v.cols; ++q) {
                        const T d = at(r, v, row, q) - at(s, v, row, q);
                        t += (double)d * d;
                        if (need[i]) at(D[i].data(), v, row, q) = (T)-2 * d;
                    }
                    acc += (T)t;
                } else {
                    double rs_ = 0, rr_ = 0, ss_ = 0;
                    for (int q = 0; q < v.cols; ++q) {
                        const double a = at(r, v, row, q), b = at(s, v, row, q);
                        rs_ += a * b; rr_ += a * a; ss_ += b * b;
                    }
                    const T rs = (T)rs_, rr = (T)rr_, ss = (T)ss_;
                    const T nr = std::sqrt(rr), ns = std::sqrt(ss), den = nr * ns + (T)1e-6;
                    acc += (T)1 - rs / den;
                    if (need[i])
                        for (int q = 0; q < v.cols; ++q) {  // d/ds of -rs/den: -(r/den) + rs * nr * (s/ns) / den^2   (ns = 0: the norm's subgradient 0)
                            const T a = at(r, v, row, q), b = at(s, v, row, q);
                            at(D[i].data(), v, row, q) = -a / den + (ns > 0 ? rs * nr * (b / ns) / (den * den) : (T)0);
                        }
                }
            }
            total += acc;
        }
    } else {  // "mse" / "cos" over the concatenation of all eight tensors
        double rs_ = 0, rr_ = 0, ss_ = 0, sq_ = 0;
        for (int i = 0; i < 8; ++i)
            for (size_t e = 0; e < gs[i].size(); ++e) {
                const double a = gr[i][e], b = gs[i][e];
                rs_ += a * b; rr_ += a * a; ss_ += b * b; sq_ += (b - a) * (b - a);
            }
        const T rs = (T)rs_, rr = (T)rr_, ss = (T)ss_, sq = (T)sq_;
        const T nr = std::sqrt(rr), ns = std::sqrt(ss), den = nr * ns + (T)1e-6;
        total = metric == 2 ? sq : (T)1 - rs / den;
        for (int i = 0; i < 8; ++i) {
            if (!need[i]) continue;
            D[i].resize(gs[i].size());
            for (size_t e = 0; e < gs[i].size(); ++e) {
                const T a = gr[i][e], b = gs[i][e];
                D[i][e] = metric == 2 ? (T)2 * (b - a) : -a / den + (ns > 0 ? rs * nr * (b / ns) / (den * den) : (T)0);
            }
        }
    }
    return total;
}

// d loss / d (synthetic features): the second-order pass around MinkowskiEngine's opaque backward functions.  Differentiable entries of
// gw_syn: the two instance-norm weights (gw = sum ga * xhat, ga constant), the classifier weight (dl^T dense) and bias (sum dl).
template <class T>
void second_order(const Net<T>& n, const Pass<T>& p, const Vec<T>* D, T* dfeats) {
    const Maps& m = p.m;
    const int B = m.B, M0 = m.L0.M, M1 = m.L1.M, M2 = m.L2.M, J = n.J;
    // cotangent of dl: u[b][j] = <D[Wc][j], dense[b]> + D[bc][j]; cotangent of dense: sum_j dl[b][j] D[Wc][j]
    Vec<T> u((size_t)B * J), dz((size_t)B * J), hP1((size_t)M2 * C);
    linear_fwd(p.P1.data(), m, D[6].data(), D[7].data(), J, u.data());
    linear_bwd_x(p.dl.data(), m, D[6].data(), J, false, hP1.data());
    for (int b = 0; b < B; ++b) {  // through dl = (softmax(z) - onehot) / B:  dz = p * (u - <p, u>) / B
        T d = 0;
        for (int j = 0; j < J; ++j) d += p.prob[(size_t)b * J + j] * u[(size_t)b * J + j];
        for (int j = 0; j < J; ++j) dz[(size_t)b * J + j] = p.prob[(size_t)b * J + j] * (u[(size_t)b * J + j] - d) / (T)B;
    }
    linear_bwd_x(dz.data(), m, n.Wc, J, true, hP1.data());
    Vec<T> tmp((size_t)M1 * C), hx1((size_t)M1 * C), hY1((size_t)M1 * C), hP0((size_t)M1 * C);
    affine_relu_pool_bwd(hP1.data(), p.arg1.data(), m.par2.data(), p.xh1.data(), n.w5, n.b5, M1, tmp.data(), hx1.data());
    for (int i = 0; i < M1; ++i)
        for (int c = 0; c < C; ++c) hx1[(size_t)i * C + c] += p.ga1[(size_t)i * C + c] * D[4][c];
    inorm_bwd(hx1.data(), p.xh1.data(), p.is1.data(), m.L1.seg, B, hY1.data());
    conv_bwd_data(hY1.data(), C, n.W1, C, m.nbr1.data(), M1, hP0.data());
    Vec<T> tmp0((size_t)M0 * C), hx0((size_t)M0 * C), hY0((size_t)M0 * C), hF0((size_t)M0 * 3);
    affine_relu_pool_bwd(hP0.data(), p.arg0.data(), m.par1.data(), p.xh0.data(), n.w1, n.b1, M0, tmp0.data(), hx0.data());
    for (int i = 0; i < M0; ++i)
        for (int c = 0; c < C; ++c) hx0[(size_t)i * C + c] += p.ga0[(size_t)i * C + c] * D[1][c];
    inorm_bwd(hx0.data(), p.xh0.data(), p.is0.data(), m.L0.seg, B, hY0.data());
    conv_bwd_data(hY0.data(), C, n.W0, 3, m.nbr0.data(), M0, hF0.data());
    for (int i = 0; i < m.N; ++i)  // voxel mean backward
        for (int a = 0; a < 3; ++a) dfeats[(size_t)i * 3 + a] = hF0[(size_t)m.inv[i] * 3 + a] / (T)m.cnt0[m.inv[i]];
}

template <class T>
int64_t param_offsets(int J, int S, int64_t* off) {
    const int64_t sz[8] = {27 * 3 * C, C, C, (int64_t)27 * C * C, C, C, (int64_t)J * C * S, J};
    int64_t t = 0;
    for (int i = 0; i < 8; ++i) {
        off[i] = t;
        t += sz[i];
    }
    return t;
}

template <class T>
int class_step(const dd3d_cpu_net* cn, const double* rcoords, const T* rfeats, const int* roff, int Br, const double* scoords, const T* sfeats,
               const int* soff, int Bs, int label, int metric, T* loss, T* grad_sfeats, T* gw_real, T* gw_syn) {
    if (!cn || Br <= 0 || Bs <= 0 || metric < 0 || metric > 3 || label < 0 || label >= cn->J) return -1;
    if (!roff || !soff || roff[Br] <= roff[0] || soff[Bs] <= soff[0]) return -1;  // a batch without a single point (the reference's TensorField raises)
    Net<T> n{cn->J, cn->E0 * cn->E1 * cn->E2, (const T*)cn->W0, (const T*)cn->w1, (const T*)cn->b1, (const T*)cn->W1,
             (const T*)cn->w5, (const T*)cn->b5, (const T*)cn->Wc, (const T*)cn->bc};
    // The two passes live on between calls (one set per arithmetic type): their ~60 MB of row buffers per class would otherwise be mapped,
    // zeroed and unmapped again by every call -- a fifth of a step's time.  Single-threaded callers only (this is a test / baseline leg).
    static Pass<T> pr, ps;
    pr.m = Maps();
    ps.m = Maps();
    if (!forward_backward(n, *cn, rcoords, rfeats, roff, Br, label, pr)) return pr.m.range_ok ? -2 : -3;
    if (!forward_backward(n, *cn, scoords, sfeats, soff, Bs, label, ps)) return ps.m.range_ok ? -2 : -3;
    Vec<T> D[8];
    *loss = match_loss(ps.gw, pr.gw, metric, n.J, n.S, D);
    if (grad_sfeats) second_order(n, ps, D, grad_sfeats);
    int64_t off[8];
    param_offsets<T>(n.J, n.S, off);
    for (int i = 0; i < 8; ++i) {
        if (gw_real) std::memcpy(gw_real + off[i], pr.gw[i].data(), sizeof(T) * pr.gw[i].size());
        if (gw_syn) std::memcpy(gw_syn + off[i], ps.gw[i].data(), sizeof(T) * ps.gw[i].size());
    }
    return 0;
}

}  // namespace

CPU_API int64_t dd3d_cpu_num_params(const dd3d_cpu_net* n) {
    int64_t off[8];
    return param_offsets<float>(n->J, n->E0 * n->E1 * n->E2, off);
}
CPU_API void dd3d_cpu_set_threads(int t) { omp_set_num_threads(t > 0 ? t : 1); }
CPU_API int dd3d_cpu_max_threads(void) { return omp_get_max_threads(); }
CPU_API void dd3d_cpu_prof_dump(void) {
    for (int i = 0; i < 9; ++i)
        if (g_prof.t[i] > 0) fprintf(stderr, "dd3d_cpu %-16s %8.3f s\n", g_prof.name[i], g_prof.t[i]);
}

// One class of one matching step.  coords: [N,3] spatial coordinates ALREADY divided by the quantisation size (as doubles: a float value
// is exact in a double, so the floor is the caller's); feats: [N,3]; off[B+1]: first point of every cloud.  Outputs: the class's
// matching loss, d loss / d (synthetic features) [Ns,3] (nullable), the two gradient lists concatenated in parameters() order (nullable).
// Returns 0, -1 for bad arguments, -2 when a stride-4 voxel falls outside the dense extent (MinkowskiEngine raises there), -3 when a
// coordinate does not fit the 16 bits per axis of this leg's packed voxel keys (|floor(c)| >= 32000, or more than 65535 clouds).
CPU_API int dd3d_cpu_class_step_f64(const dd3d_cpu_net* n, const double* rc, const double* rf, const int* ro, int Br, const double* sc, const double* sf,
                                    const int* so, int Bs, int label, int metric, double* loss, double* grad, double* gw_real, double* gw_syn) {
    return class_step<double>(n, rc, rf, ro, Br, sc, sf, so, Bs, label, metric, loss, grad, gw_real, gw_syn);
}
CPU_API int dd3d_cpu_class_step_f32(const dd3d_cpu_net* n, const double* rc, const float* rf, const int* ro, int Br, const double* sc, const float* sf,
                                    const int* so, int Bs, int label, int metric, float* loss, float* grad, float* gw_real, float* gw_syn) {
    return class_step<float>(n, rc, rf, ro, Br, sc, sf, so, Bs, label, metric, loss, grad, gw_real, gw_syn);
}


// ================================================================================================================== op by op
// The same restatement one operator at a time, on the array layouts of include/dd3d.h (int32 coordinate rows (batch, c1, c2, c3), neighbour
// tables nbr[K][ld] with -1 = absent, children tables kids[8][ld], seg_off[B+1], arg-max rows, parent rows, flat site per pooled voxel), so
// that a test can hand the arrays of a HIP call to the CPU leg and compare: maps bit for bit, features to rounding.  Host pointers, no
// streams, no workspaces; every function returns 0, -1 (bad arguments), -2 (outside the dense extent) or -3 (beyond the 16-bit keys).
// oracle/cpu/dd3d_cpu.h names the libdd3d_hip.so entry point each one mirrors.
namespace {

bool level_from_rows(const int32_t* rows, int64_t M, Level& lv) {
    lv.M = (int)M;
    lv.Cd.assign(rows, rows + 4 * M);
    lv.tab.reserve((size_t)M * 2);
    for (int64_t v = 0; v < M; ++v) {
        const int32_t* c = rows + 4 * v;
        if (c[0] < 0 || c[0] > 65535) return false;
        for (int a = 1; a < 4; ++a)
            if (c[a] <= -32000 || c[a] >= 32000) return false;
        lv.tab.emplace(Key::pack(c[0], c[1], c[2], c[3]), (int)v);
    }
    return true;
}

// a [K][ld] table as the [K][M] table the loops above index
const int* dense_table(const int32_t* t, int64_t ld, int64_t M, int K, std::vector<int>& tmp) {
    if (ld == M) return t;
    tmp.resize((size_t)K * M);
    for (int k = 0; k < K; ++k) std::memcpy(&tmp[(size_t)k * M], t + (size_t)k * ld, sizeof(int) * M);
    return tmp.data();
}

void maps_for_linear(const int32_t* site, int64_t M2, const int32_t* seg_off, int B, int S, Maps& m) {
    m.B = B;
    m.S = S;
    m.L2.M = (int)M2;
    m.L2.seg.assign(seg_off, seg_off + B + 1);
    m.L2.Cd.assign((size_t)M2 * 4, 0);
    for (int b = 0; b < B; ++b)
        for (int v = seg_off[b]; v < seg_off[b + 1]; ++v) m.L2.Cd[(size_t)v * 4] = b;
    m.site.assign(site, site + M2);
}

template <class T>
int op_match_loss(const T* gw_syn, const T* gw_real, int J, int S, int metric, T* loss, T* d_gw_syn) {
    if (!gw_syn || !gw_real || !loss || J <= 0 || S <= 0 || metric < 0 || metric > 3) return -1;
    int64_t off[8];
    const int64_t total = param_offsets<T>(J, S, off);
    Vec<T> gs[8], gr[8], D[8];
    for (int i = 0; i < 8; ++i) {
        const int64_t n = (i < 7 ? off[i + 1] : total) - off[i];
        gs[i].assign(gw_syn + off[i], gw_syn + off[i] + n);
        gr[i].assign(gw_real + off[i], gw_real + off[i] + n);
    }
    const bool all[8] = {true, true, true, true, true, true, true, true};
    *loss = match_loss(gs, gr, metric, J, S, D, all);
    if (d_gw_syn)
        for (int i = 0; i < 8; ++i) std::memcpy(d_gw_syn + off[i], D[i].data(), sizeof(T) * D[i].size());
    return 0;
}

template <class T>
int op_softmax_ce(const T* logits, const int32_t* labels, int B, int J, T* loss, T* dlogits) {
    if (!logits || !labels || B <= 0 || J <= 0) return -1;
    T total = 0;
    for (int b = 0; b < B; ++b) {
        const T* z = logits + (size_t)b * J;
        if (labels[b] < 0 || labels[b] >= J) return -1;
        T mx = z[0];
        for (int j = 1; j < J; ++j) mx = std::max(mx, z[j]);
        T se = 0;
        for (int j = 0; j < J; ++j) se += std::exp(z[j] - mx);
        const T lse = mx + std::log(se);
        total += lse - z[labels[b]];
        if (dlogits)
            for (int j = 0; j < J; ++j) dlogits[(size_t)b * J + j] = (std::exp(z[j] - lse) - (j == labels[b] ? (T)1 : (T)0)) / (T)B;
    }
    if (loss) *loss = total / (T)B;
    return 0;
}

}  // namespace

CPU_API int dd3d_cpu_quantize_unique(const double* coords, const int32_t* pt_off, int B, int32_t* Cq, int32_t* inv, int32_t* seg_off, int64_t* M) {
    if (!coords || !pt_off || B <= 0 || B > 65535 || !Cq || !inv || !seg_off || !M) return -1;
    const int N = pt_off[B];
    std::vector<int> q((size_t)N * 4), map;
    for (int b = 0; b < B; ++b)
        for (int i = pt_off[b]; i < pt_off[b + 1]; ++i) {
            q[(size_t)i * 4] = b;
            for (int a = 0; a < 3; ++a) {
                const double f = std::floor(coords[(size_t)i * 3 + a]);
                if (!(f > -32000.0 && f < 32000.0)) return -3;
                q[(size_t)i * 4 + 1 + a] = (int)f;
            }
        }
    Level lv;
    unique_rows(q, N, lv, map, B);
    std::memcpy(Cq, lv.Cd.data(), sizeof(int) * lv.Cd.size());
    std::memcpy(inv, map.data(), sizeof(int) * (size_t)N);
    std::memcpy(seg_off, lv.seg.data(), sizeof(int) * (size_t)(B + 1));
    *M = lv.M;
    return 0;
}

CPU_API int dd3d_cpu_stride_coords(const int32_t* Cin, int64_t M_in, int B, int stride, int32_t* Cout, int32_t* parent, int32_t* seg_off, int64_t* M_out) {
    if (!Cin || M_in < 0 || B <= 0 || stride <= 0 || !Cout || !parent || !seg_off || !M_out) return -1;
    Level in, out;
    in.M = (int)M_in;
    in.Cd.assign(Cin, Cin + 4 * M_in);
    std::vector<int> par;
    stride_up(in, stride, out, par, B);
    std::memcpy(Cout, out.Cd.data(), sizeof(int) * out.Cd.size());
    std::memcpy(parent, par.data(), sizeof(int) * (size_t)M_in);
    std::memcpy(seg_off, out.seg.data(), sizeof(int) * (size_t)(B + 1));
    *M_out = out.M;
    return 0;
}

CPU_API int dd3d_cpu_kernel_map(const int32_t* Cq, int64_t M, int tensor_stride, int32_t* nbr) {
    if (!Cq || M < 0 || tensor_stride <= 0 || !nbr) return -1;
    Level lv;
    if (!level_from_rows(Cq, M, lv)) return -3;
    std::vector<int> t;
    kernel_map3(lv, tensor_stride, t);
    std::memcpy(nbr, t.data(), sizeof(int) * t.size());
    return 0;
}

CPU_API int dd3d_cpu_pool_map(const int32_t* Cin, int64_t M_in, const int32_t* Cout, int64_t M_out, int tensor_stride_in, int32_t* kids) {
    if (!Cin || !Cout || M_in < 0 || M_out < 0 || tensor_stride_in <= 0 || !kids) return -1;
    Level in, out;
    if (!level_from_rows(Cin, M_in, in)) return -3;
    out.M = (int)M_out;
    out.Cd.assign(Cout, Cout + 4 * M_out);
    std::vector<int> t;
    pool_map(in, out, tensor_stride_in, t);
    std::memcpy(kids, t.data(), sizeof(int) * t.size());
    return 0;
}

CPU_API int dd3d_cpu_dense_sites(const int32_t* Cq, int64_t M, int tensor_stride, int E0, int E1, int E2, int32_t* site) {
    if (!Cq || M <= 0 || tensor_stride <= 0 || E0 <= 0 || E1 <= 0 || E2 <= 0 || !site) return -1;
    int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
    for (int64_t v = 0; v < M; ++v)
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], Cq[v * 4 + 1 + a]);
    for (int64_t v = 0; v < M; ++v) {
        const int32_t* c = Cq + v * 4;
        const int a = (c[1] - mn[0]) / tensor_stride, d = (c[2] - mn[1]) / tensor_stride, e = (c[3] - mn[2]) / tensor_stride;
        if (a >= E0 || d >= E1 || e >= E2) return -2;
        site[v] = (a * E1 + d) * E2 + e;
    }
    return 0;
}

#define DD3D_CPU_OPS(T, SFX)                                                                                                                          \
    CPU_API int dd3d_cpu_voxel_mean_fwd_##SFX(const T* feats, int64_t N, int Cf, const int32_t* inv, int64_t M, T* F0) {                             \
        if (!feats || !inv || !F0 || N < 0 || M < 0 || Cf <= 0) return -1;                                                                           \
        std::vector<int> cnt((size_t)M, 0);                                                                                                          \
        std::fill(F0, F0 + (size_t)M * Cf, (T)0);                                                                                                    \
        for (int64_t i = 0; i < N; ++i) {                                                                                                            \
            if (inv[i] < 0 || inv[i] >= M) return -1;                                                                                                \
            cnt[inv[i]]++;                                                                                                                           \
            for (int a = 0; a < Cf; ++a) F0[(size_t)inv[i] * Cf + a] += feats[(size_t)i * Cf + a];                                                   \
        }                                                                                                                                            \
        for (int64_t v = 0; v < M; ++v)                                                                                                              \
            for (int a = 0; a < Cf; ++a) F0[(size_t)v * Cf + a] /= (T)std::max(cnt[v], 1);                                                           \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_voxel_mean_bwd_##SFX(const T* dF0, int64_t M, int Cf, const int32_t* inv, int64_t N, T* dfeats) {                           \
        if (!dF0 || !inv || !dfeats || N < 0 || M < 0 || Cf <= 0) return -1;                                                                         \
        std::vector<int> cnt((size_t)M, 0);                                                                                                          \
        for (int64_t i = 0; i < N; ++i) {                                                                                                            \
            if (inv[i] < 0 || inv[i] >= M) return -1;                                                                                                \
            cnt[inv[i]]++;                                                                                                                           \
        }                                                                                                                                            \
        for (int64_t i = 0; i < N; ++i)                                                                                                              \
            for (int a = 0; a < Cf; ++a) dfeats[(size_t)i * Cf + a] = dF0[(size_t)inv[i] * Cf + a] / (T)cnt[inv[i]];                                 \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_conv_fwd_##SFX(const T* X, int Cin, const T* W, int Cout, const int32_t* nbr, int64_t ld, int64_t M, T* Y) {                \
        if (!X || !W || !nbr || !Y || Cin <= 0 || Cout <= 0 || M < 0 || ld < M) return -1;                                                           \
        std::vector<int> tmp;                                                                                                                        \
        conv_fwd(X, Cin, W, Cout, dense_table(nbr, ld, M, 27, tmp), (int)M, Y);                                                                      \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_conv_bwd_data_##SFX(const T* dY, int Cout, const T* W, int Cin, const int32_t* nbr, int64_t ld, int64_t M, T* dX) {         \
        if (!dY || !W || !nbr || !dX || Cin <= 0 || Cout <= 0 || M < 0 || ld < M) return -1;                                                         \
        std::vector<int> tmp;                                                                                                                        \
        conv_bwd_data(dY, Cout, W, Cin, dense_table(nbr, ld, M, 27, tmp), (int)M, dX);                                                               \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_conv_bwd_weight_##SFX(const T* X, int Cin, const T* dY, int Cout, const int32_t* nbr, int64_t ld, int64_t M, T* dW) {       \
        if (!X || !dY || !nbr || !dW || Cin <= 0 || Cout <= 0 || M < 0 || ld < M) return -1;                                                         \
        std::vector<int> tmp;                                                                                                                        \
        conv_bwd_weight(X, Cin, dY, Cout, dense_table(nbr, ld, M, 27, tmp), (int)M, dW);                                                             \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_inorm_fwd_##SFX(const T* Y, const int32_t* seg_off, int B, int Cc, T* xhat, T* inv_std) {                                   \
        if (!Y || !seg_off || !xhat || !inv_std || B <= 0 || Cc != C) return -1;                                                                     \
        inorm_fwd(Y, std::vector<int>(seg_off, seg_off + B + 1), B, xhat, inv_std);                                                                  \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_inorm_bwd_##SFX(const T* g, const T* xhat, const T* inv_std, const int32_t* seg_off, int B, int Cc, T* dY) {                \
        if (!g || !xhat || !inv_std || !seg_off || !dY || B <= 0 || Cc != C) return -1;                                                              \
        inorm_bwd(g, xhat, inv_std, std::vector<int>(seg_off, seg_off + B + 1), B, dY);                                                              \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_affine_relu_pool_fwd_##SFX(const T* xhat, const T* w, const T* b, int Cc, const int32_t* kids, int64_t ld, int64_t M_out,   \
                                                    T* out, int32_t* arg) {                                                                          \
        if (!xhat || !w || !b || !kids || !out || !arg || Cc != C || M_out < 0 || ld < M_out) return -1;                                             \
        std::vector<int> tmp;                                                                                                                        \
        affine_relu_pool_fwd(xhat, w, b, dense_table(kids, ld, M_out, 8, tmp), (int)M_out, out, arg);                                                \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_affine_relu_pool_bwd_##SFX(const T* g_pool, const int32_t* arg, const int32_t* parent, const T* xhat, const T* w,           \
                                                    const T* b, int Cc, int64_t M_in, T* ga, T* g_xhat) {                                            \
        if (!g_pool || !arg || !parent || !xhat || !w || !b || !ga || !g_xhat || Cc != C || M_in < 0) return -1;                                     \
        affine_relu_pool_bwd(g_pool, arg, parent, xhat, w, b, (int)M_in, ga, g_xhat);                                                                \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_sparse_linear_fwd_##SFX(const T* F, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int Cc, int S,           \
                                                 const T* Wc, const T* bias, int J, T* logits) {                                                     \
        if (!F || !site || !seg_off || !Wc || !logits || Cc != C || B <= 0 || S <= 0 || J <= 0 || M < 0) return -1;                                  \
        Maps m;                                                                                                                                      \
        maps_for_linear(site, M, seg_off, B, S, m);                                                                                                  \
        linear_fwd(F, m, Wc, bias, J, logits);                                                                                                       \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_sparse_linear_bwd_x_##SFX(const T* dl, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int Cc, int S,        \
                                                   const T* Wc, int J, T* dF) {                                                                      \
        if (!dl || !site || !seg_off || !Wc || !dF || Cc != C || B <= 0 || S <= 0 || J <= 0 || M < 0) return -1;                                     \
        Maps m;                                                                                                                                      \
        maps_for_linear(site, M, seg_off, B, S, m);                                                                                                  \
        linear_bwd_x(dl, m, Wc, J, false, dF);                                                                                                       \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_sparse_linear_bwd_w_##SFX(const T* dl, const T* F, const int32_t* site, int64_t M, const int32_t* seg_off, int B, int Cc,   \
                                                   int S, int J, T* gW) {                                                                            \
        if (!dl || !F || !site || !seg_off || !gW || Cc != C || B <= 0 || S <= 0 || J <= 0 || M < 0) return -1;                                      \
        Maps m;                                                                                                                                      \
        maps_for_linear(site, M, seg_off, B, S, m);                                                                                                  \
        linear_bwd_w(dl, F, m, J, gW);                                                                                                               \
        return 0;                                                                                                                                    \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_softmax_ce_##SFX(const T* logits, const int32_t* labels, int B, int J, T* loss, T* dlogits) {                               \
        return op_softmax_ce<T>(logits, labels, B, J, loss, dlogits);                                                                                \
    }                                                                                                                                                \
    CPU_API int dd3d_cpu_match_loss_##SFX(const T* gw_syn, const T* gw_real, int J, int S, int metric, T* loss, T* d_gw_syn) {                       \
        return op_match_loss<T>(gw_syn, gw_real, J, S, metric, loss, d_gw_syn);                                                                      \
    }

DD3D_CPU_OPS(double, f64)
DD3D_CPU_OPS(float, f32)
