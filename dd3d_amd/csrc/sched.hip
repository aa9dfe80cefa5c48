// Launch schedule of the 128-channel sparse convolution (k_conv128_rows with `perm` and `tile_map`, conv.hip).
//
// The convolution is output-stationary: a 32-row block skips an offset only when NONE of its rows has that neighbour, and a 128-row
// tile walks the union of its four blocks' offsets.  In cloud order stroke digits execute 1.37x the algorithmic MFMA work (73 % of the
// 9 in-plane neighbours present).  Two steps remove most of that:
//   1. row order: the rows are grouped by their K-bit neighbour pattern -- a counting sort over 4096 buckets of the pattern (rows
//      with the same pattern share a bucket; which of them comes first inside a bucket is left to the atomics, see below): 1.04x the
//      algorithmic work at 128-row granularity.  A row's sum does not depend on the rows it shares a block with (the skipped products
//      are exact zeros), so the convolution's output is bit-identical for ANY order -- the order only decides how fast it runs.
//   2. tile placement: a sorted launch has heavy tiles (all 9 offsets) and light ones, and a launch on an idle GPU puts workgroup b on
//      CU b mod 256 with every workgroup resident at once (tools/place_probe), so it lasts as long as the most loaded CU.  The tiles
//      are therefore ranked by their true cost (offsets in the tile's union) and dealt to the CUs in a snake -- round 0 forwards, round
//      1 backwards, ... -- which is as even as greedy longest-processing-time assignment on these distributions (76 vs 108 steps on
//      the most loaded CU for the cfg2 real batch).
// One schedule serves the forward and the mirrored (backward-data) launch over the same kernel map: mirroring permutes the bits of every
// pattern alike.
#include "common.h"

namespace {

constexpr int NB = 4096;  // buckets of the neighbour pattern

__device__ __forceinline__ unsigned row_pattern(const int* __restrict__ nbr, int64_t ld, int K, int row) {
    int v[27];  // K <= 27: all table reads of the row in flight together
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = nbr[(int64_t)min(k, K - 1) * ld + row];
    unsigned key = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) key |= (unsigned)(k < K && v[k] >= 0) << k;
    return key;
}
// Bucket of a pattern = 12 of its bits: the 8 non-centre offsets with first spatial coordinate 0 (k = 1 + 3a + 9b; ALL the offsets that
// planar clouds such as MNIST ever have, so every distinct pattern of theirs gets a bucket of its own) and 4 offsets out of that plane
// (k = 12, 14: the two face neighbours; k = 3, 23: two edge neighbours).  Rows of volumetric clouds have far more than 4096 distinct
// patterns; there the rows of a bucket agree on these 12 offsets, and the ~2/3 of them that are absent are skipped by every block of
// the bucket (a hash of the whole pattern would mix unrelated rows and skip nothing).
__device__ __forceinline__ int pattern_bucket(unsigned key) {
    const unsigned plane = ((key >> 1) & 1u) | ((key >> 4) & 1u) << 1 | ((key >> 7) & 1u) << 2 | ((key >> 10) & 1u) << 3 | ((key >> 16) & 1u) << 4 |
                           ((key >> 19) & 1u) << 5 | ((key >> 22) & 1u) << 6 | ((key >> 25) & 1u) << 7;
    const unsigned out = ((key >> 12) & 1u) | ((key >> 14) & 1u) << 1 | ((key >> 3) & 1u) << 2 | ((key >> 23) & 1u) << 3;
    return (int)(out << 8 | plane);
}

// keys[i] = pattern of row i; hist[b] += rows of bucket b (one LDS histogram per workgroup -- one row per thread, so that a
// launch of 87 k rows already spreads over many CUs -- then one global add per bucket it met)
__global__ __launch_bounds__(1024) void k_row_keys(const int* __restrict__ nbr, int64_t ld, int K, int M_bound, const int* __restrict__ m_dev,
                                                   unsigned* __restrict__ keys, int* __restrict__ hist) {
    __shared__ int s_h[NB];
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    const int nt = blockDim.x, i0 = blockIdx.x * nt;  // one row per thread
    if (i0 >= M) return;
    for (int b = threadIdx.x; b < NB; b += nt) s_h[b] = 0;
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i < M) {
        const unsigned key = row_pattern(nbr, ld, K, i);
        keys[i] = key;
        atomicAdd(&s_h[pattern_bucket(key)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += nt)
        if (s_h[b]) atomicAdd(&hist[b], s_h[b]);
}

// perm / sorted_keys: every workgroup scans the bucket counts itself (4096 ints from L2: cheaper than a scan kernel of its own in the
// dependent chain), reserves, per bucket it holds, a run behind the bucket's fill counter (one global atomic) and ranks its rows inside
// the run with LDS atomics.  The order inside a bucket therefore depends on the atomics' arrival; the convolution's output does not
// (see the file comment).  Workgroup 0 also writes ends[b] = end of bucket b (what the second level walks).
__global__ __launch_bounds__(1024) void k_row_scatter(const unsigned* __restrict__ keys, int M_bound, const int* __restrict__ m_dev,
                                                      const int* __restrict__ hist, int* __restrict__ fill, int* __restrict__ ends,
                                                      int* __restrict__ perm, unsigned* __restrict__ sorted_keys, int* __restrict__ perm_refine,
                                                      int refine_min, int* __restrict__ inv) {
    __shared__ int s_cnt[NB], s_base[NB];
    __shared__ int s_w[16];
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    if (perm_refine && M >= refine_min) perm = perm_refine, inv = nullptr;  // a second level follows (decided by the TRUE row count)
    const int i0 = blockIdx.x * 4096;
    if (i0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = hist[tid * 4 + j];
        sum += v[j];
    }
    for (int b = tid; b < NB; b += 1024) s_cnt[b] = 0;
    unsigned key[4];
    int bucket[4], rank[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 1024 + tid;
        bucket[j] = -1;
        if (i < M) {
            key[j] = keys[i];
            bucket[j] = pattern_bucket(key[j]);
        }
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int q = 0; q < w; ++q) base += s_w[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_base[tid * 4 + j] = base;  // first position of bucket tid*4+j
        base += v[j];
        if (blockIdx.x == 0) ends[tid * 4 + j] = base;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (bucket[j] >= 0) rank[j] = atomicAdd(&s_cnt[bucket[j]], 1);
    __syncthreads();
    for (int b = tid; b < NB; b += 1024)
        if (s_cnt[b]) s_base[b] += atomicAdd(&fill[b], s_cnt[b]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (bucket[j] < 0) continue;
        const int pos = s_base[bucket[j]] + rank[j];
        perm[pos] = i0 + j * 1024 + tid;
        sorted_keys[pos] = key[j];
    }
    // position of every row in the pattern order (k_table_permute scatters the table's columns there); not with a second level
    if (inv) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (bucket[j] >= 0) inv[i0 + j * 1024 + tid] = s_base[bucket[j]] + rank[j];
    }
}

// The kernel map in pattern order, nbr_sorted[k][pos] = nbr[k][perm[pos]]: the table columns a tile of the sorted launch needs are then
// contiguous (gathered per tile they cost every workgroup a dozen microseconds of scattered 4-byte reads before its first MFMA).
// Coalesced reads; 4-byte writes that land in runs wherever neighbouring rows share a bucket.  One workgroup per (256 rows, offset).
__global__ __launch_bounds__(256) void k_table_permute(const int* __restrict__ nbr, int64_t ld, int M_bound, const int* __restrict__ m_dev,
                                                       const int* __restrict__ inv, int refine_min, int* __restrict__ nbr_sorted) {
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    if (refine_min > 0 && M >= refine_min) return;  // a second sort level ran: the table stays as it is (tile_mask bit 31 says so)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int64_t col = (int64_t)blockIdx.y * ld;
    nbr_sorted[col + inv[i]] = nbr[col + i];
}

// Second level, for launches of at least 262144 rows under a bound of 2^20 or more (volumetric clouds; planar ones have none of these 12
// offsets): inside every first-level bucket the rows are ordered by 12 further bits of the pattern (the six edge neighbours that touch the plane's axes first, then six corners), so that a 32-row block of
// an average bucket agrees on ~5 more offsets.  One workgroup per bucket, counting sort in LDS (ends[b] = the bucket's end: the cursors
// after k_row_scatter); the order among equal keys is again left to the atomics.
__device__ __forceinline__ int pattern_bucket2(unsigned key) {
    constexpr int ks[12] = {26, 24, 20, 18, 8, 6, 17, 15, 11, 9, 21, 5};  // least significant first
    unsigned b = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) b |= ((key >> ks[j]) & 1u) << j;
    return (int)b;
}
__global__ __launch_bounds__(1024) void k_bucket_refine(const int* __restrict__ ends, const int* __restrict__ perm_in, const unsigned* __restrict__ keys_in,
                                                        int* __restrict__ perm_out, unsigned* __restrict__ keys_out, int M_bound,
                                                        const int* __restrict__ m_dev, int refine_min) {
    __shared__ int s_cnt[NB];
    __shared__ int s_w[16];
    if ((m_dev ? min(*m_dev, M_bound) : M_bound) < refine_min) return;  // the first level wrote perm itself
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int b = blockIdx.x; b < NB; b += gridDim.x) {  // 512 workgroups, eight buckets each
        const int start = b ? ends[b - 1] : 0, end = ends[b];
        if (end - start <= 0) continue;
        if (end - start <= 32) {  // one block whatever the order
            if (tid < end - start) {
                perm_out[start + tid] = perm_in[start + tid];
                keys_out[start + tid] = keys_in[start + tid];
            }
            continue;
        }
        for (int q = tid; q < NB; q += 1024) s_cnt[q] = 0;
        __syncthreads();
        for (int i = start + tid; i < end; i += 1024) atomicAdd(&s_cnt[pattern_bucket2(keys_in[i])], 1);
        __syncthreads();
        int v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = s_cnt[tid * 4 + j];
            sum += v[j];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        int base = start + incl - sum;
        for (int q = 0; q < w; ++q) base += s_w[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_cnt[tid * 4 + j] = base;  // the bin's write cursor
            base += v[j];
        }
        __syncthreads();
        for (int i = start + tid; i < end; i += 1024) {
            const unsigned key = keys_in[i];
            const int pos = atomicAdd(&s_cnt[pattern_bucket2(key)], 1);
            perm_out[pos] = perm_in[i];
            keys_out[pos] = key;
        }
        __syncthreads();  // s_cnt / s_w are reused by the next bucket
    }
}

// tile_map[b] = tile that workgroup b runs, or -1.  Up to 2048 tiles (every workgroup resident at once, or nearly): a tile's cost is the number of
// offsets present in it (thread t folds the 128 patterns of tile t); the tiles are ranked by (cost descending, index) and rank i goes to
// round i / 256, CU position i % 256 forwards in even rounds and backwards in odd ones.  More tiles than that are dispatched as slots
// free up, where plain order is as good as any.
// cost[t] = offsets present in tile t (thread t folds the tile's patterns: 64 cache lines per load instruction, which one CU alone takes
// 40 us to get through -- hence a kernel of its own, 64 tiles per workgroup)
template <int TR>  // rows per tile: 128 (k_conv128_rows) or 32 (k_conv128<32,1,4>)
__global__ __launch_bounds__(64) void k_tile_cost(const unsigned* __restrict__ sorted_keys, int M_bound, const int* __restrict__ m_dev,
                                                  const unsigned* __restrict__ refined_keys, int refine_min, int* __restrict__ cost,
                                                  unsigned* __restrict__ tile_mask) {
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    const bool refined = refined_keys && M >= refine_min;
    if (refined) sorted_keys = refined_keys;
    const int ntile = (M + TR - 1) / TR, t = blockIdx.x * 64 + threadIdx.x;
    if (t >= ntile || (t >= 2048 && !tile_mask)) return;
    unsigned u = 0;
    const uint4* k4 = (const uint4*)(sorted_keys + (int64_t)t * TR);   // TR * 4 B aligned: the workspace carve is 256-B aligned
    const int n = min(TR, M - t * TR);
    if (n == TR) {  // TR / 4 independent 16-B loads
        uint4 q[TR / 4];
#pragma unroll
        for (int i = 0; i < TR / 4; ++i) q[i] = k4[i];
#pragma unroll
        for (int i = 0; i < TR / 4; ++i) u |= q[i].x | q[i].y | q[i].z | q[i].w;
    } else {
        for (int i = 0; i < n; ++i) u |= sorted_keys[(int64_t)t * TR + i];
    }
    if (t < 2048) cost[t] = __popc(u);
    // the offsets some row of the tile has: all its workgroup needs to read of the table (bit 31: the table was NOT permuted, see the launcher)
    if (tile_mask) tile_mask[t] = u | (refined ? 0x80000000u : 0u);
}

template <int TR>
__global__ __launch_bounds__(1024) void k_tile_schedule(const int* __restrict__ g_cost, int M_bound, const int* __restrict__ m_dev, int n_blocks,
                                                        int* __restrict__ tile_map) {
    constexpr int MAXT = 2048;  // two tiles per thread
    __shared__ short s_cnt[28][32], s_base[28][32];
    __shared__ int s_tot[28];
    __shared__ __align__(16) int s_load[256];
    __shared__ short s_tile[MAXT], s_rcost[MAXT];
    const int tid = threadIdx.x;
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    const int ntile = (M + TR - 1) / TR;
    for (int b = tid; b < n_blocks; b += 1024) tile_map[b] = -1;
    __syncthreads();
    if (ntile > MAXT) {
        for (int t = tid; t < ntile && t < n_blocks; t += 1024) tile_map[t] = t;
        return;
    }
    for (int i = tid; i < 28 * 32; i += 1024) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    // rank of tile t in (cost descending, index ascending): a counting sort over the 28 possible costs.  Chunk q = 64 consecutive tiles
    // (one wave, ballots per cost value); s_cnt[v][q] = tiles of cost v in chunk q.
    const int lane = tid & 63, w = tid >> 6;
    int cost[2], within[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = tid + j * 1024, q = j * 16 + w;
        const int c = t < ntile ? min(g_cost[t], 27) : -1;
        cost[j] = c;
        within[j] = 0;
        for (unsigned long long rest = __ballot(c >= 0); rest;) {  // one pass per distinct cost of the chunk
            const int v = __shfl(c, __builtin_ctzll(rest), 64);
            const unsigned long long m = __ballot(c == v);
            if (c == v) within[j] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_cnt[v][q] = (short)__popcll(m);
            rest &= ~m;
        }
    }
    __syncthreads();
    // s_base[v][q] = tiles ranked before the first cost-v tile of chunk q: all tiles of a higher cost, and the cost-v tiles of earlier chunks
    if (tid < 28) {
        int tot = 0;
        for (int q = 0; q < 32; ++q) tot += s_cnt[tid][q];
        s_tot[tid] = tot;
    }
    __syncthreads();
    if (tid < 28) {
        int before = 0;
        for (int v = 27; v > tid; --v) before += s_tot[v];
        for (int q = 0; q < 32; ++q) {
            s_base[tid][q] = (short)before;
            before += s_cnt[tid][q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = tid + j * 1024, c = cost[j];
        if (t >= ntile) continue;
        const int rank = s_base[c][j * 16 + w] + within[j];
        s_tile[rank] = (short)t;
        s_rcost[rank] = (short)c;
    }
    __syncthreads();
    // Round r deals the tiles ranked [256 r, 256 r + 256) to the 256 CUs (workgroup 256 r + c runs on CU c): the heaviest of them to the CU
    // with the least work so far, the next to the second least loaded, ... (ties by CU index).  Rounds 0 and 1 come out as the snake
    // (forwards, backwards); from round 2 on the match follows the loads the earlier rounds left, which the snake ignored.
    const int c = tid & 255;
    int load = 0;
    // (rounds 0 and 1 in closed form: CU c takes rank c, then -- the loads now descending in c -- rank 256 + (255 - c))
    if (tid < 256) {
        if (c < ntile) {
            if (c < n_blocks) tile_map[c] = s_tile[c];
            load = s_rcost[c];
        }
        const int i = 256 + 255 - c;
        if (i < ntile) {
            if (256 + c < n_blocks) tile_map[256 + c] = s_tile[i];
            load += s_rcost[i];
        }
    }
    for (int r = 2; r * 256 < ntile; ++r) {
        if (tid < 256) s_load[c] = load;
        __syncthreads();
        if (tid < 256) {
            int pos = 0;
            const int4* l4 = (const int4*)s_load;
            for (int u4 = 0; u4 < 64; ++u4) {
                const int4 q = l4[u4];
                const int u = u4 * 4;
                pos += (q.x < load) || (q.x == load && u < c);
                pos += (q.y < load) || (q.y == load && u + 1 < c);
                pos += (q.z < load) || (q.z == load && u + 2 < c);
                pos += (q.w < load) || (q.w == load && u + 3 < c);
            }
            const int i = r * 256 + pos;
            if (i < ntile) {
                if (r * 256 + c < n_blocks) tile_map[r * 256 + c] = s_tile[i];
                load += s_rcost[i];
            }
        }
        __syncthreads();
    }
}

struct SchedWs {
    unsigned *keys, *sorted_keys;
    int *hist, *fill, *ends, *cost, *perm_tmp, *inv;  // hist and fill are adjacent: one memset
};
constexpr int64_t REFINE_MIN = 1 << 18;         // true rows from which the second sort level runs (below, a bucket holds too few rows to gain)
constexpr int64_t REFINE_LAUNCH_MIN = 1 << 20;  // row bound from which its kernel is launched at all (an idle node still costs the chain ~15 us)

int64_t carve(int64_t M, void* ws, SchedWs* out) {
    const int64_t n = M > 0 ? M : 1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        void* p = ws ? (char*)ws + off : nullptr;
        off += align_up64(bytes, 256);
        return p;
    };
    SchedWs w;
    w.keys = (unsigned*)take(n * 4);
    w.sorted_keys = (unsigned*)take(n * 4);
    w.hist = (int*)take(2 * NB * 4);
    w.fill = w.hist ? w.hist + NB : nullptr;
    w.ends = (int*)take(NB * 4);
    w.cost = (int*)take(2048 * 4);
    w.perm_tmp = M >= REFINE_LAUNCH_MIN ? (int*)take(n * 4) : nullptr;
    w.inv = (int*)take(n * 4);
    if (out) *out = w;
    return off;
}

}  // namespace

// rows per tile of a scheduled launch over M rows (M = the bound when a device-side count is used): the same rule picks the kernel in
// dd3d_conv128_ordered
DD3D_API int dd3d_conv_schedule_tile_rows(int64_t M) { return M >= 65536 ? 128 : 32; }

DD3D_API int64_t dd3d_conv_schedule_blocks(int64_t M) { return ceil_div64(ceil_div64(M > 0 ? M : 1, dd3d_conv_schedule_tile_rows(M)), 256) * 256; }

DD3D_API int64_t dd3d_conv_schedule_ws_bytes(int64_t M) { return carve(M, nullptr, nullptr); }

// perm[M] (int32): the rows in pattern order; tile_map[dd3d_conv_schedule_blocks(M)] (int32): workgroup -> tile of that order, or -1.
// Optional (both or neither): nbr_sorted[K][ld] = the kernel map with its columns in pattern order, tile_mask[ceil(M / tile rows)] = the
// offsets present in each tile -- with them a workgroup of dd3d_conv128_ordered reads its part of the table in a few coalesced loads.
// M is a bound when m_dev is given (rows beyond *m_dev sort last and get no tile).
DD3D_API int dd3d_conv_schedule(const int32_t* nbr, int64_t ld, int K, int64_t M, const int32_t* m_dev, int32_t* perm, int32_t* tile_map,
                                int32_t* nbr_sorted, uint32_t* tile_mask, void* ws, int64_t ws_bytes, void* stream) {
    if (M < 0 || K <= 0 || K > 27 || ld < M) return DD3D_EARG;
    if (M >= (1ll << 31) - 256) return DD3D_ESIZE;
    if (M == 0) return 0;
    if (!nbr || !perm || !tile_map || (nbr_sorted == nullptr) != (tile_mask == nullptr) || nbr_sorted == nbr) return DD3D_EARG;
    if (!ws || ws_bytes < dd3d_conv_schedule_ws_bytes(M)) return DD3D_EWS;
    SchedWs w;
    carve(M, ws, &w);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nwg = (unsigned)ceil_div64(M, 4096);
    DD3D_HIP(hipMemsetAsync(w.hist, 0, 2 * NB * 4, st));
    // 512-thread workgroups: in the step this kernel starts beside the first layer's convolution, whose workgroups leave less than a
    // quarter of a CU's registers free, and a 1024-thread workgroup waited for a whole CU to drain (42 us instead of 7; cfg2 step 1.243 ->
    // 1.223 ms, 256 threads: 1.229).  The same change to k_row_scatter made no difference.  DD3D_ROWKEYS_T: measurements.
    static const int rk_t = getenv("DD3D_ROWKEYS_T") ? atoi(getenv("DD3D_ROWKEYS_T")) : 512;
    if (rk_t < 64 || rk_t > 1024 || rk_t % 64) return DD3D_EARG;
    k_row_keys<<<dim3((unsigned)ceil_div64(M, rk_t)), rk_t, 0, st>>>(nbr, ld, K, (int)M, m_dev, w.keys, w.hist);
    const bool refine = M >= REFINE_LAUNCH_MIN;  // M may be a bound: the kernels take the second level only when the TRUE count reaches REFINE_MIN
    // (with a second sort level the table is not permuted: dd3d_conv128_ordered then gathers it through perm, told by tile_mask bit 31)
    k_row_scatter<<<dim3(nwg), 1024, 0, st>>>(w.keys, (int)M, m_dev, w.hist, w.fill, w.ends, perm, w.sorted_keys, refine ? w.perm_tmp : nullptr,
                                              (int)REFINE_MIN, nbr_sorted ? w.inv : nullptr);
    if (nbr_sorted)
        k_table_permute<<<dim3((unsigned)ceil_div64(M, 256), (unsigned)K), 256, 0, st>>>(nbr, ld, (int)M, m_dev, w.inv, refine ? (int)REFINE_MIN : 0, nbr_sorted);
    if (refine)
        k_bucket_refine<<<dim3(512), 1024, 0, st>>>(w.ends, w.perm_tmp, w.sorted_keys, perm, w.keys /* done with: reused */, (int)M, m_dev, (int)REFINE_MIN);
    // the placement uses the costs of up to 2048 tiles (of the TRUE row count: M may be a bound)
    const int nb = (int)dd3d_conv_schedule_blocks(M);
    const int64_t tr = dd3d_conv_schedule_tile_rows(M), nt = (ceil_div64(M, tr) < 2048 || tile_mask) ? ceil_div64(M, tr) : 2048;
    if (tr == 128) {
        k_tile_cost<128><<<dim3((unsigned)ceil_div64(nt, 64)), 64, 0, st>>>(w.sorted_keys, (int)M, m_dev, refine ? w.keys : nullptr, (int)REFINE_MIN, w.cost,
                                                                             tile_mask);
        k_tile_schedule<128><<<1, 1024, 0, st>>>(w.cost, (int)M, m_dev, nb, tile_map);
    } else {
        k_tile_cost<32><<<dim3((unsigned)ceil_div64(nt, 64)), 64, 0, st>>>(w.sorted_keys, (int)M, m_dev, nullptr, 0, w.cost, tile_mask);
        k_tile_schedule<32><<<1, 1024, 0, st>>>(w.cost, (int)M, m_dev, nb, tile_map);
    }
    DD3D_CHECK_LAUNCH();
    return 0;
}
