// Cloud-local coordinate maps: the whole pyramid behind MEConvExp.forward (models/MEConv.py:247,224,227,250-254) in three launches.
//
// Every key of the voxel hash starts with the batch index, so duplicates, kernel-map neighbours and pooling children of a voxel
// all live in the voxel's own cloud.  A workgroup therefore takes ONE cloud, keeps its coordinates and three small hash tables
// (tensor strides 1, 2, 4) in LDS and produces, without a single global random access:
//   quantise (floor) -> unique (first occurrence) -> inverse map + voxel->points CSR        [TensorField.sparse()]
//   k=3 kernel maps at strides 1 and 2                                                    [MinkowskiConvolution x2]
//   stride-2 coordinate maps 1->2 and 2->4 with their k=2 pooling maps and parent rows    [MinkowskiMaxPooling x2]
//   per-cloud row ranges at the three strides and the dense() placement of the last one   [SparseTensor.dense()]
// Pass A counts the voxels of every cloud at the three strides (and folds the per-group minimum corner), one scan turns the counts
// into row ranges, pass B rebuilds the same tables (deterministic) and emits everything at its global row.  The results are
// bit-identical to the generic global-hash path in coords.hip (tests/test_gpu_ops.py), which stays for clouds that do not fit in LDS.
#include "common.h"
// Barriers: every barrier of the per-cloud kernels orders LDS traffic only (their global accesses are input reads and write-only
// outputs), so they use lds_barrier() and the output stores stay in flight across them; k_cloud_scan communicates through global
// memory inside its workgroup and keeps __syncthreads().

namespace {

constexpr int CT = 256;  // threads per cloud

template <int NMAX>
struct CloudSmem {
    // table capacity: load factor <= 0.5, so every probe sequence meets an empty slot; small clouds get 4 slots per point, which also lets
    // a 28 x 28 pixel grid (784 cells) be addressed DIRECTLY (see Cells)
    static constexpr int HC = NMAX <= 256 ? 4 * NMAX : 2 * NMAX;
    // strides 2 and 4 hold at most NMAX keys too, but their direct grids are small (14 x 14, 7 x 7): 2 slots per key keep the load factor.
    // With that, no separate minimum-index table (a slot keeps the smallest index of its key itself) and the two scratch arrays folded
    // into arrays that are idle at the time, the struct is 31.1 KB for NMAX = 256: five clouds per CU (it was 41 KB: three)
    static constexpr int HC2 = 2 * NMAX;
    int4 q[NMAX];                        // floored point coordinates
    int tab1[HC], tab2[HC2], tab4[HC2];  // coordinate -> local voxel id, linear probing, -1 = empty
    int s_w[8];
    int box[2][6];                       // bounding boxes of the stride-1 and stride-2 voxels: lo[3], hi[3]
    // ---- k_cloud_count uses nothing below this line: its launch allocates offsetof(U1) bytes (12 KB for NMAX = 256: ten clouds per CU)
    int4 U1[NMAX], U2[NMAX], U4[NMAX];   // unique coordinates at strides 1, 2, 4 (first-occurrence order)
    int lid1[NMAX];                      // point -> local voxel id (stride 1)
    int par2[NMAX], par4[NMAX];          // stride-1 voxel -> stride-2 voxel ; stride-2 voxel -> stride-4 voxel (local ids)
    int first1[NMAX];                    // stride-1 voxel -> its first point (local index)
    int first2[NMAX], first4[NMAX];      // stride-2 voxel -> its first stride-1 voxel ; stride-4 voxel -> its first stride-2 voxel
    int cnt[NMAX + 1];                   // CSR scratch; before that, while the tables are built: slot of every key (lds_unique)
};

__device__ __forceinline__ int4 stride_coord(int4 c, int s) {
    return make_int4(c.x, floor_div_i(c.y, s) * s, floor_div_i(c.z, s) * s, floor_div_i(c.w, s) * s);
}

// exclusive prefix of `v` over the 256 threads of the block (4 waves); *total = block sum.  s_w needs 4 ints.
__device__ __forceinline__ int block_scan(int v, int* total, int* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    lds_barrier();  // s_w may still be read by the previous call
    if (lane == 63) s_w[w] = incl;
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = s_w[i];
        if (i < w) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

// When the bounding box of a table's keys has no more cells than the table has slots, a key's slot is its cell: no hashing, no probing,
// no key compare (flat 28 x 28 digit clouds: 784 / 196 / 49 cells at strides 1 / 2 / 4).  Otherwise (3-D clouds) the table is a
// linear-probing hash.  Both give the same voxel order and maps -- the slot never leaves the table.
struct Cells {
    int lo0, lo1, lo2, n1, n2, sh, direct;
};
__device__ __forceinline__ Cells cells_of(const int* box /*lo[3], hi[3]*/, int sh, int HC) {
    Cells c;
    c.lo0 = box[0]; c.lo1 = box[1]; c.lo2 = box[2]; c.sh = sh;
    const long long n0 = (((long long)box[3] - box[0]) >> sh) + 1, n1 = (((long long)box[4] - box[1]) >> sh) + 1,
                    n2 = (((long long)box[5] - box[2]) >> sh) + 1;
    c.n1 = (int)n1; c.n2 = (int)n2;
    c.direct = n0 > 0 && n1 > 0 && n2 > 0 && n0 <= HC && n1 <= HC && n2 <= HC && n0 * n1 * n2 <= HC;
    return c;
}
__device__ __forceinline__ int cell_of(const Cells& c, int4 key) {  // key inside the box, aligned to the stride
    return (((key.y - c.lo0) >> c.sh) * c.n1 + ((key.z - c.lo1) >> c.sh)) * c.n2 + ((key.w - c.lo2) >> c.sh);
}

// Unique keys[0..n) in first-occurrence order -> U[0..count); tab maps a coordinate to its index in U afterwards;
// lid[i] = index in U of keys[i]; first_idx[r] (nullable) = smallest i with keys[i] == U[r].  Returns count (block-uniform).
// While the keys are inserted a slot holds an INDEX of its key -- the claimer's at first, then the smallest among the duplicates
// (atomicMin; every value a slot ever holds indexes the same key, so concurrent probes compare against the right coordinate).
// `slot` is scratch for n ints.
template <int HC>
__device__ int lds_unique(const int4* keys, int n, const Cells cells, int* tab, int* slot, int4* U, int* lid, int* first_idx, int* s_w) {
    const int tid = threadIdx.x;
    for (int i = tid; i < HC; i += CT) tab[i] = -1;
    lds_barrier();
    for (int i = tid; i < n; i += CT) {
        const int4 key = keys[i];
        uint32_t h = cells.direct ? (uint32_t)cell_of(cells, key) : hash_coord(key) & (HC - 1);
        for (;;) {  // direct cells: the first slot is the key's own (empty, or a duplicate of the key)
            const int old = atomicCAS(&tab[h], -1, i);
            if (old == -1) break;
            if (coord_eq(keys[old], key)) {
                atomicMin(&tab[h], i);  // winner among duplicates = first occurrence
                break;
            }
            h = (h + 1) & (HC - 1);
        }
        slot[i] = (int)h;
    }
    lds_barrier();
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += CT) {  // stable compaction of the winners, 256 keys at a time
        const int i = c0 + tid;
        // (a slot rewritten by an earlier round holds a voxel id r < c0 <= i, under a tag or not: never i)
        const bool first = i < n && tab[slot[i]] == i;
        int tot;
        const int ex = block_scan(first ? 1 : 0, &tot, s_w);
        if (first) {
            const int r = base + ex;
            const int4 key = keys[i];
            U[r] = key;
            if (first_idx) first_idx[r] = i;
            // the slot now holds the voxel id (low 10 bits; NMAX <= 1024) under 21 bits of the key's hash, so a
            // hashed lookup rejects a foreign slot from the one word it has already read; keys live in U
            tab[slot[i]] = cells.direct ? r : (int)(((hash_coord(key) >> 11) & 0x1FFFFFu) << 10) | r;
        }
        base += tot;
    }
    lds_barrier();
    for (int i = tid; i < n; i += CT) lid[i] = tab[slot[i]] & 1023;
    lds_barrier();
    return base;
}

// A key outside the bounding box of the table's voxels cannot be present: no probe.  Flat clouds (MNIST digits: depth 0) lose every
// offset with a displacement along the flat axis this way -- 18 of the 27 kernel-map offsets, half of the pooling offsets.
template <int HC>
__device__ __forceinline__ int lds_lookup(const int* tab, const int4* U, const int* box, const Cells& cells, int4 key) {
    if (key.y < box[0] || key.y > box[3] || key.z < box[1] || key.z > box[4] || key.w < box[2] || key.w > box[5]) return -1;
    if (cells.direct) {
        const int e = tab[cell_of(cells, key)];
        return e < 0 ? -1 : e;
    }
    const uint32_t hk = hash_coord(key);
    const int tag = (int)((hk >> 11) & 0x1FFFFFu);
    uint32_t h = hk & (HC - 1);
    for (;;) {
        const int e = tab[h];
        if (e < 0) return -1;
        if ((e >> 10) == tag && coord_eq(U[e & 1023], key)) return e & 1023;  // the key itself is read only behind a matching tag
        h = (h + 1) & (HC - 1);
    }
}

// wave-reduce a thread's partial bounding box, one LDS atomic per wave and bound
__device__ __forceinline__ void box_merge(int* box, int lo0, int lo1, int lo2, int hi0, int hi1, int hi2) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, o, 64)); lo1 = min(lo1, __shfl_xor(lo1, o, 64)); lo2 = min(lo2, __shfl_xor(lo2, o, 64));
        hi0 = max(hi0, __shfl_xor(hi0, o, 64)); hi1 = max(hi1, __shfl_xor(hi1, o, 64)); hi2 = max(hi2, __shfl_xor(hi2, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&box[0], lo0); atomicMin(&box[1], lo1); atomicMin(&box[2], lo2);
        atomicMax(&box[3], hi0); atomicMax(&box[4], hi1); atomicMax(&box[5], hi2);
    }
}

struct CloudCounts {
    int m1, m2, m4;
};

// load + floor the cloud's points, then the three unique levels.  Sets *err |= 1 if a point's batch column is not b,
// |= 8 if the cloud has more than NMAX points (it is truncated: results are then meaningless but nothing goes out of bounds).
template <int NMAX>
__device__ CloudCounts build_cloud(CloudSmem<NMAX>& S, const float4* __restrict__ coords, int p0, int n_raw, int b, int* err, int* n_out, Cells* cells1_out,
                                   Cells* cells2_out) {
    constexpr int HC = CloudSmem<NMAX>::HC, HC2 = CloudSmem<NMAX>::HC2;
    const int tid = threadIdx.x;
    int n = n_raw;
    if (n > NMAX) {
        n = NMAX;
        if (tid == 0) atomicOr(err, 8);
    }
    if (tid < 12) S.box[tid / 6][tid % 6] = (tid % 6) < 3 ? 0x7FFFFFFF : (int)0x80000000;
    lds_barrier();
    bool bad = false;
    {
        int lo0 = 0x7FFFFFFF, lo1 = lo0, lo2 = lo0, hi0 = (int)0x80000000, hi1 = hi0, hi2 = hi0;
        for (int i = tid; i < n; i += CT) {
            const float4 c = coords[p0 + i];
            const int4 k = make_int4((int)floorf(c.x), (int)floorf(c.y), (int)floorf(c.z), (int)floorf(c.w));
            bad |= k.x != b;
            S.q[i] = k;
            lo0 = min(lo0, k.y); lo1 = min(lo1, k.z); lo2 = min(lo2, k.w);
            hi0 = max(hi0, k.y); hi1 = max(hi1, k.z); hi2 = max(hi2, k.w);
        }
        box_merge(S.box[0], lo0, lo1, lo2, hi0, hi1, hi2);  // box of the stride-1 voxels = box of the floored points
    }
    if (bad) atomicOr(err, 1);
    lds_barrier();
    // the stride-2 voxels' box follows from it (floor is monotone); likewise the stride-4 box, which only sizes the third table
    if (tid < 6) S.box[1][tid] = floor_div_i(S.box[0][tid], 2) * 2;
    lds_barrier();
    int box4[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) box4[j] = floor_div_i(S.box[1][j], 4) * 4;
    const Cells cells1 = cells_of(S.box[0], 0, HC), cells2 = cells_of(S.box[1], 1, HC2);
    *cells1_out = cells1;
    *cells2_out = cells2;
    const Cells cells4 = cells_of(box4, 2, HC2);
    CloudCounts c;
    c.m1 = lds_unique<HC>(S.q, n, cells1, S.tab1, S.cnt, S.U1, S.lid1, S.first1, S.s_w);
    // stride 2: keys = stride-2 coordinate of every stride-1 voxel (q is free again and serves as key storage)
    for (int v = tid; v < c.m1; v += CT) S.q[v] = stride_coord(S.U1[v], 2);
    lds_barrier();
    c.m2 = lds_unique<HC2>(S.q, c.m1, cells2, S.tab2, S.cnt, S.U2, S.par2, S.first2, S.s_w);
    for (int v = tid; v < c.m2; v += CT) S.q[v] = stride_coord(S.U2[v], 4);
    lds_barrier();
    c.m4 = lds_unique<HC2>(S.q, c.m2, cells4, S.tab4, S.cnt, S.U4, S.par4, S.first4, S.s_w);
    *n_out = n;
    return c;
}

// ---- pass A: voxel counts of every cloud at strides 1, 2, 4 + per-group minimum corner of the stride-4 coordinates ----
// Counting needs none of the ordered outputs of build_cloud: the number of distinct stride-s coordinates of the POINTS equals the number
// of stride-s voxels (a stride-2 voxel is the stride-2 coordinate of some stride-1 voxel, which is the coordinate of some point), and
// it is the number of inserts that found their slot empty.  So the three levels are three independent hash-set inserts per point
// -- two barriers in all instead of the ~40 of the ordered build -- and the minimum corner folds over the points' stride-4 coordinates.
template <int NMAX>
__global__ __launch_bounds__(CT) void k_cloud_count(const float4* __restrict__ coords, const int* __restrict__ pt_off, int B,
                                                    const int* __restrict__ cloud_group, int* __restrict__ cnt /*[3][B]*/, int* gmin /*[G][3]*/,
                                                    int* err, int* __restrict__ cmin /*[3][B] or null*/, int* __restrict__ vox_at /*[S][B] or null*/,
                                                    int S_sites) {
    constexpr int HC = CloudSmem<NMAX>::HC, HC2 = CloudSmem<NMAX>::HC2;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    CloudSmem<NMAX>& S = *reinterpret_cast<CloudSmem<NMAX>*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = pt_off[b];
    int n = pt_off[b + 1] - p0;
    if (n > NMAX) {
        n = NMAX;
        if (tid == 0) atomicOr(err, 8);
    }
    for (int i = tid; i < HC; i += CT) S.tab1[i] = -1;
    for (int i = tid; i < HC2; i += CT) S.tab2[i] = S.tab4[i] = -1;
    if (tid < 6) S.s_w[tid] = tid < 3 ? 0 : 0x7F7F7F7F;
    if (vox_at)  // this cloud's column of the dense table starts empty (k_cloud_emit fills it in): no memset node
        for (int s = tid; s < S_sites; s += CT) vox_at[(int64_t)s * B + b] = -1;
    bool bad = false;
    for (int i = tid; i < n; i += CT) {
        const float4 c = coords[p0 + i];
        const int4 k = make_int4((int)floorf(c.x), (int)floorf(c.y), (int)floorf(c.z), (int)floorf(c.w));
        bad |= k.x != b;
        S.q[i] = k;
    }
    if (bad) atomicOr(err, 1);
    lds_barrier();
    const int big = 0x7F7F7F7F;
    int a = big, d = big, e = big;
    int won1 = 0, won2 = 0, won4 = 0;
    for (int i = tid; i < n; i += CT) {
        const int4 k1 = S.q[i], k2 = stride_coord(k1, 2), k4 = stride_coord(k1, 4);
        a = min(a, k4.y); d = min(d, k4.z); e = min(e, k4.w);
        uint32_t h = hash_coord(k1) & (HC - 1);
        for (;;) {
            const int old = atomicCAS(&S.tab1[h], -1, i);
            if (old == -1) { ++won1; break; }
            if (coord_eq(S.q[old], k1)) break;
            h = (h + 1) & (HC - 1);
        }
        h = hash_coord(k2) & (HC2 - 1);
        for (;;) {
            const int old = atomicCAS(&S.tab2[h], -1, i);
            if (old == -1) { ++won2; break; }
            if (coord_eq(stride_coord(S.q[old], 2), k2)) break;
            h = (h + 1) & (HC2 - 1);
        }
        h = hash_coord(k4) & (HC2 - 1);
        for (;;) {
            const int old = atomicCAS(&S.tab4[h], -1, i);
            if (old == -1) { ++won4; break; }
            if (coord_eq(stride_coord(S.q[old], 4), k4)) break;
            h = (h + 1) & (HC2 - 1);
        }
    }
    won1 = wave_sum_i(won1); won2 = wave_sum_i(won2); won4 = wave_sum_i(won4);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = min(a, __shfl_xor(a, o, 64));
        d = min(d, __shfl_xor(d, o, 64));
        e = min(e, __shfl_xor(e, o, 64));
    }
    if ((tid & 63) == 0) {
        atomicAdd(&S.s_w[0], won1);
        atomicAdd(&S.s_w[1], won2);
        atomicAdd(&S.s_w[2], won4);
        atomicMin(&S.s_w[3], a);
        atomicMin(&S.s_w[4], d);
        atomicMin(&S.s_w[5], e);
    }
    lds_barrier();
    if (tid < 3) {
        cnt[tid * B + b] = S.s_w[tid];
        // one global atomic per cloud and axis, and only when it can lower the group's corner (most clouds share it): the values only
        // decrease, so a stale read costs an unnecessary atomic at worst
        const int m = S.s_w[3 + tid];
        if (cmin) cmin[tid * B + b] = m;  // folded per group by k_cloud_scan
        else if (gmin && m != big) {
            int* dst = &gmin[(cloud_group ? cloud_group[b] : 0) * 3 + tid];
            if (m < __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMin(dst, m);
        }
    }
}

// ---- scan: seg[l][0..B] = exclusive prefix of cnt[l][0..B), counts[l] = total, for the three levels (one block) ----
constexpr int GMIN_LDS_GROUPS = 1024;  // groups whose minimum corners k_cloud_scan folds in LDS (more: global atomics after a memset)
__global__ void k_cloud_scan(const int* __restrict__ cnt, int B, int* __restrict__ seg /*[3][B+1]*/, int* __restrict__ counts /*[3]*/,
                             const int* __restrict__ group_cloud_off, int G, int* __restrict__ gro /*[3][G+1] or null*/,
                             const int* __restrict__ cmin /*[3][B] or null*/, const int* __restrict__ cloud_group, int* __restrict__ gmin /*[G][3]*/,
                             const int* __restrict__ gmin_floor /*[G][3] or null*/) {
    __shared__ int s_w[48];
    __shared__ int s_gmin[GMIN_LDS_GROUPS * 3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (cmin) {  // gmin[g][a] = min over the clouds of group g (G <= GMIN_LDS_GROUPS, checked by the host)
        // (a caller that holds only PART of a group's clouds brings the corner of the rest: dense() places by the whole group's minimum)
        for (int i = threadIdx.x; i < G * 3; i += 1024) s_gmin[i] = gmin_floor ? gmin_floor[i] : 0x7F7F7F7F;
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * B; i += 1024) {
            const int a = i / B, b = i - a * B;
            int* dst = &s_gmin[(cloud_group ? cloud_group[b] : 0) * 3 + a];
            const int v = cmin[i];
            if (v < *(volatile int*)dst) atomicMin(dst, v);  // values only decrease: a stale read costs a needless atomic at worst
        }
        __syncthreads();
        for (int i = threadIdx.x; i < G * 3; i += 1024) gmin[i] = s_gmin[i];
    }
    // One pass for the three levels: thread t takes the clouds [t per, (t + 1) per) of every level, sums them, the 1024 partial sums
    // are scanned across the block (wave shuffles + 16 wave totals per level, ONE barrier), and the thread walks its clouds again to
    // write their offsets (the second read hits L2).
    const int per = (B + 1023) / 1024, b0 = min((int)threadIdx.x * per, B), b1 = min(b0 + per, B);
    int part[3] = {0, 0, 0};
#pragma unroll
    for (int l = 0; l < 3; ++l)
        for (int i = b0; i < b1; ++i) part[l] += cnt[l * B + i];
    int incl[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        incl[l] = part[l];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl[l], o, 64);
            if (lane >= o) incl[l] += t;
        }
        if (lane == 63) s_w[l * 16 + w] = incl[l];
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        int wbase = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int x = s_w[l * 16 + j];
            if (j < w) wbase += x;
            tot += x;
        }
        int run = wbase + incl[l] - part[l];
        for (int i = b0; i < b1; ++i) {
            seg[l * (B + 1) + i] = run;
            run += cnt[l * B + i];
        }
        if (threadIdx.x == 0) {
            seg[l * (B + 1) + B] = tot;
            counts[l] = tot;
        }
    }
    if (gro) {  // first row of every group of clouds at each stride (what the per-group reductions of the feature kernels walk)
        __syncthreads();  // the offsets written above, read back through global memory inside the workgroup
        for (int i = threadIdx.x; i < 3 * (G + 1); i += 1024) {
            const int l = i / (G + 1), g = i - l * (G + 1);
            gro[i] = seg[l * (B + 1) + group_cloud_off[g]];
        }
    }
}

struct CloudOut {
    int4 *C1, *C2, *C4;            // [bound,4] coordinates per stride
    int *inv1, *uidx1;             // [N] point -> stride-1 row ; [bound] stride-1 row -> first point
    int *vox_off, *vox_pts;        // [bound+1], [N]: points of every stride-1 voxel, ascending
    int *par2, *uidx2;             // [bound] stride-1 row -> stride-2 row ; stride-2 row -> first stride-1 row
    int *par4, *uidx4;             // [bound] stride-2 row -> stride-4 row ; stride-4 row -> first stride-2 row
    int *nbr1, *nbr2;              // [27][ld] k=3 kernel maps at strides 1 and 2
    int *kid2, *kid4;              // [8][ld] k=2 pooling maps 1->2 and 2->4
    int *site, *vox_at;            // dense placement of stride 4: [bound], [S][B] (nullable pair)
    const int* gmin;               // [G][3]
    const float* feats;            // [N][3] point features and F0 [bound][3] = their mean per stride-1 voxel (nullable pair)
    float* F0;
    int64_t ld;
    int E0, E1, E2;
};

__device__ __forceinline__ void k3_offset(int k, int ts, int* d0, int* d1, int* d2) {
    *d0 = (k % 3 - 1) * ts;
    *d1 = ((k / 3) % 3 - 1) * ts;
    *d2 = (k / 9 - 1) * ts;
}

// ---- pass B: rebuild the tables and emit every map at its global rows ----
// The scan folded into pass B (cnt != null): every cloud's workgroup sums the counts of the clouds before it itself (B / 256 coalesced L2
// reads per thread and level) and folds its group's minimum corner from the per-cloud corners, instead of waiting for a one-workgroup
// scan kernel between the two passes -- five dependent memory round trips, ~15 us of an otherwise idle GPU per step.  Who writes what:
// cloud b its own seg[l][b]; the last cloud also seg[l][B] and counts[l]; the cloud that STARTS group g (group_cloud_off[g] == b; the
// last cloud for groups that start at B) writes gro[l][g] and gmin[g].  cloud_group[b] must be the group whose range holds b.
struct ScanIn {
    const int* cnt;         // [3][B] voxel counts of pass A (null: seg / gmin come from k_cloud_scan)
    const int* cmin;        // [3][B] per-cloud minimum corners of pass A (null without dense placement)
    const int* gco;         // [G+1] group_cloud_off (null: no per-group outputs, one group)
    const int* gmin_floor;  // [G][3] or null
    int* seg;               // [3][B+1]
    int* counts;            // [3]
    int* gro;               // [3][G+1] or null
    int* gmin;              // [G][3] or null
    int G;
};

template <int NMAX>
__global__ __launch_bounds__(CT) void k_cloud_emit(const float4* __restrict__ coords, const int* __restrict__ pt_off, int B,
                                                   const int* __restrict__ cloud_group, const int* __restrict__ seg /*[3][B+1]*/, CloudOut o,
                                                   int* err, ScanIn si) {
    constexpr int HC = CloudSmem<NMAX>::HC, HC2 = CloudSmem<NMAX>::HC2;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    CloudSmem<NMAX>& S = *reinterpret_cast<CloudSmem<NMAX>*>(smem_raw);
    __shared__ int s_scan[4 * 6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = pt_off[b];
    int r1, r2, r4, gm0 = 0, gm1 = 0, gm2 = 0;
    if (si.cnt) {
        const int big = 0x7F7F7F7F;
        const int g = cloud_group ? cloud_group[b] : 0;
        int v[6] = {0, 0, 0, big, big, big};
        for (int i = tid; i < b; i += CT) {
            v[0] += si.cnt[i];
            v[1] += si.cnt[B + i];
            v[2] += si.cnt[2 * B + i];
        }
        if (si.cmin) {  // minimum corner of this cloud's group (its clouds are contiguous: group_cloud_off; without it, one group)
            const int c0 = si.gco ? si.gco[g] : 0, c1 = si.gco ? si.gco[g + 1] : B;
            for (int i = c0 + tid; i < c1; i += CT) {
                v[3] = min(v[3], si.cmin[i]);
                v[4] = min(v[4], si.cmin[B + i]);
                v[5] = min(v[5], si.cmin[2 * B + i]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = wave_sum_i(v[q]);
#pragma unroll
        for (int q = 3; q < 6; ++q)
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) v[q] = min(v[q], __shfl_xor(v[q], sh, 64));
        if ((tid & 63) == 0)
#pragma unroll
            for (int q = 0; q < 6; ++q) s_scan[(tid >> 6) * 6 + q] = v[q];
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = s_scan[q] + s_scan[6 + q] + s_scan[12 + q] + s_scan[18 + q];
#pragma unroll
        for (int q = 3; q < 6; ++q) v[q] = min(min(s_scan[q], s_scan[6 + q]), min(s_scan[12 + q], s_scan[18 + q]));
        r1 = v[0]; r2 = v[1]; r4 = v[2];
        if (si.cmin && si.gmin_floor) {
            v[3] = min(v[3], si.gmin_floor[g * 3 + 0]);
            v[4] = min(v[4], si.gmin_floor[g * 3 + 1]);
            v[5] = min(v[5], si.gmin_floor[g * 3 + 2]);
        }
        gm0 = v[3]; gm1 = v[4]; gm2 = v[5];
        const bool last = b == B - 1;
        if (tid < 3) {
            const int r = tid == 0 ? r1 : tid == 1 ? r2 : r4;
            si.seg[tid * (B + 1) + b] = r;
            if (last) {
                const int tot = r + si.cnt[tid * B + b];
                si.seg[tid * (B + 1) + B] = tot;
                si.counts[tid] = tot;
            }
        }
        if (si.gco) {  // per-group outputs of the groups that start at this cloud (or, for the last cloud, behind it)
            for (int q = tid; q <= si.G; q += CT) {
                const int start = si.gco[q];
                const bool mine = start == b, tail = last && start >= B;
                if (!mine && !tail) continue;
                if (si.gro) {
                    si.gro[q] = mine ? r1 : r1 + si.cnt[b];
                    si.gro[(si.G + 1) + q] = mine ? r2 : r2 + si.cnt[B + b];
                    si.gro[2 * (si.G + 1) + q] = mine ? r4 : r4 + si.cnt[2 * B + b];
                }
                if (si.gmin && q < si.G) {
                    const bool own = mine && q == g;  // a group that starts here but is not this cloud's is empty
                    for (int a = 0; a < 3; ++a)
                        si.gmin[q * 3 + a] = own ? (a == 0 ? gm0 : a == 1 ? gm1 : gm2) : (si.gmin_floor ? si.gmin_floor[q * 3 + a] : big);
                }
            }
        } else if (si.gmin && b == 0 && tid < 3) {
            si.gmin[tid] = tid == 0 ? gm0 : tid == 1 ? gm1 : gm2;
        }
    }
    int n;
    Cells cells1, cells2;
    const CloudCounts c = build_cloud<NMAX>(S, coords, p0, pt_off[b + 1] - p0, b, err, &n, &cells1, &cells2);
    if (!si.cnt) {
        r1 = seg[b]; r2 = seg[(B + 1) + b]; r4 = seg[2 * (B + 1) + b];
    }

    // stride 1: coordinates, first point, inverse map
    for (int v = tid; v < c.m1; v += CT) {
        o.C1[r1 + v] = S.U1[v];
        o.uidx1[r1 + v] = p0 + S.first1[v];
        o.par2[r1 + v] = r2 + S.par2[v];
    }
    for (int i = tid; i < n; i += CT) o.inv1[p0 + i] = r1 + S.lid1[i];
    // voxel -> points CSR (local counting sort; the tiny lists are put in ascending point order afterwards)
    for (int v = tid; v <= c.m1; v += CT) S.cnt[v] = 0;
    lds_barrier();
    for (int i = tid; i < n; i += CT) atomicAdd(&S.cnt[S.lid1[i]], 1);
    lds_barrier();
    int* cursor = S.tab4;  // the stride-4 table is not looked up after build_cloud: NMAX write cursors fit in its 2 NMAX slots
    {   // exclusive scan of cnt[0..m1) in place, 256 at a time
        int carry = 0;
        for (int c0 = 0; c0 < c.m1; c0 += CT) {
            const int v = c0 + tid;
            const int x = v < c.m1 ? S.cnt[v] : 0;
            int tot;
            const int ex = block_scan(x, &tot, S.s_w);
            if (v < c.m1) {
                S.cnt[v] = carry + ex;
                cursor[v] = 0;
            }
            carry += tot;
        }
        if (tid == 0) S.cnt[c.m1] = carry;
    }
    lds_barrier();
    for (int v = tid; v < c.m1; v += CT) o.vox_off[r1 + v] = p0 + S.cnt[v];
    if (tid == 0) o.vox_off[r1 + c.m1] = p0 + n;  // = the next cloud's first entry; the last cloud closes the array
    int* lpts = reinterpret_cast<int*>(S.q);       // q is free: local point lists
    for (int i = tid; i < n; i += CT) {
        const int v = S.lid1[i];
        lpts[S.cnt[v] + atomicAdd(&cursor[v], 1)] = i;
    }
    lds_barrier();
    for (int v = tid; v < c.m1; v += CT) {
        const int a = S.cnt[v], e = S.cnt[v + 1];
        for (int i = a + 1; i < e; ++i) {
            int x = lpts[i], j = i - 1;
            while (j >= a && lpts[j] > x) {
                lpts[j + 1] = lpts[j];
                --j;
            }
            lpts[j + 1] = x;
        }
        for (int i = a; i < e; ++i) o.vox_pts[p0 + i] = p0 + lpts[i];
        if (o.F0) {  // TensorField.sparse() averages the features of a voxel's points: k_voxel_mean_fwd's sum, in its order (C = 3)
            float f[3] = {0.f, 0.f, 0.f};
            for (int i = a; i < e; ++i) {
                const float* src = o.feats + (int64_t)(p0 + lpts[i]) * 3;
                f[0] += src[0]; f[1] += src[1]; f[2] += src[2];
            }
            float* dst = o.F0 + (int64_t)(r1 + v) * 3;
            dst[0] = f[0] / (float)(e - a); dst[1] = f[1] / (float)(e - a); dst[2] = f[2] / (float)(e - a);
        }
    }
    // stride 2 and 4: coordinates, parents, first rows
    for (int u = tid; u < c.m2; u += CT) {
        o.C2[r2 + u] = S.U2[u];
        o.par4[r2 + u] = r4 + S.par4[u];
        o.uidx2[r2 + u] = r1 + S.first2[u];
    }
    for (int t = tid; t < c.m4; t += CT) {
        o.C4[r4 + t] = S.U4[t];
        o.uidx4[r4 + t] = r2 + S.first4[t];
    }

    // k=3 kernel maps: one (offset, voxel) probe per thread iteration, writes coalesced along the voxel index
    for (int idx = tid; idx < 27 * c.m1; idx += CT) {
        const int k = idx / c.m1, v = idx - k * c.m1;
        int d0, d1, d2;
        k3_offset(k, 1, &d0, &d1, &d2);
        const int4 u = S.U1[v];
        const int r = lds_lookup<HC>(S.tab1, S.U1, S.box[0], cells1, make_int4(u.x, u.y + d0, u.z + d1, u.w + d2));
        o.nbr1[(int64_t)k * o.ld + r1 + v] = r < 0 ? -1 : r1 + r;
    }
    for (int idx = tid; idx < 27 * c.m2; idx += CT) {
        const int k = idx / c.m2, v = idx - k * c.m2;
        int d0, d1, d2;
        k3_offset(k, 2, &d0, &d1, &d2);
        const int4 u = S.U2[v];
        const int r = lds_lookup<HC2>(S.tab2, S.U2, S.box[1], cells2, make_int4(u.x, u.y + d0, u.z + d1, u.w + d2));
        o.nbr2[(int64_t)k * o.ld + r2 + v] = r < 0 ? -1 : r2 + r;
    }
    // k=2 pooling maps: children of a stride-2s voxel at offsets {0,1}*s, k = e0 + 2 e1 + 4 e2
    for (int idx = tid; idx < 8 * c.m2; idx += CT) {
        const int k = idx / c.m2, v = idx - k * c.m2;
        const int4 u = S.U2[v];
        const int r = lds_lookup<HC>(S.tab1, S.U1, S.box[0], cells1, make_int4(u.x, u.y + (k & 1), u.z + ((k >> 1) & 1), u.w + (k >> 2)));
        o.kid2[(int64_t)k * o.ld + r2 + v] = r < 0 ? -1 : r1 + r;
    }
    for (int idx = tid; idx < 8 * c.m4; idx += CT) {
        const int k = idx / c.m4, v = idx - k * c.m4;
        const int4 u = S.U4[v];
        const int r = lds_lookup<HC2>(S.tab2, S.U2, S.box[1], cells2, make_int4(u.x, u.y + 2 * (k & 1), u.z + 2 * ((k >> 1) & 1), u.w + 2 * (k >> 2)));
        o.kid4[(int64_t)k * o.ld + r4 + v] = r < 0 ? -1 : r2 + r;
    }
    // dense() placement of the stride-4 voxels: site = flat index relative to the group's minimum corner
    if (o.site) {
        const int g = cloud_group ? cloud_group[b] : 0;
        const int m0 = si.cnt ? gm0 : o.gmin[g * 3 + 0], m1 = si.cnt ? gm1 : o.gmin[g * 3 + 1], m2 = si.cnt ? gm2 : o.gmin[g * 3 + 2];
        for (int t = tid; t < c.m4; t += CT) {
            const int4 u = S.U4[t];
            const int a = (u.y - m0) / 4, d = (u.z - m1) / 4, e = (u.w - m2) / 4;
            if (a < 0 || a >= o.E0 || d < 0 || d >= o.E1 || e < 0 || e >= o.E2) {
                atomicOr(err, 2);  // MinkowskiEngine's dense() raises here
                o.site[r4 + t] = -1;
            } else {
                const int s = (a * o.E1 + d) * o.E2 + e;
                o.site[r4 + t] = s;
                if (o.vox_at) o.vox_at[(int64_t)s * B + b] = r4 + t;
            }
        }
    }
}

template <int NMAX>
int launch_cloud_maps(const float4* coords, const int* pt_off, int B, const int* cloud_group, int* cnt, int* seg, int* counts, const CloudOut& o,
                      int* gmin, int* err, const int* group_cloud_off, int G, int* gro, const int* gmin_floor, hipStream_t st) {
    int* cmin = (gmin && G <= GMIN_LDS_GROUPS) ? cnt + 3 * (int64_t)B : nullptr;
    const size_t smem = sizeof(CloudSmem<NMAX>), smem_count = offsetof(CloudSmem<NMAX>, U1);
    static_assert(offsetof(CloudSmem<NMAX>, U1) % 16 == 0 && (NMAX > 256 || sizeof(CloudSmem<NMAX>) <= 32 * 1024), "LDS layout");
    if (smem > 64 * 1024) {
        DD3D_HIP(hipFuncSetAttribute((const void*)k_cloud_count<NMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        DD3D_HIP(hipFuncSetAttribute((const void*)k_cloud_emit<NMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    // The scan runs inside pass B (see ScanIn) unless the per-group corners need the global-atomic path (more groups than
    // GMIN_LDS_GROUPS) or dense placement comes without group_cloud_off (the groups' clouds are then not known to be contiguous).
    static const bool fuse_env = !(getenv("DD3D_CLOUD_SCAN_KERNEL") && atoi(getenv("DD3D_CLOUD_SCAN_KERNEL")));  // 1: the separate scan kernel (measurements)
    const bool fuse = fuse_env && (!gmin || (cmin && (group_cloud_off || G == 1)));
    ScanIn si = {};
    if (fuse) {
        si.cnt = cnt; si.cmin = gmin ? cmin : nullptr; si.gco = group_cloud_off; si.gmin_floor = gmin_floor; si.seg = seg; si.counts = counts;
        si.gro = gro; si.gmin = gmin; si.G = G;
    }
    k_cloud_count<NMAX><<<dim3(B), CT, smem_count, st>>>(coords, pt_off, B, cloud_group, cnt, gmin, err, cmin, o.vox_at, o.E0 * o.E1 * o.E2);
    if (!fuse) k_cloud_scan<<<1, 1024, 0, st>>>(cnt, B, seg, counts, group_cloud_off, G, gro, cmin, cloud_group, gmin, gmin_floor);
    k_cloud_emit<NMAX><<<dim3(B), CT, smem, st>>>(coords, pt_off, B, cloud_group, seg, o, err, si);
    DD3D_CHECK_LAUNCH();
    return 0;
}

}  // namespace

DD3D_API int dd3d_cloud_maps_max_points(void) { return 1024; }

DD3D_API int64_t dd3d_cloud_maps_ws_bytes(int B) { return align_up64((int64_t)(B > 0 ? B : 1) * 6 * 4, 256); }  // counts and minimum corners, [3][B] each

// See include/dd3d.h.  All outputs are sized by `bound` (>= number of points); rows beyond the counts are not written.
static int cloud_maps_impl(const float* coords, const int32_t* pt_off, int B, int max_points, int64_t bound, const int32_t* cloud_group, int G,
                           int E0, int E1, int E2, int32_t* counts /*[3]*/, int32_t* seg /*[3][B+1]*/, int32_t* C1, int32_t* C2, int32_t* C4,
                           int32_t* inv1, int32_t* uidx1, int32_t* vox_off, int32_t* vox_pts, int32_t* par2, int32_t* uidx2, int32_t* par4,
                           int32_t* uidx4, int32_t* nbr1, int32_t* nbr2, int32_t* kid2, int32_t* kid4, int64_t ld, int32_t* gmin, int32_t* site,
                           int32_t* vox_at, const int32_t* group_cloud_off, int32_t* group_row_off /*[3][G+1]*/, const int32_t* gmin_floor /*[G][3] or null*/,
                           int32_t* err_dev, void* ws, int64_t ws_bytes, void* stream, const float* feats, float* F0) {
    if ((feats == nullptr) != (F0 == nullptr)) return DD3D_EARG;
    if (B < 0 || max_points <= 0 || max_points > dd3d_cloud_maps_max_points() || bound < 0 || ld < bound || G <= 0 || !counts || !seg || !err_dev)
        return DD3D_EARG;
    if (bound >= (1ll << 31) / 32) return DD3D_ESIZE;
    if (ws_bytes < dd3d_cloud_maps_ws_bytes(B) || !ws) return DD3D_EWS;
    const bool dense = site != nullptr;
    if (dense && (!gmin || E0 <= 0 || E1 <= 0 || E2 <= 0)) return DD3D_EARG;
    if (vox_at && !dense) return DD3D_EARG;
    if (group_row_off && !group_cloud_off) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    // the dense table is cleared per cloud by k_cloud_count and the group corners are folded by k_cloud_scan: no memset nodes in a
    // captured step (only more groups than the scan holds in LDS, or no clouds at all, take the memset)
    if (dense && (G > GMIN_LDS_GROUPS || B == 0)) {
        if (gmin_floor) DD3D_HIP(hipMemcpyAsync(gmin, gmin_floor, (size_t)G * 3 * 4, hipMemcpyDeviceToDevice, st));
        else DD3D_HIP(hipMemsetAsync(gmin, 0x7F, (size_t)G * 3 * 4, st));
    }
    if (B == 0) {
        DD3D_HIP(hipMemsetAsync(counts, 0, 12, st));
        DD3D_HIP(hipMemsetAsync(seg, 0, 12, st));
        if (group_row_off) DD3D_HIP(hipMemsetAsync(group_row_off, 0, (size_t)3 * (G + 1) * 4, st));
        if (vox_off) DD3D_HIP(hipMemsetAsync(vox_off, 0, 4, st));
        return 0;
    }
    if (!coords || !pt_off || !C1 || !C2 || !C4 || !inv1 || !uidx1 || !vox_off || !vox_pts || !par2 || !uidx2 || !par4 || !uidx4 || !nbr1 || !nbr2 ||
        !kid2 || !kid4)
        return DD3D_EARG;
    CloudOut o;
    o.C1 = (int4*)C1; o.C2 = (int4*)C2; o.C4 = (int4*)C4;
    o.inv1 = inv1; o.uidx1 = uidx1; o.vox_off = vox_off; o.vox_pts = vox_pts;
    o.par2 = par2; o.uidx2 = uidx2; o.par4 = par4; o.uidx4 = uidx4;
    o.nbr1 = nbr1; o.nbr2 = nbr2; o.kid2 = kid2; o.kid4 = kid4;
    o.site = site; o.vox_at = vox_at; o.gmin = gmin;
    o.feats = feats; o.F0 = F0;
    o.ld = ld; o.E0 = E0; o.E1 = E1; o.E2 = E2;
    int* cnt = (int*)ws;
    if (max_points <= 256)
        return launch_cloud_maps<256>((const float4*)coords, pt_off, B, cloud_group, cnt, seg, counts, o, dense ? gmin : nullptr, err_dev, group_cloud_off, G, group_row_off, gmin_floor, st);
    return launch_cloud_maps<1024>((const float4*)coords, pt_off, B, cloud_group, cnt, seg, counts, o, dense ? gmin : nullptr, err_dev, group_cloud_off, G, group_row_off, gmin_floor, st);
}

DD3D_API int dd3d_cloud_maps(const float* coords, const int32_t* pt_off, int B, int max_points, int64_t bound, const int32_t* cloud_group, int G,
                             int E0, int E1, int E2, int32_t* counts /*[3]*/, int32_t* seg /*[3][B+1]*/, int32_t* C1, int32_t* C2, int32_t* C4,
                             int32_t* inv1, int32_t* uidx1, int32_t* vox_off, int32_t* vox_pts, int32_t* par2, int32_t* uidx2, int32_t* par4,
                             int32_t* uidx4, int32_t* nbr1, int32_t* nbr2, int32_t* kid2, int32_t* kid4, int64_t ld, int32_t* gmin, int32_t* site,
                             int32_t* vox_at, const int32_t* group_cloud_off, int32_t* group_row_off /*[3][G+1]*/, const int32_t* gmin_floor /*[G][3] or null*/,
                             int32_t* err_dev, void* ws, int64_t ws_bytes, void* stream) {
    return cloud_maps_impl(coords, pt_off, B, max_points, bound, cloud_group, G, E0, E1, E2, counts, seg, C1, C2, C4, inv1, uidx1, vox_off, vox_pts, par2, uidx2,
                           par4, uidx4, nbr1, nbr2, kid2, kid4, ld, gmin, site, vox_at, group_cloud_off, group_row_off, gmin_floor, err_dev, ws, ws_bytes,
                           stream, nullptr, nullptr);
}

// dd3d_cloud_maps + dd3d_voxel_mean_fwd for 3-channel point features in the same two launches: F0[bound][3] = mean of feats3[N][3] over the
// points of every stride-1 voxel (the sum in ascending point order, as dd3d_voxel_mean_fwd: same bits).  Same arguments otherwise.
DD3D_API int dd3d_cloud_maps_f0(const float* coords, const int32_t* pt_off, int B, int max_points, int64_t bound, const int32_t* cloud_group, int G,
                                int E0, int E1, int E2, int32_t* counts /*[3]*/, int32_t* seg /*[3][B+1]*/, int32_t* C1, int32_t* C2, int32_t* C4,
                                int32_t* inv1, int32_t* uidx1, int32_t* vox_off, int32_t* vox_pts, int32_t* par2, int32_t* uidx2, int32_t* par4,
                                int32_t* uidx4, int32_t* nbr1, int32_t* nbr2, int32_t* kid2, int32_t* kid4, int64_t ld, int32_t* gmin, int32_t* site,
                                int32_t* vox_at, const int32_t* group_cloud_off, int32_t* group_row_off /*[3][G+1]*/, const int32_t* gmin_floor /*[G][3] or null*/,
                                int32_t* err_dev, void* ws, int64_t ws_bytes, void* stream, const float* feats3, float* F0) {
    if (!feats3 || !F0) return DD3D_EARG;
    return cloud_maps_impl(coords, pt_off, B, max_points, bound, cloud_group, G, E0, E1, E2, counts, seg, C1, C2, C4, inv1, uidx1, vox_off, vox_pts, par2, uidx2,
                           par4, uidx4, nbr1, nbr2, kid2, kid4, ld, gmin, site, vox_at, group_cloud_off, group_row_off, gmin_floor, err_dev, ws, ws_bytes,
                           stream, feats3, F0);
}
