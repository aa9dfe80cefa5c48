// Coordinate machinery: quantisation, voxel hash, first-occurrence unique, stride maps, kernel maps, pair lists.
// Replaces MinkowskiEngine's CoordinateMapManager for the calls reached from models/MEConv.py:224,227,247
// (SURVEY.md N2, N4, N9).  All integer work, bit-exact by construction:
//   * winner among duplicate coordinates = minimum input row (atomicMin), so voxel rows come out in
//     first-occurrence order after a stable ballot compaction;
//   * kernel maps are dense neighbour tables nbr[k][out_row] (-1 = absent), from which canonical pair lists
//     (grouped by offset, sorted by out_row) are compacted with the same stable ballot scan.
// Every kernel takes a bound (array length) and an optional device-side count, so a whole chain of maps can be
// built without a host round trip.
#include "common.h"

namespace {

constexpr int TB = 256;

__device__ __forceinline__ int load_count(const int* n_dev, int64_t bound) { return n_dev ? min((int64_t)*n_dev, bound) : (int)bound; }

__global__ void k_floor_coords(const float4* __restrict__ coords, int64_t N, int4* __restrict__ q) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= N) return;
    float4 c = coords[p];
    q[p] = make_int4((int)floorf(c.x), (int)floorf(c.y), (int)floorf(c.z), (int)floorf(c.w));
}

__global__ void k_stride_coords(const int4* __restrict__ C, const int* n_dev, int64_t bound, int stride, int4* __restrict__ q) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= load_count(n_dev, bound)) return;
    int4 c = C[p];
    q[p] = make_int4(c.x, floor_div_i(c.y, stride) * stride, floor_div_i(c.z, stride) * stride, floor_div_i(c.w, stride) * stride);
}

// insert every key; tab_rep[h] = some row with that key (claim winner), tab_min[h] = minimum row with that key
__global__ void k_hash_insert(const int4* __restrict__ q, const int* n_dev, int64_t bound, int* tab_rep, int* tab_min, uint32_t mask,
                              int* __restrict__ slot_of) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= load_count(n_dev, bound)) return;
    int4 key = q[p];
    uint32_t h = hash_coord(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        int old = atomicCAS(&tab_rep[h], -1, (int)p);
        if (old == -1) break;
        if (coord_eq(q[old], key)) break;
        h = (h + 1) & mask;
    }
    slot_of[p] = (int)h;
    atomicMin(&tab_min[h], (int)p);
}

__global__ void k_first_count(const int* __restrict__ slot_of, const int* __restrict__ tab_min, const int* n_dev, int64_t bound,
                              int* __restrict__ block_counts) {
    __shared__ int s_w[4];
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    bool first = false;
    if (p < load_count(n_dev, bound)) first = tab_min[slot_of[p]] == (int)p;
    unsigned long long b = __ballot(first);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan, in place, of n ints by ONE block of 1024 threads; total -> *total_out (if non-null)
__global__ void k_scan_single(int* data, int64_t n, int* total_out) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        int v = i < n ? data[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int x = s_w[j];
            if (j < w) wbase += x;
            tot += x;
        }
        int carry = s_carry;
        if (i < n) data[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = s_carry;
}

__global__ void k_unique_fill(const int4* __restrict__ q, const int* __restrict__ slot_of, const int* __restrict__ tab_min, int* tab_rep,
                              const int* n_dev, int64_t bound, const int* __restrict__ block_off, int4* __restrict__ C, int* __restrict__ uidx) {
    __shared__ int s_w[4];
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    bool first = false;
    int slot = 0;
    if (p < load_count(n_dev, bound)) {
        slot = slot_of[p];
        first = tab_min[slot] == (int)p;
    }
    unsigned long long b = __ballot(first);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = __popcll(b);
    __syncthreads();
    int wbase = 0;
    for (int j = 0; j < w; ++j) wbase += s_w[j];
    if (first) {
        int rank = block_off[blockIdx.x] + wbase + __popcll(b & ((1ull << lane) - 1ull));
        C[rank] = q[p];
        uidx[rank] = (int)p;
        tab_rep[slot] = rank;  // from here on the table maps coordinate -> voxel row (keys live in C)
    }
}

__global__ void k_inverse(const int* __restrict__ slot_of, const int* __restrict__ tab_rep, const int* n_dev, int64_t bound, int* __restrict__ inv) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= load_count(n_dev, bound)) return;
    inv[p] = tab_rep[slot_of[p]];
}

__global__ void k_fill_i32(int* p, int64_t n, int v) {
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- voxel -> point CSR ---------------------------------------------------------------------
__global__ void k_count_inv(const int* __restrict__ inv, const int* n_dev, int64_t bound, int* cnt) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= load_count(n_dev, bound)) return;
    atomicAdd(&cnt[inv[p]], 1);
}

// block-level pieces of a large exclusive scan (1024 elements per block)
__global__ void k_scan_reduce(const int* __restrict__ in, int64_t n, int* __restrict__ bsum) {
    __shared__ int s_w[4];
    int64_t base = (int64_t)blockIdx.x * 1024;
    int v = 0;
    for (int j = 0; j < 4; ++j) {
        int64_t i = base + j * TB + threadIdx.x;
        if (i < n) v += in[i];
    }
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void k_scan_apply(const int* __restrict__ in, int64_t n, const int* __restrict__ boff, int* __restrict__ out) {
    __shared__ int s_w[5];
    int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;  // 4 consecutive elements per thread
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0;
        s += v[j];
    }
    int tot;
    int ex = block_excl_scan_256(s, &tot, s_w) + boff[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
    }
}

__global__ void k_csr_fill(const int* __restrict__ inv, const int* n_dev, int64_t bound, const int* __restrict__ off, int* cursor, int* __restrict__ pts) {
    int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= load_count(n_dev, bound)) return;
    int v = inv[p];
    int pos = off[v] + atomicAdd(&cursor[v], 1);
    pts[pos] = (int)p;
}

// lists are tiny (points sharing one voxel): insertion sort each so the order is ascending point index
__global__ void k_csr_sort(const int* __restrict__ off, const int* m_dev, int64_t bound, int* pts) {
    int64_t v = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (v >= load_count(m_dev, bound)) return;
    int a = off[v], b = off[v + 1];
    for (int i = a + 1; i < b; ++i) {
        int x = pts[i], j = i - 1;
        while (j >= a && pts[j] > x) {
            pts[j + 1] = pts[j];
            --j;
        }
        pts[j + 1] = x;
    }
}

// ---- kernel maps ----------------------------------------------------------------------------
__device__ __forceinline__ void region_offset(int k, int ksize, int ts, int* d0, int* d1, int* d2) {
    if (ksize & 1) {
        int h = ksize >> 1;
        *d0 = (k % ksize - h) * ts;
        *d1 = ((k / ksize) % ksize - h) * ts;
        *d2 = (k / (ksize * ksize) - h) * ts;
    } else {
        *d0 = (k % ksize) * ts;
        *d1 = ((k / ksize) % ksize) * ts;
        *d2 = (k / (ksize * ksize)) * ts;
    }
}

__global__ void k_kernel_map(const int4* __restrict__ Cq, const int* mq_dev, int64_t bound_q, const int* __restrict__ tab, uint32_t mask,
                             const int4* __restrict__ keys, int ksize, int ts, int K, int* __restrict__ nbr, int64_t ld) {
    int64_t o = (int64_t)blockIdx.x * TB + threadIdx.x;
    int k = blockIdx.y;
    int mq = load_count(mq_dev, bound_q);
    if (o >= ld) return;
    int r = -1;
    if (o < mq) {
        int d0, d1, d2;
        region_offset(k, ksize, ts, &d0, &d1, &d2);
        int4 c = Cq[o];
        r = table_lookup(tab, mask, keys, make_int4(c.x, c.y + d0, c.z + d1, c.w + d2));
    }
    nbr[(int64_t)k * ld + o] = r;
}

__global__ void k_pairs_count(const int* __restrict__ nbr, int64_t ld, const int* m_dev, int64_t bound, int* __restrict__ block_counts, int nblk) {
    __shared__ int s_w[4];
    int64_t o = (int64_t)blockIdx.x * TB + threadIdx.x;
    int k = blockIdx.y;
    bool v = o < load_count(m_dev, bound) && nbr[(int64_t)k * ld + o] >= 0;
    unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[(int64_t)k * nblk + blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void k_pairs_fill(const int* __restrict__ nbr, int64_t ld, const int* m_dev, int64_t bound, const int* __restrict__ block_off, int nblk,
                             int* __restrict__ in_rows, int* __restrict__ out_rows, int* __restrict__ offs, int K) {
    __shared__ int s_w[4];
    int64_t o = (int64_t)blockIdx.x * TB + threadIdx.x;
    int k = blockIdx.y;
    int r = -1;
    if (o < load_count(m_dev, bound)) r = nbr[(int64_t)k * ld + o];
    bool v = r >= 0;
    unsigned long long b = __ballot(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = __popcll(b);
    __syncthreads();
    int wbase = 0;
    for (int j = 0; j < w; ++j) wbase += s_w[j];
    int boff = block_off[(int64_t)k * nblk + blockIdx.x];
    if (v) {
        int pos = boff + wbase + __popcll(b & ((1ull << lane) - 1ull));
        in_rows[pos] = r;
        out_rows[pos] = (int)o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offs[k] = boff;  // offs[K] (total) is written by the scan
}

// ---- per-cloud row ranges -------------------------------------------------------------------
// rows are sorted by batch index (first-occurrence order of collated clouds); seg_off[b] = first row of cloud b
__global__ void k_segment_offsets(const int4* __restrict__ C, const int* m_dev, int64_t bound, int B, int* __restrict__ seg_off, int* err) {
    int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    int M = load_count(m_dev, bound);
    if (r > M) return;
    int prev = r == 0 ? -1 : C[r - 1].x;
    int cur = r == M ? B : C[r].x;
    if (r < M && (cur < prev || cur < 0 || cur >= B)) {
        atomicOr(err, 1);  // rows not grouped by ascending batch index, or batch index out of range
        return;
    }
    for (int b = prev + 1; b <= cur && b <= B; ++b) seg_off[b] = (int)r;
}

__global__ void k_min_coord(const int4* __restrict__ C, const int* m_dev, int64_t bound, const int* __restrict__ cloud_group, int* gmin /*[G,3]*/) {
    int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    const int big = 0x7F7F7F7F;
    int g = -1, a = big, b = big, d = big;
    if (r < load_count(m_dev, bound)) {
        int4 c = C[r];
        g = cloud_group ? cloud_group[c.x] : 0;
        a = c.y; b = c.z; d = c.w;
    }
    // rows are grouped, so a wave almost always sees one group: reduce in the wave, one atomic per wave
    int g0 = __builtin_amdgcn_readfirstlane(g);
    if (__all(g == g0 || g < 0) && g0 >= 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = min(a, __shfl_xor(a, o, 64));
            b = min(b, __shfl_xor(b, o, 64));
            d = min(d, __shfl_xor(d, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&gmin[g0 * 3 + 0], a);
            atomicMin(&gmin[g0 * 3 + 1], b);
            atomicMin(&gmin[g0 * 3 + 2], d);
        }
    } else if (g >= 0) {
        atomicMin(&gmin[g * 3 + 0], a);
        atomicMin(&gmin[g * 3 + 1], b);
        atomicMin(&gmin[g * 3 + 2], d);
    }
}

// dense() placement: site[v] = ((z-minz)/ts * E1 + (y-miny)/ts) * E2 + (x-minx)/ts ; vox_at[site][b] = v (site-major: the clouds of one site are contiguous)
__global__ void k_dense_sites(const int4* __restrict__ C, const int* m_dev, int64_t bound, const int* __restrict__ cloud_group, const int* __restrict__ gmin,
                              int ts, int E0, int E1, int E2, int B, int* __restrict__ site, int* __restrict__ vox_at, int* err) {
    int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (r >= load_count(m_dev, bound)) return;
    int4 c = C[r];
    int g = cloud_group ? cloud_group[c.x] : 0;
    int a = (c.y - gmin[g * 3 + 0]) / ts, b = (c.z - gmin[g * 3 + 1]) / ts, d = (c.w - gmin[g * 3 + 2]) / ts;
    if (a < 0 || a >= E0 || b < 0 || b >= E1 || d < 0 || d >= E2) {
        atomicOr(err, 2);  // ME's dense() would raise an index error here
        site[r] = -1;
        return;
    }
    int s = (a * E1 + b) * E2 + d;
    site[r] = s;
    if (vox_at) vox_at[(int64_t)s * B + c.x] = (int)r;
}

// ---- real-batch assembly on the device ------------------------------------------------------
// pt_off[b] = sum_{b' < b} min(points of cloud ids[b'], cap): ONE block scans the B sampled clouds; total -> *n_dev
__global__ void k_cloud_offsets(const int* __restrict__ pool_off, const int* __restrict__ ids, int B, int n_pool, int cap, int64_t bound,
                                int point_base, int* __restrict__ pt_off, int* __restrict__ n_dev, int* err) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        int b = base + threadIdx.x;
        int v = 0;
        if (b < B) {
            int id = ids[b];
            if (id < 0 || id >= n_pool) atomicOr(err, 4);  // sampled index outside the pool: the cloud comes out empty
            else v = min(pool_off[id + 1] - pool_off[id], cap);
        }
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int x = s_w[j];
            if (j < w) wbase += x;
            tot += x;
        }
        int carry = s_carry;
        // (clamped like the total below: behind an overfull batch every later cloud is empty, none starts outside the caller's buffers)
        if (b < B) pt_off[b] = (int)min((int64_t)(point_base + carry + wbase + incl - v), bound);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int total = point_base + s_carry;
        if (total > bound) {
            atomicOr(err, 4);  // the caller's bound is too small: rows beyond it are dropped
            total = (int)bound;
        }
        pt_off[B] = total;
        *n_dev = total;
    }
}

// one workgroup per sampled cloud: rows (b, p0, p1, p2) into coords and (p0, p1, p2) into feats, in pool order
__global__ void k_gather_clouds(const float* __restrict__ pool_xyz, const int* __restrict__ pool_off, const int* __restrict__ ids, int n_pool,
                                const int* __restrict__ pt_off, int B, int cloud_base, float quant, float4* __restrict__ coords, float* __restrict__ feats) {
    const int b = blockIdx.x;
    const int id = ids[b];
    if (id < 0 || id >= n_pool) return;
    const int dst0 = min(pt_off[b], pt_off[B]), n = min(pt_off[b + 1], pt_off[B]) - dst0;
    const float* src = pool_xyz + (int64_t)pool_off[id] * 3;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float x = src[i * 3 + 0], y = src[i * 3 + 1], z = src[i * 3 + 2];
        coords[dst0 + i] = make_float4((float)(cloud_base + b), x / quant, y / quant, z / quant);  // create_input_batch: coords / quantization_size
        float* f = feats + (int64_t)(dst0 + i) * 3;
        f[0] = x; f[1] = y; f[2] = z;
    }
}

// The head of a step in ONE launch (dd3d_assemble_batch): workgroups [0, syn_blocks) clamp the learnable points and write the owned
// synthetic clouds into the head of the merged batch (what k_syn_head of linear_loss.hip does, point for point), workgroup syn_blocks + b
// gathers sampled cloud b behind them (k_gather_clouds) after summing the point counts of the clouds before it ITSELF (what the
// one-workgroup k_cloud_offsets did in a launch of its own: B / 128 dependent index -> offset reads per thread, from L2), and the
// first gather workgroup clears `zero_words` ints at `zero` (the map builder's per-step count / error words).  Same outputs, bit for bit.
__global__ __launch_bounds__(128) void k_assemble_batch(float* __restrict__ xyz, int64_t n_points, int npts, const int* __restrict__ dest_cloud, int do_clamp,
                                                        float lo, float hi, int zero_first, float quant, int syn_blocks,
                                                        const float* __restrict__ pool_xyz, const int* __restrict__ pool_off, const int* __restrict__ ids,
                                                        int n_pool, int B, int cap, int64_t bound, int cloud_base, int point_base,
                                                        float4* __restrict__ coords, float* __restrict__ feats, int* __restrict__ pt_off /* + cloud_base */,
                                                        int* __restrict__ n_dev, int* err, int* __restrict__ zero, int zero_words) {
    __shared__ int s_w[2];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < syn_blocks) {
        float* cf = (float*)coords;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // (k_syn_head runs 256 points per workgroup)
            const int64_t p = (int64_t)blockIdx.x * 256 + half * 128 + tid;
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
            if (d < 0) continue;
            const int64_t r = (int64_t)d * npts + (p - c * npts);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cf[r * 4 + 1 + a] = v[a] / quant;
                feats[r * 3 + a] = v[a];
            }
        }
        return;
    }
    const int b = blockIdx.x - syn_blocks;
    if (b == 0)
        for (int i = tid; i < zero_words; i += 128) zero[i] = 0;
    // points of the clouds before this one (an index outside the pool: an empty cloud, flagged by the cloud's own workgroup)
    int sum = 0;
    for (int i0 = tid; i0 < b; i0 += 128 * 8) {  // eight clouds of this lane at a time: their ids, then their 16 offsets, in flight together
        int id[8], lo8[8], hi8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + 128 * q;
            id[q] = i < b ? ids[i] : -1;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool okq = id[q] >= 0 && id[q] < n_pool;
            lo8[q] = pool_off[okq ? id[q] : 0];
            hi8[q] = okq ? pool_off[id[q] + 1] : lo8[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += min(hi8[q] - lo8[q], cap);
    }
    sum = wave_sum_i(sum);
    if ((tid & 63) == 0) s_w[tid >> 6] = sum;
    __syncthreads();
    const int start = point_base + s_w[0] + s_w[1];
    const int id = ids[b];
    const bool ok = id >= 0 && id < n_pool;
    const int cnt = ok ? min(pool_off[id + 1] - pool_off[id], cap) : 0;
    if (tid == 0) {
        // clamped like the total: in a batch that exceeds the bound (flagged below) the clouds behind it come out empty or cut short, and no
        // offset points outside the buffers that every later kernel sizes by the bound
        pt_off[b] = (int)min((int64_t)start, bound);
        if (!ok) atomicOr(err, 4);  // sampled index outside the pool: the cloud comes out empty
        if (b == B - 1) {
            int total = start + cnt;
            if (total > bound) {
                atomicOr(err, 4);  // the caller's bound is too small: rows beyond it are dropped
                total = (int)bound;
            }
            pt_off[B] = total;
            *n_dev = total;
        }
    }
    if (!ok) return;
    const int dst0 = (int)min((int64_t)start, bound), n = (int)min((int64_t)start + cnt, bound) - dst0;
    const float* src = pool_xyz + (int64_t)pool_off[id] * 3;
    for (int i = tid; i < n; i += 128) {
        const float x = src[i * 3 + 0], y = src[i * 3 + 1], z = src[i * 3 + 2];
        coords[dst0 + i] = make_float4((float)(cloud_base + b), x / quant, y / quant, z / quant);  // create_input_batch: coords / quantization_size
        float* f = feats + (int64_t)(dst0 + i) * 3;
        f[0] = x; f[1] = y; f[2] = z;
    }
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)ceil_div64(n > 0 ? n : 1, TB)); }

int large_exclusive_scan(const int* in, int64_t n, int* out, int* bsum /*[ceil(n/1024)]*/, hipStream_t st) {
    int64_t nb = ceil_div64(n, 1024);
    if (nb == 0) return 0;
    k_scan_reduce<<<dim3((unsigned)nb), TB, 0, st>>>(in, n, bsum);
    k_scan_single<<<1, 1024, 0, st>>>(bsum, nb, nullptr);
    k_scan_apply<<<dim3((unsigned)nb), TB, 0, st>>>(in, n, bsum, out);
    DD3D_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================

DD3D_API int dd3d_version() { return 101; }

DD3D_API int64_t dd3d_table_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

DD3D_API int64_t dd3d_unique_ws_bytes(int64_t n, int64_t cap) {
    return align_up64(n * 4, 256) + align_up64(cap * 4, 256) + align_up64((ceil_div64(n, TB) + 1) * 4, 256);
}

DD3D_API int dd3d_floor_coords(const float* coords, int64_t N, int32_t* q, void* stream) {
    if (N < 0 || (N > 0 && (!coords || !q))) return DD3D_EARG;
    if (N == 0) return 0;
    k_floor_coords<<<grid1(N), TB, 0, (hipStream_t)stream>>>((const float4*)coords, N, (int4*)q);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_stride_coords(const int32_t* C, const int32_t* m_dev, int64_t bound, int stride, int32_t* q, void* stream) {
    if (bound < 0 || stride <= 0 || (bound > 0 && (!C || !q))) return DD3D_EARG;
    if (bound == 0) return 0;
    k_stride_coords<<<grid1(bound), TB, 0, (hipStream_t)stream>>>((const int4*)C, m_dev, bound, stride, (int4*)q);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// Unique rows of q[bound,4] (first n = *n_dev rows valid) in first-occurrence order.
DD3D_API int dd3d_unique_rows(const int32_t* q, const int32_t* n_dev, int64_t bound, int32_t* table, int64_t cap, int32_t* C, int32_t* inv,
                              int32_t* uidx, int32_t* count_dev, void* ws, int64_t ws_bytes, void* stream) {
    if (bound < 0 || cap < 2 * bound || (cap & (cap - 1)) || !table || !count_dev) return DD3D_EARG;
    if (bound >= (1ll << 30)) return DD3D_ESIZE;
    if (ws_bytes < dd3d_unique_ws_bytes(bound, cap) || (!ws && bound > 0)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    DD3D_HIP(hipMemsetAsync(table, 0xFF, cap * 4, st));
    if (bound == 0) {
        DD3D_HIP(hipMemsetAsync(count_dev, 0, 4, st));
        return 0;
    }
    if (!q || !C || !inv || !uidx) return DD3D_EARG;
    char* w = (char*)ws;
    int* slot_of = (int*)w;
    w += align_up64(bound * 4, 256);
    int* tab_min = (int*)w;
    w += align_up64(cap * 4, 256);
    int* bcnt = (int*)w;
    int64_t nblk = ceil_div64(bound, TB);
    DD3D_HIP(hipMemsetAsync(tab_min, 0x7F, cap * 4, st));  // 0x7F7F7F7F > any row index
    k_hash_insert<<<grid1(bound), TB, 0, st>>>((const int4*)q, n_dev, bound, table, tab_min, (uint32_t)(cap - 1), slot_of);
    k_first_count<<<grid1(bound), TB, 0, st>>>(slot_of, tab_min, n_dev, bound, bcnt);
    k_scan_single<<<1, 1024, 0, st>>>(bcnt, nblk, count_dev);
    k_unique_fill<<<grid1(bound), TB, 0, st>>>((const int4*)q, slot_of, tab_min, table, n_dev, bound, bcnt, (int4*)C, uidx);
    k_inverse<<<grid1(bound), TB, 0, st>>>(slot_of, table, n_dev, bound, inv);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// floor + unique in one call: TensorField.sparse() coordinate half (models/MEConv.py:247)
DD3D_API int dd3d_quantize_unique(const float* coords, int64_t N, int32_t* q, int32_t* table, int64_t cap, int32_t* C, int32_t* inv,
                                  int32_t* uidx, int32_t* count_dev, void* ws, int64_t ws_bytes, void* stream) {
    int rc = dd3d_floor_coords(coords, N, q, stream);
    if (rc) return rc;
    return dd3d_unique_rows(q, nullptr, N, table, cap, C, inv, uidx, count_dev, ws, ws_bytes, stream);
}

DD3D_API int64_t dd3d_csr_ws_bytes(int64_t n) { return align_up64((n + 1) * 4, 256) * 2 + align_up64((ceil_div64(n + 1, 1024) + 1) * 4, 256); }

// vox_off[bound+1] (first *m_dev+1 entries meaningful), vox_pts[bound]: points of each voxel, ascending
DD3D_API int dd3d_voxel_csr(const int32_t* inv, const int32_t* n_dev, int64_t bound, const int32_t* m_dev, int32_t* vox_off, int32_t* vox_pts,
                            void* ws, int64_t ws_bytes, void* stream) {
    if (bound < 0 || !vox_off) return DD3D_EARG;
    if (ws_bytes < dd3d_csr_ws_bytes(bound) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    int* cnt = (int*)w;
    w += align_up64((bound + 1) * 4, 256);
    int* cursor = (int*)w;
    w += align_up64((bound + 1) * 4, 256);
    int* bsum = (int*)w;
    DD3D_HIP(hipMemsetAsync(cnt, 0, (bound + 1) * 4, st));
    DD3D_HIP(hipMemsetAsync(cursor, 0, (bound + 1) * 4, st));
    if (bound == 0) {
        DD3D_HIP(hipMemsetAsync(vox_off, 0, 4, st));
        return 0;
    }
    if (!inv || !vox_pts) return DD3D_EARG;
    k_count_inv<<<grid1(bound), TB, 0, st>>>(inv, n_dev, bound, cnt);
    int rc = large_exclusive_scan(cnt, bound + 1, vox_off, bsum, st);
    if (rc) return rc;
    k_csr_fill<<<grid1(bound), TB, 0, st>>>(inv, n_dev, bound, vox_off, cursor, vox_pts);
    k_csr_sort<<<grid1(bound), TB, 0, st>>>(vox_off, m_dev, bound, vox_pts);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// nbr[K][ld]: row of `keys` (through `table`) at Cq[o] + offset_k, or -1.  ksize 3 -> centred offsets * ts, ksize 2 -> {0,1} * ts.
DD3D_API int dd3d_kernel_map(const int32_t* Cq, const int32_t* mq_dev, int64_t bound_q, const int32_t* table, int64_t cap, const int32_t* keys,
                             int ksize, int tensor_stride, int32_t* nbr, int64_t ld, void* stream) {
    if (bound_q < 0 || ld < bound_q || (ksize != 2 && ksize != 3) || tensor_stride <= 0 || (cap & (cap - 1)) || !table) return DD3D_EARG;
    if (ld == 0) return 0;
    if (!Cq || !keys || !nbr) return DD3D_EARG;
    int K = ksize * ksize * ksize;
    dim3 grid((unsigned)ceil_div64(ld, TB), K);
    k_kernel_map<<<grid, TB, 0, (hipStream_t)stream>>>((const int4*)Cq, mq_dev, bound_q, table, (uint32_t)(cap - 1), (const int4*)keys, ksize,
                                                       tensor_stride, K, nbr, ld);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int64_t dd3d_pairs_ws_bytes(int64_t bound, int K) { return align_up64((ceil_div64(bound > 0 ? bound : 1, TB) * K + 1) * 4, 256); }

// canonical pair lists from a neighbour table: offs[K+1], in_rows/out_rows [<= K*bound], grouped by k, sorted by out_row
DD3D_API int dd3d_kernel_map_pairs(const int32_t* nbr, int64_t ld, const int32_t* m_dev, int64_t bound, int K, int32_t* in_rows, int32_t* out_rows,
                                   int32_t* offs, void* ws, int64_t ws_bytes, void* stream) {
    if (bound < 0 || K <= 0 || ld < bound || !offs) return DD3D_EARG;
    if (ws_bytes < dd3d_pairs_ws_bytes(bound, K) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    if (bound == 0) {
        DD3D_HIP(hipMemsetAsync(offs, 0, (K + 1) * 4, st));
        return 0;
    }
    if (!nbr || !in_rows || !out_rows) return DD3D_EARG;
    int nblk = (int)ceil_div64(bound, TB);
    int* bcnt = (int*)ws;
    dim3 grid(nblk, K);
    k_pairs_count<<<grid, TB, 0, st>>>(nbr, ld, m_dev, bound, bcnt, nblk);
    k_scan_single<<<1, 1024, 0, st>>>(bcnt, (int64_t)nblk * K, offs + K);
    k_pairs_fill<<<grid, TB, 0, st>>>(nbr, ld, m_dev, bound, bcnt, nblk, in_rows, out_rows, offs, K);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// seg_off[B+1]: first row of every cloud (rows grouped by ascending batch index); err |= 1 otherwise
DD3D_API int dd3d_segment_offsets(const int32_t* C, const int32_t* m_dev, int64_t bound, int B, int32_t* seg_off, int32_t* err_dev, void* stream) {
    if (bound < 0 || B < 0 || !seg_off || !err_dev) return DD3D_EARG;
    k_segment_offsets<<<grid1(bound + 1), TB, 0, (hipStream_t)stream>>>((const int4*)C, m_dev, bound, B, seg_off, err_dev);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dense() placement of the last sparse tensor (models/MEConv.py:250-254): per-group minimum corner, site index per voxel,
// and the inverse table vox_at[E0*E1*E2][B] (-1 = no voxel).  cloud_group may be null (one group).
DD3D_API int dd3d_dense_sites(const int32_t* C, const int32_t* m_dev, int64_t bound, const int32_t* cloud_group, int G, int B, int tensor_stride,
                              int E0, int E1, int E2, int32_t* gmin /*[G,3]*/, int32_t* site /*[bound]*/, int32_t* vox_at /*[S,B] or null*/,
                              int32_t* err_dev, void* stream) {
    if (bound < 0 || G <= 0 || B < 0 || tensor_stride <= 0 || E0 <= 0 || E1 <= 0 || E2 <= 0 || !gmin || !err_dev) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    DD3D_HIP(hipMemsetAsync(gmin, 0x7F, (size_t)G * 3 * 4, st));
    if (vox_at && B > 0) DD3D_HIP(hipMemsetAsync(vox_at, 0xFF, (size_t)B * E0 * E1 * E2 * 4, st));
    if (bound == 0) return 0;
    if (!C || !site) return DD3D_EARG;
    k_min_coord<<<grid1(bound), TB, 0, st>>>((const int4*)C, m_dev, bound, cloud_group, gmin);
    k_dense_sites<<<grid1(bound), TB, 0, st>>>((const int4*)C, m_dev, bound, cloud_group, gmin, tensor_stride, E0, E1, E2, B, site, vox_at, err_dev);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// Collated real batch from a resident point pool: cloud b of the batch = pool cloud ids[b], first `cap` points.  The batch may be
// appended to clouds that are already in the buffers: its clouds get batch indices cloud_base + b, its points start at row
// point_base, and pt_off (the array of the WHOLE batch) receives entries [cloud_base, cloud_base + B].  coords[bound,4] /
// feats[bound,3] (first *n_dev rows valid); coordinates are the points divided by `quant` (models/MEConv.py:262), features the raw
// points.  *err_dev |= 4 if an id is outside the pool or the points do not fit into `bound`.
DD3D_API int dd3d_gather_clouds(const float* pool_xyz, const int32_t* pool_off, int n_pool, const int32_t* ids, int B, int cap, int64_t bound,
                                int cloud_base, int64_t point_base, float quant, float* coords, float* feats, int32_t* pt_off, int32_t* n_dev,
                                int32_t* err_dev, void* stream) {
    if (B < 0 || n_pool < 0 || cap < 0 || bound < 0 || bound >= (1ll << 31) || cloud_base < 0 || point_base < 0 || point_base > bound || !(quant > 0.f) || !pt_off ||
        !n_dev || !err_dev)
        return DD3D_EARG;
    if (B > 0 && (!pool_xyz || !pool_off || !ids || !coords || !feats)) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    int32_t* po = pt_off + cloud_base;
    k_cloud_offsets<<<1, 1024, 0, st>>>(pool_off, ids, B, n_pool, cap, bound, (int)point_base, po, n_dev, err_dev);
    if (B > 0) k_gather_clouds<<<dim3(B), 128, 0, st>>>(pool_xyz, pool_off, ids, n_pool, po, B, cloud_base, quant, (float4*)coords, feats);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_syn_head (linear_loss.hip) + dd3d_gather_clouds in ONE launch, the cloud offsets computed by the gathering workgroups
// themselves, plus an optional clear of `zero_words` ints at `zero` (nullable): what a captured step runs first.  The synthetic clouds
// are the first syn_clouds_owned clouds of the batch (cloud_base = their number, point_base = their points).  Same outputs as the two
// calls, bit for bit.
DD3D_API int dd3d_assemble_batch(float* xyz, int64_t n_syn_clouds, int npts, const int32_t* dest_cloud, int do_clamp, float lo, float hi, int zero_first,
                                 float quant, const float* pool_xyz, const int32_t* pool_off, int n_pool, const int32_t* ids, int B, int cap, int64_t bound,
                                 int cloud_base, int64_t point_base, float* coords, float* feats, int32_t* pt_off, int32_t* n_dev, int32_t* err_dev,
                                 int32_t* zero, int zero_words, void* stream) {
    if (B < 0 || n_pool < 0 || cap < 0 || bound < 0 || bound >= (1ll << 31) || cloud_base < 0 || point_base < 0 || point_base > bound || !(quant > 0.f) || !pt_off ||
        !n_dev || !err_dev || n_syn_clouds < 0 || npts <= 0 || zero_words < 0 || (zero_words > 0 && !zero))
        return DD3D_EARG;
    if (B > 0 && (!pool_xyz || !pool_off || !ids || !coords || !feats)) return DD3D_EARG;
    const int64_t n = n_syn_clouds * npts;
    if (n > 0 && (!xyz || !coords || !feats)) return DD3D_EARG;
    if (B == 0)  // no real clouds: the two-call path handles the degenerate batch (offsets of an empty sample, nothing to gather)
        return DD3D_EARG;
    const int64_t syn_blocks = ceil_div64(n, 256);
    if (syn_blocks + B >= (1ll << 31)) return DD3D_ESIZE;
    k_assemble_batch<<<dim3((unsigned)(syn_blocks + B)), 128, 0, (hipStream_t)stream>>>(xyz, n, npts, dest_cloud, do_clamp, lo, hi, zero_first, quant, (int)syn_blocks,
                                                                                      pool_xyz, pool_off, ids, n_pool, B, cap, bound, cloud_base, (int)point_base,
                                                                                      (float4*)coords, feats, pt_off + cloud_base, n_dev, err_dev, zero, zero_words);
    DD3D_CHECK_LAUNCH();
    return 0;
}
