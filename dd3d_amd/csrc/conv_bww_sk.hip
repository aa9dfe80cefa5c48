// Kernel gradient of the Cin = Cout = 128 sparse convolution (MinkowskiConvolution backward, dW; models/MEConv.py:224) as a
// STREAM-K launch: dW[g][k] = sum over the rows o of group g of X[nbr[k][o]]^T . dY[o].
//
// Unit of work = one 32-pair MFMA step of one (group, offset) tile [128 x 128].  A planning launch counts the pairs of every
// (512-row window, offset) and therefore the steps of every tile; the persistent workgroups then split the list of ALL steps -- in
// the order (group, offset, window, step) -- into equal contiguous ranges, so every workgroup runs the same number of steps (no
// tile-granularity tail: the row-chunk form ended on its most loaded CU) and keeps the [128 x 128] accumulator of a tile in registers
// across as many windows as its range covers: a tile that lies inside one range is written straight into dW, a tile that is cut by
// a range boundary leaves ONE partial per workgroup that touches it (at most two partials per workgroup, head and tail) and a
// reduce launch adds them in range order.  Everything that fixes the order of the additions -- window size, step counts, range
// boundaries -- is a function of the TRUE row counts (group_row_off, on the device) and of the neighbour table, never of the launch
// bound: a bound-sized launch (HIP-graph replay) gives the bits of an exact one, and a launch gives the same bits every time.
//
// Logical range p of workgroup b is XCD-major (p = (b mod 8) * nwg/8 + b / 8): the workgroups of one XCD then work on neighbouring
// ranges, i.e. at any moment on the SAME rows of a group for different offsets, and the gathered X rows and dY rows are fetched
// into that XCD's L2 once instead of once per offset.
#include "common.h"

namespace {

constexpr int WIN = DD3D_BWW_WIN;              // rows per window (windows never straddle a group)
constexpr int MAX_TILES = DD3D_BWW_MAX_TILES;  // G * 27 tiles whose step prefix fits the workgroup's LDS table

__device__ __forceinline__ unsigned wave_or_u(unsigned m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    return m;
}

// ---- planning: steps of every (window, offset), steps of every tile, mask of the offsets present anywhere ----
// steps_t[k][w] (k-major: the windows of a tile are consecutive), tile_steps[g * 27 + k] and *active are zero before the launch.
__global__ __launch_bounds__(256) void k_bww_plan(const int* __restrict__ nbr, int64_t ld, const int* __restrict__ group_row_off, int G, int nwinB,
                                                  int* __restrict__ steps_t, int* __restrict__ tile_steps, unsigned* __restrict__ active) {
    __shared__ int s_cnt[27];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int g, r0, r1;
    if (!resolve_chunk(w, group_row_off, G, WIN, &g, &r0, &r1)) {  // a slot beyond the windows of this batch
        if (tid < 27) steps_t[(int64_t)tid * nwinB + w] = 0;
        return;
    }
    if (tid < 27) s_cnt[tid] = 0;
    __syncthreads();
    static_assert(WIN == 512, "two rows per thread");
    const int ra = r0 + tid, rb = r0 + 256 + tid;
    int va[27], vb[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {  // clamped addresses, all 54 loads in flight
        va[k] = nbr[(int64_t)k * ld + min(ra, r1 - 1)];
        vb[k] = nbr[(int64_t)k * ld + min(rb, r1 - 1)];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int c = __popcll(__ballot(ra < r1 && va[k] >= 0)) + __popcll(__ballot(rb < r1 && vb[k] >= 0));
        if (lane == 0 && c) atomicAdd(&s_cnt[k], c);
    }
    __syncthreads();
    int st = 0;
    if (tid < 27) {
        st = (s_cnt[tid] + 31) / 32;
        steps_t[(int64_t)tid * nwinB + w] = st;
        if (st) atomicAdd(&tile_steps[g * 27 + tid], st);  // integer sums: the order of the atomics does not matter
    }
    if (tid < 64) {
        const unsigned m = (unsigned)__ballot(st > 0);
        // bits are only ever added: a window that brings nothing new skips the atomic; a stale read costs a needless atomic at worst
        if (tid == 0 && (m & ~__atomic_load_n(active, __ATOMIC_RELAXED))) atomicOr(active, m);
    }
}

// S(p) = first step of logical range p of nwg
__device__ __forceinline__ int range_start(int T, int p, int nwg) { return (int)((int64_t)T * p / nwg); }

__global__ __launch_bounds__(256, 3) void k_conv128_bww_sk(const float* __restrict__ X, const float* __restrict__ dY, const int* __restrict__ nbr, int64_t ld,
                                                           const int* __restrict__ group_row_off, int G, const int* __restrict__ steps_t, int nwinB,
                                                           const int* __restrict__ tile_steps, int* __restrict__ meta, float* __restrict__ part,
                                                           float* __restrict__ dW, int nwg) {
    // The contraction of a step has the shape of k_conv128_rows (conv.hip): wave w owns input channels [32w, 32w+32) and all 128 output
    // channels (four 32x32 accumulators); its A operand -- X^T of the step's 32 gathered rows, one float per lane and row pair -- goes
    // global -> registers; the 32 dY rows of the step (16 KB) are shared by the four waves through a double-buffered LDS chunk: ONE
    // barrier per step, the loads of step s+1 in flight under the 64 MFMAs of step s and issued between them.  The pairs of a window
    // are compacted first (stable ballot scan), so the matrix cores only see rows that have a neighbour at the tile's offset.
    __shared__ f32x4 Bs4[2][32 * 32];
    __shared__ __align__(16) int s_row[WIN + 32], s_nb[WIN + 32];
    __shared__ int s_wc[8], s_scan[5], s_hit[2];
    __shared__ int s_ts[MAX_TILES + 1];  // first step of every tile (exclusive prefix of tile_steps); s_ts[ntile] = all steps
    __shared__ int s_wf[DD3D_BWW_MAX_GROUPS + 1];  // first window of every group
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int ntile = G * 27;
    const int p = (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);

    // ---- tile prefix (every workgroup for itself: ~2 us, no dependent planning node) ----
    {
        int loc[MAX_TILES / 256], sum = 0;
#pragma unroll
        for (int j = 0; j < MAX_TILES / 256; ++j) {
            const int idx = tid * (MAX_TILES / 256) + j;
            loc[j] = sum;
            sum += idx < ntile ? tile_steps[idx] : 0;
        }
        int total;
        const int base = block_excl_scan_256(sum, &total, s_scan);
#pragma unroll
        for (int j = 0; j < MAX_TILES / 256; ++j) {
            const int idx = tid * (MAX_TILES / 256) + j;
            if (idx < ntile) s_ts[idx] = base + loc[j];
        }
        if (tid == 0) {
            s_ts[ntile] = total;
            int first = 0;
            for (int g = 0; g < G; ++g) {
                s_wf[g] = first;
                first += (group_row_off[g + 1] - group_row_off[g] + WIN - 1) / WIN;
            }
            s_wf[G] = first;
        }
    }
    __syncthreads();
    const int T = s_ts[ntile];
    if (blockIdx.x == 0)  // for the reduce launch
        for (int i = tid; i <= ntile; i += 256) meta[i] = s_ts[i];
    const int s0 = range_start(T, p, nwg), s1 = range_start(T, p + 1, nwg);
    if (s0 >= s1) return;  // (uniform) fewer steps than workgroups
    // the tile that holds step s0: number of tiles that END at or before s0
    int t;
    {
        int c = 0;
        for (int i = tid; i < ntile; i += 256) c += s_ts[i + 1] <= s0;
        c = wave_sum_i(c);
        if (lane == 0) s_wc[wid] = c;
        __syncthreads();
        t = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
        __syncthreads();
    }

    const float* Xc = X + wid * 32 + i32;  // this lane's input channel
    const f32x4* D4 = (const f32x4*)dY;
    for (int cur = s0; cur < s1;) {
        while (s_ts[t + 1] <= cur) ++t;  // tiles without a step (an offset the group does not have)
        const int ts = s_ts[t], te = s_ts[t + 1], pe = min(te, s1);  // this workgroup's piece of the tile: steps [cur, pe)
        const int g = t / 27, k = t - g * 27;
        const int* nk = nbr + (int64_t)k * ld;
        const int* stw = steps_t + (int64_t)k * nwinB + s_wf[g];  // steps of the tile's windows
        const int nwin = s_wf[g + 1] - s_wf[g], gr0 = group_row_off[g], gr1 = group_row_off[g + 1];
        // the window that holds step `rel` of the tile, and the steps of the tile before that window
        const int rel = cur - ts;
        int wj = 0, wstart = 0;
        if (rel > 0) {
            for (int basew = 0; basew < nwin; basew += 256) {
                const int j = basew + tid, v = j < nwin ? stw[j] : 0;
                int tot;
                const int ex = block_excl_scan_256(v, &tot, s_scan);
                if (v > 0 && wstart + ex <= rel && rel < wstart + ex + v) s_hit[0] = j, s_hit[1] = wstart + ex;  // exactly one thread, in one pass
                __syncthreads();
                if (rel < wstart + tot) {
                    wj = s_hit[0], wstart = s_hit[1];
                    break;
                }
                wstart += tot;
            }
        }
        f32x16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

        int skip = rel - wstart;  // steps of the first window that belong to the workgroup before this one
        for (int done = cur; done < pe && wj < nwin; ++wj) {  // (wj < nwin: an exit every wave reaches even if the table changed under the plan)
            const int wb = gr0 + wj * WIN, r1 = min(wb + WIN, gr1);
            // ---- compaction of the window: all table reads first, one pass of ballots ----
            int nbv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wb + j * 256 + tid;
                nbv[j] = row < r1 ? nk[row] : -1;
            }
            __syncthreads();  // the previous window's lists are no longer read
            unsigned long long bal[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bal[j] = __ballot(nbv[j] >= 0);
                if (lane == 0) s_wc[j * 4 + wid] = __popcll(bal[j]);
            }
            __syncthreads();
            int total = 0, base[2];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int x = s_wc[q];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (q == j * 4 + wid) base[j] = total;
                total += x;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (nbv[j] >= 0) {
                    const int pos = base[j] + __popcll(bal[j] & ((1ull << lane) - 1ull));
                    s_row[pos] = wb + j * 256 + tid;
                    s_nb[pos] = nbv[j];
                }
            if (total == 0) continue;  // (uniform)
            __syncthreads();
            if (tid < 32) {  // the tail of the last step reads the last pair again; its A operand is zeroed at use
                s_row[total + tid] = s_row[total - 1];
                s_nb[total + tid] = s_nb[total - 1];
            }
            __syncthreads();
            const int steps = (total + 31) / 32, last = total - 1;
            const int sa = skip, sb_end = min(steps, sa + (pe - done));  // steps [sa, sb_end) of this window are this piece's
            skip = 0;
            done += sb_end - sa;
            float curx[16], nxt[16];
            f32x4 rb[4];
            {
#pragma unroll
                for (int q = 0; q < 16; ++q) curx[q] = Xc[(int64_t)s_nb[min(sa * 32 + h * 16 + q, last)] * 128];
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs4[0][j * 256 + tid] = D4[(int64_t)s_row[min(sa * 32 + j * 8 + (tid >> 5), last)] * 32 + (tid & 31)];
            }
            __syncthreads();
            int buf = 0;
            for (int sb = sa; sb < sb_end; ++sb) {
                // the next step (the last one "prefetches" itself again: one code path, see k_conv128_rows)
                const int nbase = (sb + 1 < sb_end ? sb + 1 : sb) * 32;
                const int live = total - sb * 32 - h * 16;  // pairs of this lane's half that exist in this step
                const f32x4* Bp = &Bs4[buf][(h * 16) * 32 + i32];
                const int4* ia = (const int4*)(s_nb + nbase + h * 16);  // (the lists are padded by 32 entries: no clamp on the address)
                const int* ib = s_row + nbase + (tid >> 5);
                f32x4 b0 = Bp[0], b1 = Bp[32];
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    f32x4 n0 = b0, n1 = b1;
                    if (q + 2 < 16) n0 = Bp[(q + 2) * 32], n1 = Bp[(q + 3) * 32];
                    if (q < 8) {  // all loads of the next step go out in the first half of this one
                        const int4 ix = ia[q / 2];
                        nxt[2 * q] = Xc[(int64_t)ix.x * 128];
                        nxt[2 * q + 1] = Xc[(int64_t)ix.y * 128];
                        nxt[2 * q + 2] = Xc[(int64_t)ix.z * 128];
                        nxt[2 * q + 3] = Xc[(int64_t)ix.w * 128];
                        rb[q / 2] = D4[(int64_t)ib[q / 2 * 8] * 32 + (tid & 31)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float a0 = q < live ? curx[q] : 0.f, a1 = q + 1 < live ? curx[q + 1] : 0.f;
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[n], acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[n], acc[n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    b0 = n0, b1 = n1;
                }
                buf ^= 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs4[buf][j * 256 + tid] = rb[j];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; ++q) curx[q] = nxt[q];
            }
        }
        // ---- the piece is done: the whole tile -> dW; cut at its front -> head partial; cut at its back only -> tail partial ----
        // C/D layout of a 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); block n, lane col i32 <-> output channel 4*i32 + n
        f32x4* out4 = (cur == ts && pe == te) ? (f32x4*)dW + (int64_t)t * (128 * 32)
                                              : (f32x4*)part + ((int64_t)p * 2 + (cur == ts ? 1 : 0)) * (128 * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            out4[ci * 32 + i32] = v;
        }
        cur = pe;
    }
}

// dW[t] for every tile that no single range covered: zeros for a tile without steps, else its partials in range order (the tail
// partial of the range the tile starts in, then the head partials of the ranges it continues through).
__global__ __launch_bounds__(256) void k_bww_reduce(const f32x4* __restrict__ part, const int* __restrict__ meta, int G, int nwg, f32x4* __restrict__ dW) {
    const int e = blockIdx.x * 256 + threadIdx.x, t = blockIdx.z * 27 + blockIdx.y, ntile = G * 27;
    const int ts = meta[t], te = meta[t + 1], T = meta[ntile];
    f32x4* out = dW + (int64_t)t * (128 * 32) + e;
    if (ts == te) {
        *out = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    // p_lo / p_hi: the ranges that hold the tile's first / last step (largest p with S(p) <= step)
    int p_lo = (int)((int64_t)ts * nwg / T), p_hi = (int)((int64_t)(te - 1) * nwg / T);
    while (p_lo + 1 < nwg && range_start(T, p_lo + 1, nwg) <= ts) ++p_lo;
    while (p_hi + 1 < nwg && range_start(T, p_hi + 1, nwg) <= te - 1) ++p_hi;
    if (p_lo == p_hi) return;  // the tile went straight into dW
    f32x4 s = part[((int64_t)p_lo * 2 + 1) * (128 * 32) + e];
    for (int p = p_lo + 1; p <= p_hi; p += 8) {  // eight partials in flight, added in range order
        f32x4 x[8];
        bool ok[8];  // (uniform) fewer steps than ranges: a range between the two may be empty, its workgroup wrote nothing
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pq = min(p + q, p_hi);
            ok[q] = p + q <= p_hi && range_start(T, pq, nwg) < range_start(T, pq + 1, nwg);
            x[q] = part[((int64_t)pq * 2) * (128 * 32) + e];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (ok[q]) s += x[q];
    }
    *out = s;
}

inline int64_t nwin_bound(int64_t M_bound, int G) { return ceil_div64(M_bound > 0 ? M_bound : 1, WIN) + G; }

}  // namespace

int64_t conv128_bww_sk_ws_bytes(int64_t M_bound, int G) {
    return 256 + align_up64((int64_t)G * 27 * 4, 256) + align_up64(((int64_t)G * 27 + 1) * 4, 256) + align_up64(27 * nwin_bound(M_bound, G) * 4, 256) +
           (int64_t)DD3D_BWW_MAX_NWG * 2 * 128 * 128 * 4;
}

int conv128_bww_sk(const float* X, const float* dY, const int32_t* nbr, int64_t ld, int64_t M_bound, const int32_t* group_row_off, int G, float* dW, void* ws,
                   int nwg, hipStream_t st) {
    char* b = (char*)ws;
    unsigned* active = (unsigned*)b;
    int* tile_steps = (int*)(b + 256);
    int* meta = (int*)((char*)tile_steps + align_up64((int64_t)G * 27 * 4, 256));
    int* steps_t = (int*)((char*)meta + align_up64(((int64_t)G * 27 + 1) * 4, 256));
    const int64_t nwinB = nwin_bound(M_bound, G);
    float* part = (float*)((char*)steps_t + align_up64(27 * nwinB * 4, 256));
    DD3D_HIP(hipMemsetAsync(b, 0, 256 + (size_t)G * 27 * 4, st));  // the offsets mask and the tiles' step counts
    k_bww_plan<<<dim3((unsigned)nwinB), 256, 0, st>>>(nbr, ld, group_row_off, G, (int)nwinB, steps_t, tile_steps, active);
    k_conv128_bww_sk<<<dim3((unsigned)nwg), 256, 0, st>>>(X, dY, nbr, ld, group_row_off, G, steps_t, (int)nwinB, tile_steps, meta, part, dW, nwg);
    k_bww_reduce<<<dim3(16, 27, (unsigned)G), 256, 0, st>>>((const f32x4*)part, meta, G, nwg, (f32x4*)dW);
    DD3D_CHECK_LAUNCH();
    return 0;
}
