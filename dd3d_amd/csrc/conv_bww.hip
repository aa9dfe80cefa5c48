// Kernel gradient of the Cin = Cout = 128 sparse convolution (MinkowskiConvolution k=3, stride 1; models/MEConv.py:224)
// for batches of SMALL clouds -- the cloud-window form.
//
//   dW[g][k][ci][co] = sum over the rows o of group g with a neighbour at offset k of  X[nbr[k][o]][ci] * dY[o][co]
//
// A neighbour never leaves its cloud (the batch index is part of every voxel key), so a window of WHOLE clouds is closed under the
// kernel map: a workgroup loads the window's rows of X and dY into LDS once and every one of the up to 27 x |window| products reads its
// operands from there.  HBM/L2 sees each operand row once per quarter-workgroup instead of once per offset (the row-chunk kernel in
// conv.hip re-gathers both operands for every offset: 600 MB per cfg2 launch against 2 x 46 MB of operands), and the partial sums stay in
// registers over a whole chunk of ~M/64 rows, so the slab spill shrinks from ~2300 to ~550 128 x 128 slabs.
//
// Decomposition: grid (chunk, quarter, pass).  A chunk is a run of whole clouds of one group (~M/64 rows); a quarter is a (64 input
// channels) x (64 output channels) block of the 128 x 128 gradient, so a window row costs 256 B of X and 256 B of dY in LDS and a
// 256-row window fits beside the pair lists; a pass takes the nine offsets k = pass + 3 slot (flat clouds -- MNIST digits, depth 0 -- have
// all their pairs in pass 1, the other two passes find nothing in their chunk and leave at once).  Twelve waves: wave w owns the 32 x 32
// tile (w & 3) of the quarter for the three offset slots {w >> 2, (w >> 2) + 3, (w >> 2) + 6} (three accumulators, 48 registers).
// Per window and offset slot one wave compacts the (row, neighbour) pairs with ballots into an LDS list (padded to a multiple of 32 with
// a pair of all-zero rows), so the matrix cores only see pairs that exist; v_mfma_f32_32x32x2_f32, exact fp32.
// Deterministic: fixed chunks, windows in cloud order, pairs in row order, slabs reduced in chunk order.
#include "common.h"

namespace {

constexpr int WMAX = 256;         // rows of a window; a single cloud must fit
constexpr int NT = 768;           // 12 waves
constexpr int NSLOT = 9;          // offsets per pass
constexpr int LCAP = WMAX + 32;   // pair-list capacity per slot (padded to a multiple of 32)
constexpr int ZROW = WMAX;        // index of the all-zero row of both windows

// rows per chunk: 64 equal chunks per launch (each taken by four quarter-workgroups: ONE round of 256 workgroups on 256 CUs), at least
// 256 rows.  A function of the TRUE row count (read on the device), so a bound-sized launch chunks exactly like an exact one.  Chunks
// do not stop at group borders: a workgroup that crosses one writes the finished group's partial sums and starts over, so every
// workgroup carries the same number of rows whatever the group sizes.  Partial slabs are indexed by piece = chunk + group (strictly
// increasing along the rows, < 64 + G).
__host__ __device__ __forceinline__ int bwwc_chunk_rows(int M) {
    const int ch = ((M > 0 ? M : 1) + 63) / 64;
    return ch < 256 ? 256 : ch;
}
inline int64_t bwwc_max_chunks(int64_t bound) {
    const int64_t a = ceil_div64(bound > 0 ? bound : 1, 256);
    return a < 64 ? a : 64;
}
// non-empty group holding `row` (row < gro[G])
__device__ __forceinline__ int bwwc_group_of(const int* __restrict__ gro, int G, int row) {
    int g = 0;
    while (g + 1 < G && gro[g + 1] <= row) ++g;
    return g;
}

// first index i in [0, n) with seg[i] >= key (n if none); seg ascending.  64 probes per round (two rounds up to 4096 entries).  Wave-uniform.
__device__ __forceinline__ int lower_bound_wave(const int* __restrict__ seg, int n, int key) {
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = n;
    for (;;) {
        const int len = hi - lo;
        if (len <= 0) return lo;
        const int step = (len + 63) / 64;
        const int idx = lo + lane * step;
        const bool below = idx < hi && seg[idx] < key;
        const int c = __popcll(__ballot(below));
        if (step == 1) return lo + c;
        if (c == 0) return lo;
        lo = lo + (c - 1) * step + 1;
        hi = min(lo + step - 1, hi);
    }
}

// mask |= bit k for every offset k with a pair anywhere in the table (the launch's ACTIVE offsets: nine for clouds that are flat along an
// axis).  grid (64, K): workgroup (x, k) scans its 64th of row k; one atomic per workgroup that found a pair.
__global__ __launch_bounds__(256) void k_bwwc_active(const int* __restrict__ nbr, int64_t ld, const int* __restrict__ gro, int G, unsigned* __restrict__ mask) {
    __shared__ int s_any;
    const int M = gro[G], k = blockIdx.y;
    const int per = (M + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(r0 + per, M);
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    const int* nk = nbr + (int64_t)k * ld;
    bool any = false;
    for (int i = r0 + threadIdx.x; i < r1; i += 256) any |= nk[i] >= 0;
    if (__ballot(any) && (threadIdx.x & 63) == 0) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s_any) atomicOr(mask, 1u << k);
}

__global__ __launch_bounds__(NT) void k_conv128_bww_clouds(const float* __restrict__ X, const float* __restrict__ dY, const int* __restrict__ nbr, int64_t ld,
                                                           int K, const int* __restrict__ seg, int B, const int* __restrict__ gro, int G,
                                                           const unsigned* __restrict__ active, float* __restrict__ slabs,
                                                           int* __restrict__ slab_valid, int* __restrict__ err) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    f32x4* Xs4 = reinterpret_cast<f32x4*>(smem_raw);                 // [(WMAX + 1)][16]: 64 input channels of every window row
    f32x4* Ds4 = Xs4 + (WMAX + 1) * 16;                              // [(WMAX + 1)][16]: 64 output channels
    int* s_list = reinterpret_cast<int*>(Ds4 + (WMAX + 1) * 16);     // [NSLOT][LCAP]: (row << 16) | neighbour, both window-local
    int* s_cnt = s_list + NSLOT * LCAP;                              // [NSLOT] padded pair count of the current window
    const float* Xs = reinterpret_cast<const float*>(Xs4);
    const float* Ds = reinterpret_cast<const float*>(Ds4);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int c = blockIdx.x, cih = blockIdx.y >> 1, coh = blockIdx.y & 1, pass = blockIdx.z;
    // this pass's offsets: the 9 pass-th .. active offsets in ascending order (slot s -> offset koff(s)); passes beyond the active set leave
    const unsigned amask = *active;
    const int nact = __popc(amask), nslot = min(NSLOT, nact - NSLOT * pass);
    if (nslot <= 0) return;
    auto koff = [&](int slot) {  // offset of the (NSLOT * pass + slot)-th set bit
        unsigned m = amask;
        for (int i = 0; i < NSLOT * pass + slot; ++i) m &= m - 1;
        return __builtin_ctz(m);
    };
    const int M = gro[G], CH = bwwc_chunk_rows(M);
    const int r0 = c * CH, r1 = min(r0 + CH, M);
    if (r0 >= M) return;
    // the chunk's clouds: those whose first row lies in [r0, r1); its groups: those of rows r0 .. r1 - 1
    const int b0 = lower_bound_wave(seg, B + 1, r0), b1 = lower_bound_wave(seg, B + 1, r1);
    const int g_lo = bwwc_group_of(gro, G, r0), g_hi = bwwc_group_of(gro, G, r1 - 1);

    for (int i = tid; i < 16; i += NT) {
        Xs4[ZROW * 16 + i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Ds4[ZROW * 16 + i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int myk = wid < nslot ? koff(wid) : 0;   // the offset whose pair list this wave builds
    const int tp = wid & 3, cit = tp >> 1, cot = tp & 1, sb = wid >> 2;
    const int kj[3] = {sb < nslot ? koff(sb) : -1, sb + 3 < nslot ? koff(sb + 3) : -1, sb + 6 < nslot ? koff(sb + 6) : -1};
    const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
    const f32x4* D4 = reinterpret_cast<const f32x4*>(dY);
    const float* xcol = Xs + cit * 32 + i32;   // this lane's operand columns of the two windows
    const float* dcol = Ds + cot * 32 + i32;
    int cur = b0;
    for (int g = g_lo; g <= g_hi; ++g) {
        const int gend = gro[g + 1];
        f32x16 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        bool worked = false;
        while (cur < b1) {
            // the window: as many whole clouds of this group from `cur` on as fit into WMAX rows (one vector load of the next 64 cloud borders)
            const int w0 = seg[cur];
            if (w0 >= gend) break;  // the next cloud belongs to the next group
            const int e = cur + 1 + lane;
            const int v = e <= b1 ? seg[e] : 0x7FFFFFFF;
            const int cnt = __popcll(__ballot(e <= b1 && v - w0 <= WMAX && v <= gend));
            if (cnt == 0) {  // a single cloud with more rows than the window holds: flagged, skipped (the host routes such batches to the row-chunk kernel)
                if (tid == 0) atomicOr(err, 16);
                ++cur;
                continue;
            }
            const int w1 = __shfl(v, cnt - 1, 64);
            const int rows = w1 - w0;
            cur += cnt;
            if (rows <= 0) continue;
            worked = true;
            // table entries of this wave's offset first (their latency overlaps the window loads), then the window rows: 16 float4 of X
            // and 16 of dY per row, all loads of a thread in flight before its stores
            int nb[WMAX / 64];
            if (wid < nslot) {
                const int* nk = nbr + (int64_t)myk * ld + w0;
#pragma unroll
                for (int j = 0; j < WMAX / 64; ++j) {
                    const int r = j * 64 + lane;
                    nb[j] = r < rows ? nk[r] : -1;
                }
            }
            constexpr int IT = (WMAX * 16 + NT - 1) / NT;
            f32x4 rx[IT], rd[IT];
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int i = min(j * NT + tid, rows * 16 - 1);
                const int64_t grow = (int64_t)(w0 + (i >> 4)) * 32;
                rx[j] = X4[grow + cih * 16 + (i & 15)];
                rd[j] = D4[grow + coh * 16 + (i & 15)];
            }
            __syncthreads();  // the previous window's operand reads are done
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int i = j * NT + tid;
                if (i < rows * 16) {
                    Xs4[i] = rx[j];
                    Ds4[i] = rd[j];
                }
            }
            if (wid < NSLOT) {  // one wave per offset slot: stable ballot compaction of the window's pairs
                const int s = wid;
                int total = 0;
                if (s < nslot) {
#pragma unroll
                    for (int j = 0; j < WMAX / 64; ++j) {
                        const int r = j * 64 + lane;
                        const int loc = nb[j] - w0;
                        const bool ok = nb[j] >= 0 && loc >= 0 && loc < rows;  // a neighbour outside the window would mean the rows are not grouped by cloud
                        if (nb[j] >= 0 && !ok) atomicOr(err, 1);
                        const unsigned long long bal = __ballot(ok);
                        if (ok) s_list[s * LCAP + total + __popcll(bal & ((1ull << lane) - 1ull))] = (r << 16) | loc;
                        total += __popcll(bal);
                    }
                    const int padded = (total + 31) & ~31;
                    if (total + lane < padded) s_list[s * LCAP + total + lane] = (ZROW << 16) | ZROW;
                    total = padded;
                }
                if (lane == 0) s_cnt[s] = total;
            }
            __syncthreads();
            // The wave's blocks of 32 pairs, over its three offset slots, as ONE software pipeline: while the 16 MFMAs of a block issue, the
            // list entries and the 32 operands of the next block are read from LDS in the gaps between them (an MFMA holds the pipe for 64
            // cycles; LDS reads issued behind the whole group would only start when the last one has issued).
            const int nj[3] = {s_cnt[sb], s_cnt[sb + 3], s_cnt[sb + 6]};
            int fj = 0, fblk = 0;  // the next block to fetch
            auto next_block = [&]() -> int {
                while (fj < 3 && fblk >= nj[fj]) {
                    ++fj;
                    fblk = 0;
                }
                if (fj >= 3) return -1;
                const int off = (sb + 3 * fj) * LCAP + fblk + 16 * h;
                fblk += 32;
                return off;
            };
            float A[16], Bv[16];
            auto read_ev = [&](int off, int (&ev)[16]) {
                if (off < 0) {  // no further block: the reads below go to the all-zero rows (no second code path)
#pragma unroll
                    for (int t = 0; t < 16; ++t) ev[t] = (ZROW << 16) | ZROW;
                    return;
                }
                const int4* lp = reinterpret_cast<const int4*>(s_list + off);
                const int4 e0 = lp[0], e1 = lp[1], e2 = lp[2], e3 = lp[3];
                ev[0] = e0.x; ev[1] = e0.y; ev[2] = e0.z; ev[3] = e0.w; ev[4] = e1.x; ev[5] = e1.y; ev[6] = e1.z; ev[7] = e1.w;
                ev[8] = e2.x; ev[9] = e2.y; ev[10] = e2.z; ev[11] = e2.w; ev[12] = e3.x; ev[13] = e3.y; ev[14] = e3.z; ev[15] = e3.w;
            };
            {
                int ev[16];
                read_ev(next_block(), ev);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    A[t] = xcol[(ev[t] & 0xFFFF) * 64];
                    Bv[t] = dcol[(ev[t] >> 16) * 64];
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                for (int blk = 0; blk < nj[j]; blk += 32) {
                    int evn[16];
                    read_ev(next_block(), evn);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t], Bv[t], acc[j], 0, 0, 0);
                        if (t >= 3) {  // the operand registers of element t - 3 are free: refill them for the next block in this MFMA's shadow
                            A[t - 3] = xcol[(evn[t - 3] & 0xFFFF) * 64];
                            Bv[t - 3] = dcol[(evn[t - 3] >> 16) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int t = 13; t < 16; ++t) {
                        A[t] = xcol[(evn[t] & 0xFFFF) * 64];
                        Bv[t] = dcol[(evn[t] >> 16) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // this group's piece: the workgroup's 64 x 64 block of the [128][128] slab of every offset of this pass
        const int piece = c + g;
        if (blockIdx.y == 0 && tid < nslot) slab_valid[(int64_t)piece * K + koff(tid)] = worked ? 1 : 0;
        if (worked) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (kj[j] < 0) continue;
                float* slab = slabs + ((int64_t)piece * K + kj[j]) * (128 * 128);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = cih * 64 + cit * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    slab[ci * 128 + coh * 64 + cot * 32 + i32] = acc[j][r];
                }
            }
        }
    }
}

// dW[g][k] = sum over the pieces of group g (the chunks its rows touch), in chunk order, of the slabs that exist (zeros when none does)
__global__ __launch_bounds__(256) void k_bwwc_reduce(const f32x4* __restrict__ slabs, const int* __restrict__ slab_valid, const int* __restrict__ gro, int G,
                                                     int K, const unsigned* __restrict__ active, f32x4* __restrict__ dW) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // float4 index inside a [128][128] slab (4096 of them)
    const int k = blockIdx.y, g = blockIdx.z;
    if (!((*active >> k) & 1u)) {  // no pair at this offset anywhere: no flags were written
        dW[((int64_t)g * K + k) * 4096 + e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int CH = bwwc_chunk_rows(gro[G]);
    const int a = gro[g], b = gro[g + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (b > a)
        for (int c = a / CH; c <= (b - 1) / CH; ++c)
            if (slab_valid[(int64_t)(c + g) * K + k]) s += slabs[((int64_t)(c + g) * K + k) * 4096 + e];
    dW[((int64_t)g * K + k) * 4096 + e] = s;
}

constexpr size_t BWWC_SMEM = 2 * (size_t)(WMAX + 1) * 16 * sizeof(f32x4) + (size_t)NSLOT * LCAP * 4 + NSLOT * 4;

}  // namespace

DD3D_API int dd3d_conv128_bwd_weight_clouds_max_rows(void) { return WMAX; }

DD3D_API int64_t dd3d_conv128_bwd_weight_clouds_ws_bytes(int64_t M_out, int G) {
    const int64_t slots = bwwc_max_chunks(M_out) + (G > 0 ? G : 1);  // pieces = chunk + group
    return align_up64(slots * 27 * 128 * 128 * 4, 256) + align_up64(slots * 27 * 4, 256) + 256;
}

DD3D_API int dd3d_conv128_bwd_weight_clouds(const float* X, const float* dY, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* seg_off,
                                            int B, const int32_t* group_row_off, int G, float* dW, int32_t* err_dev, void* ws, int64_t ws_bytes,
                                            void* stream) {
    if (M_out < 0 || K != 27 || ld < M_out || B < 0 || G <= 0 || !group_row_off || !seg_off || !dW || !err_dev) return DD3D_EARG;
    if (M_out >= (1ll << 31) / 128) return DD3D_ESIZE;
    if (ws_bytes < dd3d_conv128_bwd_weight_clouds_ws_bytes(M_out, G) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    const int64_t chunks = bwwc_max_chunks(M_out), slots = chunks + G;
    float* slabs = (float*)ws;
    int* valid = (int*)((char*)ws + align_up64(slots * 27 * 128 * 128 * 4, 256));
    unsigned* active = (unsigned*)((char*)valid + align_up64(slots * 27 * 4, 256));
    DD3D_HIP(hipMemsetAsync(active, 0, 4, st));
    if (M_out > 0 && B > 0) {
        if (!X || !dY || !nbr) return DD3D_EARG;
        // > 64 KB of dynamic LDS has to be requested (per device; a host-side call, legal during stream capture)
        DD3D_HIP(hipFuncSetAttribute((const void*)k_conv128_bww_clouds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWWC_SMEM));
        k_bwwc_active<<<dim3(64, K), 256, 0, st>>>(nbr, ld, group_row_off, G, active);
        k_conv128_bww_clouds<<<dim3((unsigned)chunks, 4, 3), NT, BWWC_SMEM, st>>>(X, dY, nbr, ld, K, seg_off, B, group_row_off, G, active, slabs, valid, err_dev);
    }
    k_bwwc_reduce<<<dim3(16, K, G), 256, 0, st>>>((const f32x4*)slabs, valid, group_row_off, G, K, active, (f32x4*)dW);
    DD3D_CHECK_LAUNCH();
    return 0;
}
