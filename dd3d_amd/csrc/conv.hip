// Sparse convolution (MinkowskiConvolution k=3, stride 1; models/MEConv.py:224) for gfx950.
//
// Output-stationary gather-GEMM: a workgroup owns a tile of OUTPUT rows and walks the kernel offsets k, gathering the
// neighbour rows X[nbr[k][o]] into LDS and contracting the channel dimension with W[k] on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, bit-identical to an fmaf chain).  Every output row is written exactly once by
// exactly one wave, so the "scatter" needs no atomics; offsets (and 64-row wave tiles) without any neighbour are skipped.
//   forward   : Y[o]  = sum_k X[nbr[k][o]]      . W[k]
//   bwd-data  : dX[i] = sum_k dY[nbr[K-1-k][i]] . W[k]^T     (stride-1 odd kernels are mirror-symmetric) -> same kernel, mirror=1, W^T
//   bwd-weight: dW[g][k] = sum_{o in group g} X[nbr[k][o]]^T . dY[o]   -> row-chunk partial slabs + fixed-order reduce (deterministic)
// Cin == 3 (first layer) is a bandwidth kernel on the VALU (a 3-wide contraction has nothing for MFMA to do).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// Row-chunk sizes of the backward-weight kernels.  They are functions of the TRUE number of rows (group_row_off[G], read on the
// device), never of the launch bound, so a launch over a bound (HIP-graph replay) sums in exactly the order of an exact launch.
//   Cin = 3  : cheap chunks, aim for ~512 of them (two per CU), at least 64 rows each
//   Cin = 128: about 256 row chunks per launch, at least 512 rows each (measured: fewer, larger chunks are slower even when they
//              fit the resident slots exactly -- the per-chunk compaction phases overlap better across many workgroups)
#ifndef DD3D_CHUNK3_TARGET
#define DD3D_CHUNK3_TARGET 512
#endif
__host__ __device__ __forceinline__ int chunk3(int M) {
    int ch = (((M > 0 ? M : 1) + DD3D_CHUNK3_TARGET - 1) / DD3D_CHUNK3_TARGET + 31) / 32 * 32;
    return ch < 64 ? 64 : ch;
}
__host__ __device__ __forceinline__ int chunk128(int M) {
    int ch = (((M > 0 ? M : 1) + 255) / 256 + 31) / 32 * 32;
#ifndef DD3D_CHUNK128_MIN
#define DD3D_CHUNK128_MIN 512
#endif
    const int lo = M >= 65536 ? DD3D_CHUNK128_MIN : 128;  // small launches (a synthetic batch, one class per GPU) need many short chunks to fill the CUs
    return ch < lo ? lo : ch;
}
// slots needed for ANY row count <= bound: the chunk size is smallest (64 / 512) when there are few rows
inline int64_t max_slots3(int64_t bound, int G) { return ceil_div64(bound, 64) < DD3D_CHUNK3_TARGET + G ? ceil_div64(bound, 64) + G : DD3D_CHUNK3_TARGET + G; }
inline int64_t max_slots128(int64_t bound, int G) { return ceil_div64(bound, 128) < 512 + G ? ceil_div64(bound, 128) + G : 512 + G; }

// ---------------------------------------------------------------------------------------------
// Cin = Cout = 128 forward / backward-data
// ---------------------------------------------------------------------------------------------
// Offsets-split launches (SPLIT): a launch with few tiles (the synthetic rows of the second-order pass: ~100 tiles for 256 CUs, each
// a serial chain of 108 steps) is dealt over blockIdx.y = S workgroups per tile, each taking a contiguous range of the K offsets and
// writing its partial tile to part[s][M_bound][128]; k_split_sum adds the partials in split order.  S follows the TRUE tile count.
__host__ __device__ __forceinline__ int split_count(int ntile, int smax) {
    // <= 56 tiles: 9 splits of 3 offsets; from ~170 tiles on the tiles alone keep the matrix cores of every CU busy (measured: 250 tiles
    // take 57 us split nine ways plus 8 us of summing against 60 us unsplit) and the launch runs unsplit, straight into Y
    const int s = 512 / (ntile > 0 ? ntile : 1);
    return s < 3 ? 1 : (s > smax ? smax : s);
}

__device__ __forceinline__ unsigned wave_or_u(unsigned m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    return m;
}

template <int TM, int WAVES_M, int WAVES_N, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_conv128(const float* __restrict__ X, const float* __restrict__ W, const int* __restrict__ nbr, int64_t ld,
                                                 int K, int M_bound, const int* __restrict__ m_dev, int mirror, float* __restrict__ Y, int xcd_remap,
                                                 const int* __restrict__ perm = nullptr, const int* __restrict__ tile_map = nullptr,
                                                 float* __restrict__ part = nullptr, int smax = 1) {
    constexpr int BK = 32, LDA4 = 9 /* (BK+4)/4 float4 per A row: +16 B pad keeps ds_read_b128 conflict-free */;
    constexpr int WM = TM / WAVES_M, WN = 128 / WAVES_N, MT = WM / 32, NT = WN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && MT >= 1 && NT >= 1, "4 waves");
    __shared__ f32x4 As4[TM * LDA4];
    __shared__ f32x4 Bs4[BK * 32];
    __shared__ int s_nbr[27 * TM];
    __shared__ int s_row[TM];  // output row of each tile slot: row0 + r, or perm[row0 + r] for a scheduled launch (sched.hip)
    __shared__ unsigned s_kmask;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int wrow0 = (wid / WAVES_N) * WM, wcol0 = (wid % WAVES_N) * WN;
    const int M_out = m_dev ? min(*m_dev, M_bound) : M_bound;  // launched over the bound, the true row count lives on the device
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD x takes the x-th
    // contiguous eighth of the tiles: neighbouring tiles (whose gathers overlap) then share one L2 instead of eight.
    const int ntile = (M_out + TM - 1) / TM, per_xcd = (ntile + 7) / 8;
    int tile = (xcd_remap && !tile_map) ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile_map) {
        tile = tile_map[blockIdx.x];
        if (tile < 0) return;
    }
    const int64_t row0 = (int64_t)tile * TM;
    if (tile >= ntile || (!tile_map && xcd_remap && (int)(blockIdx.x >> 3) >= per_xcd)) return;
    int ks0 = 0, ks1 = K, S = 1;  // this workgroup's offsets
    if (SPLIT) {
        S = split_count(ntile, smax);
        if ((int)blockIdx.y >= S) return;
        ks0 = K * (int)blockIdx.y / S;
        ks1 = K * ((int)blockIdx.y + 1) / S;
    }

    if (tid == 0) s_kmask = 0;
    if (tid < TM) {
        const int64_t slot = row0 + tid;
        int r = slot < M_out ? (perm ? perm[slot] : (int)slot) : M_out;
        if ((unsigned)r > (unsigned)M_out) r = M_out;  // a row index outside the launch (a corrupted schedule): a dead slot, never an address
        s_row[tid] = r;
    }
    __syncthreads();
    unsigned mymask = 0;
    for (int idx = tid; idx < K * TM; idx += 256) {
        int k = idx / TM, r = idx - k * TM;
        const int row = s_row[r];
        int ks = mirror ? K - 1 - k : k;
        int v = (row < M_out && (!SPLIT || (k >= ks0 && k < ks1))) ? nbr[(int64_t)ks * ld + row] : -1;
        s_nbr[idx] = v;
        if (v >= 0) mymask |= 1u << k;
    }
    mymask = wave_or_u(mymask);  // one LDS atomic per wave (a per-lane atomicOr compiles to a serial loop over the active lanes, ~1 us)
    if (lane == 0 && mymask) atomicOr(&s_kmask, mymask);
    __syncthreads();
    const unsigned kmask = s_kmask;
    // which offsets have a neighbour for THIS wave's rows
    unsigned wmask = 0;
    for (int k = 0; k < K; ++k) {
        if (!((kmask >> k) & 1u)) continue;
        bool v = false;
        for (int r = lane; r < WM; r += 64) v |= s_nbr[k * TM + wrow0 + r] >= 0;
        if (__ballot(v)) wmask |= 1u << k;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const f32x4* X4 = (const f32x4*)X;
    const f32x4* W4 = (const f32x4*)W;
    const float* Bs = (const float*)Bs4;

    // Steps = (offset with a neighbour in this tile) x (four 32-channel chunks).  The global loads of step s+1 are issued into registers
    // BEFORE the MFMAs of step s and written to LDS after them, so their latency hides under the matrix work.
    f32x4 ra[TM / 32], rb[4];
    auto load_regs = [&](int k, int kc) {
#pragma unroll
        for (int j = 0; j < TM / 32; ++j) {
            int idx = j * 256 + tid;
            int nb = s_nbr[k * TM + (idx >> 3)];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (nb >= 0) v = X4[(int64_t)nb * 32 + kc * 8 + (idx & 7)];
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int idx = j * 256 + tid;
            rb[j] = W4[((int64_t)k * 128 + kc * 32 + (idx >> 5)) * 32 + (idx & 31)];
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int j = 0; j < TM / 32; ++j) {
            int idx = j * 256 + tid;
            As4[(idx >> 3) * LDA4 + (idx & 7)] = ra[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Bs4[j * 256 + tid] = rb[j];
    };
    int k = kmask ? __builtin_ctz(kmask) : -1, kc = 0;
    if (k >= 0) {
        load_regs(k, 0);
        store_regs();
        __syncthreads();
    }
    while (k >= 0) {
        int nk = k, nkc = kc + 1;
        if (nkc == 4) {
            nkc = 0;
            unsigned rest = k + 1 < 32 ? kmask >> (k + 1) : 0u;
            nk = rest ? k + 1 + __builtin_ctz(rest) : -1;
        }
        if (nk >= 0) load_regs(nk, nkc);
        if ((wmask >> k) & 1u) {
            f32x4 a[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[m][j] = As4[(wrow0 + m * 32 + i32) * LDA4 + h * 4 + j];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float b[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n] = Bs[(h * 16 + s) * 128 + wcol0 + n * 32 + i32];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s >> 2][s & 3], b[n], acc[m][n], 0, 0, 0);
            }
        }
        __syncthreads();
        if (nk < 0) break;
        store_regs();
        __syncthreads();
        k = nk;
        kc = nkc;
    }
    // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = s_row[wrow0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                float* out = S > 1 ? part + (int64_t)blockIdx.y * M_bound * 128 : Y;
                if (row < M_out) out[row * 128 + wcol0 + n * 32 + i32] = acc[m][n][r];
            }
}

// Y[row] = part[0][row] + part[1][row] + ... in split order (all nine loads of a lane in flight); 8 rows per workgroup
__global__ __launch_bounds__(256) void k_split_sum(const f32x4* __restrict__ part, int M_bound, const int* __restrict__ m_dev, int smax, f32x4* __restrict__ Y) {
    const int M_out = m_dev ? min(*m_dev, M_bound) : M_bound;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= M_out) return;
    const int S = split_count((M_out + 31) / 32, smax), c4 = threadIdx.x & 31;
    if (S == 1) return;  // the tiles went straight into Y
    f32x4 v[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) v[q] = part[((int64_t)min(q, S - 1) * M_bound + row) * 32 + c4];
    f32x4 t = v[0];
#pragma unroll
    for (int q = 1; q < 9; ++q)
        if (q < S) t += v[q];
    Y[row * 32 + c4] = t;
}

// Row-private variant for large launches: wave w of the workgroup owns output rows [32w, 32w+32) of the tile and ALL 128 columns
// (four independent 32x32 accumulators; lane i32 of column block n holds column 4*i32 + n, so one ds_read_b128 feeds four MFMAs and
// the epilogue stores float4s).  The gathered A rows are private to the wave and go global -> registers (no LDS, no barrier); only
// the 16 KB weight chunk of a step is shared, double-buffered in LDS: ONE barrier per step, and the loads of step s+1 (A into the
// other register set, W into staging registers) are in flight under the 64 MFMAs of step s.  The k-index assignment and the
// order of the accumulation are those of k_conv128, so the two kernels give bit-identical results.
#ifdef DD3D_CONV_PROF  // `make prof`: wave 0..3 of the first 1024 workgroups leave timestamps (100 MHz wall clock) and their placement
__device__ long long g_prof[1024 * 4 * 8];
#define PROF_T(x) const long long x = wall_clock64()
#define PROF_ADD(acc, a, b) acc += (b) - (a)
#else
#define PROF_T(x)
#define PROF_ADD(acc, a, b)
#endif
// MIRRORED only tags the instantiation (the mirrored = backward-data launch of a step runs beside the kernel gradient on purpose, the
// forward launch alone: two symbols keep their durations apart in a kernel trace); the code is the same, `mirror` stays a runtime argument.
template <int NW, int MIRRORED = 0>
__global__ __launch_bounds__(NW * 64) void k_conv128_rows(const float* __restrict__ X, const float* __restrict__ W, const int* __restrict__ nbr, int64_t ld,
                                                          int K, int M_bound, const int* __restrict__ m_dev, int mirror, float* __restrict__ Y, int xcd_remap,
                                                          const int* __restrict__ perm, const int* __restrict__ tile_map,
                                                          const int* __restrict__ nbr_sorted, const unsigned* __restrict__ tile_mask) {
    constexpr int TM = NW * 32, NTH = NW * 64, BPT = 1024 / NTH;
    __shared__ f32x4 Bs4[2][32 * 32];
    __shared__ int s_nbr[27 * TM];
    __shared__ int s_row[TM];  // output row of each tile slot: row0 + r, or perm[row0 + r] (rows grouped by neighbour pattern, see k_row_order)
    __shared__ unsigned s_kmask;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int M_out = m_dev ? min(*m_dev, M_bound) : M_bound;
    const int ntile = (M_out + TM - 1) / TM, per_xcd = (ntile + 7) / 8;
    int tile = (xcd_remap && !tile_map) ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile_map) {  // scheduled launch (sched.hip): workgroup -> tile of the pattern-sorted row order, balanced over the CUs
        tile = tile_map[blockIdx.x];
        if (tile < 0) return;
    }
    const int64_t row0 = (int64_t)tile * TM;
    if (tile >= ntile || (!tile_map && xcd_remap && (int)(blockIdx.x >> 3) >= per_xcd)) return;

    PROF_T(pt0);
    // Neighbour table of the tile: thread t takes tile slot t % TM and every second offset, all of its table reads in flight together.
    // A scheduled launch brings the table in pattern order (nbr_sorted: the slots of a tile are consecutive, coalesced reads) and the
    // offsets present in the tile (tile_mask: the others are not read at all); without them the columns are gathered through perm --
    // scattered 4-byte reads that cost a tile of the sorted order ~12 us before its first MFMA.
    constexpr int KPT = (27 + 1) / 2;
    static_assert(NTH == 2 * TM, "two threads per tile slot");
    const int slot_r = tid & (TM - 1), k_first = tid / TM;
    const int64_t slot = row0 + slot_r;
    int my_row = slot < M_out ? (perm ? perm[slot] : (int)slot) : M_out;  // slots past the end are dead
    if ((unsigned)my_row > (unsigned)M_out) my_row = M_out;                     // (and so is a row index outside the launch)
    unsigned tmask = tile_mask ? tile_mask[tile] : 0x7fffffffu;
    const bool in_order = nbr_sorted && !(tmask >> 31);
    const int* nbr_col = in_order ? nbr_sorted + min(slot, (int64_t)M_out - 1) : nbr + min(my_row, M_out - 1);  // clamped, no branch on the row
    int v_nbr[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int kk = min(k_first + 2 * j, K - 1), ks = mirror ? K - 1 - kk : kk;  // table row ks serves offset kk of a mirrored launch
        v_nbr[j] = -1;
        if ((tmask >> ks) & 1u) v_nbr[j] = nbr_col[(int64_t)ks * ld];  // (uniform per wave: k_first is)
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j)
        if (k_first + 2 * j >= K || my_row >= M_out) v_nbr[j] = -1;
    if (tid == 0) s_kmask = 0;
    if (tid < TM) s_row[tid] = my_row;
    __syncthreads();
    unsigned mymask = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int kk = k_first + 2 * j;
        if (kk < K) s_nbr[kk * TM + slot_r] = v_nbr[j];
        if (v_nbr[j] >= 0) mymask |= 1u << kk;
    }
    mymask = wave_or_u(mymask);  // one LDS atomic per wave (a per-lane atomicOr compiles to a serial loop over the active lanes, ~1 us)
    if (lane == 0 && mymask) atomicOr(&s_kmask, mymask);
    __syncthreads();
    const unsigned kmask = s_kmask;
    // (No per-wave skip of an offset none of the wave's 32 rows has: in pattern order a tile's union is 1.03x the algorithmic work, and the
    // branch around the MFMA block made the compiler shuttle the accumulators between register files every step.)
#ifdef DD3D_CONV_PROF
    unsigned wmask = 0;  // offsets with a neighbour for one of THIS wave's 32 rows
    for (unsigned rest = kmask; rest; rest &= rest - 1) {
        const int k = __builtin_ctz(rest);
        if (__ballot(s_nbr[k * TM + wid * 32 + i32] >= 0)) wmask |= 1u << k;
    }
#endif

    PROF_T(pt1);
    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const f32x4* X4 = (const f32x4*)X;
    const f32x4* W4 = (const f32x4*)W;
#ifdef DD3D_CONV_PROF
    long long sg0 = 0, sg1 = 0, sg2 = 0, sg3 = 0;
#endif

    static_assert(BPT <= 4, "eight load slots per step: four gathered A quads, up to four weight quads");
    f32x4 cur[4], nxt[4], rb[BPT];
    int k = kmask ? __builtin_ctz(kmask) : -1, kc = 0, nb = -1, buf = 0;
    if (k >= 0) {
        nb = s_nbr[k * TM + wid * 32 + i32];
        const f32x4* pa = X4 + (int64_t)max(nb, 0) * 32 + h * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = pa[j];
#pragma unroll
        for (int j = 0; j < BPT; ++j) Bs4[0][j * NTH + tid] = W4[(int64_t)k * 128 * 32 + j * NTH + tid];
        __syncthreads();
    }
    // One step = 16 k-pairs x 4 column blocks = 64 MFMAs on the weight chunk in Bs4[buf].  A wave issues its MFMAs in order and the pipe
    // takes one every 64 cycles, so whatever else the wave has to issue must sit BETWEEN them to cost nothing: the weight fragments of
    // k-pairs s+2, s+3 are read from LDS while s, s+1 multiply, and the eight global loads of the NEXT step (four quads of the wave's
    // gathered rows, clamped address and no branch, and its share of the next weight chunk) go out one per two k-pairs.
    // Every step runs the same code, the last one included (it "prefetches" its own operands again and publishes them to nobody): with
    // the loads and the store under two separate `next exists` tests the compiler's wait-count pass saw a path "loads issued, store
    // skipped, loop again" and put waits for a step's own prefetch in front of its MFMAs.
    while (k >= 0) {
        PROF_T(q0);
        // the step after (k, kc): the next 32-channel chunk, or the first chunk of the next offset present in the tile
        int nk = k, nkc = kc + 1, nbn = nb;
        if (nkc == 4) {
            nkc = 0;
            const unsigned rest = k + 1 < 32 ? kmask >> (k + 1) : 0u;
            nk = rest ? k + 1 + __builtin_ctz(rest) : -1;
            if (nk >= 0) nbn = s_nbr[nk * TM + wid * 32 + i32];
        }
        const bool more = nk >= 0;
        if (!more) nk = k, nkc = kc;
        const f32x4* pa = X4 + (int64_t)max(nbn, 0) * 32 + nkc * 8 + h * 4;
        const f32x4* pw = W4 + ((int64_t)nk * 128 + nkc * 32) * 32 + tid;
        PROF_T(q1);
        {
            const bool valid = nb >= 0;
            const f32x4* Bp = &Bs4[buf][(h * 16) * 32 + i32];
            f32x4 b0 = Bp[0], b1 = Bp[32];
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                f32x4 n0 = b0, n1 = b1;
                if (s + 2 < 16) n0 = Bp[(s + 2) * 32], n1 = Bp[(s + 3) * 32];
                if (s == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = pa[j];
                }
                if (s < 4) {  // all of the prefetch is out after the first quarter of the step: a microsecond before it is needed
#pragma unroll
                    for (int j = s / 2 * ((BPT + 1) / 2); j < (s / 2 + 1) * ((BPT + 1) / 2) && j < BPT; ++j) rb[j] = pw[j * NTH];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float a0 = valid ? cur[s >> 2][s & 3] : 0.f, a1 = valid ? cur[(s + 1) >> 2][(s + 1) & 3] : 0.f;
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[n], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = n0, b1 = n1;
            }
        }
        PROF_T(q2);
        buf ^= 1;
#pragma unroll
        for (int j = 0; j < BPT; ++j) Bs4[buf][j * NTH + tid] = rb[j];
        PROF_T(q3);
        __syncthreads();
        PROF_T(q4);
        PROF_ADD(sg0, q0, q1);
        PROF_ADD(sg1, q1, q2);
        PROF_ADD(sg2, q2, q3);
        PROF_ADD(sg3, q3, q4);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        k = more ? nk : -1, kc = nkc, nb = nbn;
    }
    PROF_T(pt2);
    // epilogue: C/D layout of a 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); block n, lane col i32 <-> column 4*i32 + n
    f32x4* Y4 = (f32x4*)Y;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = s_row[wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        if (row < M_out) Y4[row * 32 + i32] = v;
    }
#ifdef DD3D_CONV_PROF
    if (lane == 0 && blockIdx.x < 1024) {  // [start, table done, loop done, end, steps | segment sums, wave steps | ..., HW_ID, XCC_ID]
        long long* o = g_prof + ((int64_t)blockIdx.x * 4 + wid) * 8;
        o[0] = pt0, o[1] = pt1, o[2] = pt2, o[3] = wall_clock64();
        o[4] = __builtin_popcount(kmask) | (sg0 << 8) | (sg1 << 32), o[5] = __builtin_popcount(wmask) | (sg2 << 8) | (sg3 << 32);
        o[6] = __builtin_amdgcn_s_getreg(4 | (31 << 11)), o[7] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
#endif
}
#ifdef DD3D_CONV_PROF
}  // namespace
DD3D_API int dd3d_prof_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)); }
namespace {
#endif

// ---------------------------------------------------------------------------------------------
// Cin = Cout = 128 backward-weight: one workgroup per (row chunk, offset k) -> partial slab [128][128]
// ---------------------------------------------------------------------------------------------
// *mask |= bit k for every offset k that some row below group_row_off[G] has (mask zeroed before): k_conv128_bww launches a bound-sized
// grid over (chunk, offset) and the offsets that exist nowhere -- 18 of 27 for flat clouds -- must cost their workgroups nothing.
__global__ __launch_bounds__(256) void k_offset_mask(const int* __restrict__ nbr, int64_t ld, int K, const int* __restrict__ group_row_off, int G,
                                                     unsigned* __restrict__ mask) {
    const int M = group_row_off[G];
    unsigned m = 0;
    for (int row = blockIdx.x * 256 + threadIdx.x; row < M; row += gridDim.x * 256) {
        int v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = nbr[(int64_t)min(k, K - 1) * ld + row];
#pragma unroll
        for (int k = 0; k < 27; ++k) m |= (unsigned)(k < K && v[k] >= 0) << k;
    }
    m = wave_or_u(m);
    // bits are only ever added: a workgroup that brings nothing new (all but the first few) skips the atomic; a stale read costs a
    // needless atomic at worst
    if ((threadIdx.x & 63) == 0 && (m & ~__atomic_load_n(mask, __ATOMIC_RELAXED))) atomicOr(mask, m);
}

__global__ __launch_bounds__(256, 3) void k_conv128_bww(const float* __restrict__ X, const float* __restrict__ dY, const int* __restrict__ nbr, int64_t ld,
                                                     int K, const int* __restrict__ group_row_off, int G, float* __restrict__ slabs,
                                                     int* __restrict__ slab_valid, const unsigned* __restrict__ active, int order) {
    // The (in,out) pairs of offset k inside the chunk are first COMPACTED (stable ballot scan, 1024 rows per window), so the matrix
    // cores only see rows that have a neighbour; the contraction then runs over 32 pairs per step, in the shape of k_conv128_rows:
    // wave w owns input channels [32w, 32w+32) and all 128 output channels (four 32x32 accumulators); its A operand -- X^T of the
    // step's 32 gathered rows, one float per lane and row pair -- goes global -> registers; the 32 dY rows of the step (16 KB) are
    // shared by the four waves through a double-buffered LDS chunk: ONE barrier per step, the loads of step s+1 in flight under the
    // 64 MFMAs of step s and issued between them.
    __shared__ f32x4 Bs4[2][32 * 32];
    __shared__ __align__(16) int s_row[1024 + 32], s_nb[1024 + 32];
    __shared__ int s_wc[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    // blockIdx.y = index among the offsets present anywhere (k_offset_mask); the grid's x extent is a multiple of 8, so the workgroups
    // of one chunk -- which read the same dY rows and nearly the same X rows -- land on one XCD and share its L2
    unsigned am = *active;
    int c, kidx;
    if (order) {
        // chunk-major dispatch: workgroups are dealt to the 8 XCDs round-robin in linear order, so linear id L -> XCD L & 7, and the
        // `nact` workgroups that share chunk c (same dY rows, nearly the same X rows) are CONSECUTIVE on one XCD: they run at the same
        // time and the chunk's operands are fetched into that XCD's L2 once instead of once per offset
        const int nact = __popc(am), L = blockIdx.x + gridDim.x * blockIdx.y, j = L >> 3;
        if (nact == 0) return;
        kidx = j % nact;
        c = (j / nact) * 8 + (L & 7);
        if (c >= (int)gridDim.x) return;
    } else {
        if ((int)blockIdx.y >= __popc(am)) return;
        kidx = blockIdx.y, c = blockIdx.x;
    }
    for (int j = 0; j < kidx; ++j) am &= am - 1;
    const int k = __builtin_ctz(am);
    int g, r0, r1;
    const int CH = chunk128(group_row_off[G]);
    if (!resolve_chunk(c, group_row_off, G, CH, &g, &r0, &r1)) return;
    const int* nk = nbr + (int64_t)k * ld;

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float* Xc = X + wid * 32 + i32;  // this lane's input channel
    const f32x4* D4 = (const f32x4*)dY;
    bool have_any = false;
    PROF_T(pt0);
#ifdef DD3D_CONV_PROF
    long long t_steps = 0, n_steps = 0, sg1 = 0, sg2 = 0, sg3 = 0;
#endif

    for (int wb = r0; wb < r1; wb += 1024) {
        // ---- compaction of the window: all table reads first, one pass of ballots ----
        int nbv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wb + j * 256 + tid;
            nbv[j] = row < r1 ? nk[row] : -1;
        }
        __syncthreads();  // the previous window's lists are no longer read
        unsigned long long bal[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bal[j] = __ballot(nbv[j] >= 0);
            if (lane == 0) s_wc[j * 4 + wid] = __popcll(bal[j]);
        }
        __syncthreads();
        int total = 0, base[4];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int x = s_wc[q];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q == j * 4 + wid) base[j] = total;
            total += x;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (nbv[j] >= 0) {
                const int pos = base[j] + __popcll(bal[j] & ((1ull << lane) - 1ull));
                s_row[pos] = wb + j * 256 + tid;
                s_nb[pos] = nbv[j];
            }
        if (total == 0) continue;  // (uniform)
        have_any = true;
        __syncthreads();
        if (tid < 32) {  // ... the tail of the last step reads the last pair again; its A operand is zeroed at use)
            s_row[total + tid] = s_row[total - 1];
            s_nb[total + tid] = s_nb[total - 1];
        }
        __syncthreads();
        // pairs past the end of the list (the last step's tail): clamped to the last pair, their A operand is zeroed at use
        const int steps = (total + 31) / 32, last = total - 1;
        float cur[16], nxt[16];
        f32x4 rb[4];
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) cur[q] = Xc[(int64_t)s_nb[min(h * 16 + q, last)] * 128];
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs4[0][j * 256 + tid] = D4[(int64_t)s_row[min(j * 8 + (tid >> 5), last)] * 32 + (tid & 31)];
        }
        __syncthreads();
        int buf = 0;
        PROF_T(q0);
        for (int sb = 0; sb < steps; ++sb) {
            // the next step (the last one "prefetches" itself again: one code path, see k_conv128_rows)
            PROF_T(u0);
            const int nbase = (sb + 1 < steps ? sb + 1 : sb) * 32;
            const int live = total - sb * 32 - h * 16;  // pairs of this lane's half that exist in this step
            const f32x4* Bp = &Bs4[buf][(h * 16) * 32 + i32];
            // the next step's rows: all of its loads go out in the first half of this step (the step ends waiting for them: a load
            // issued behind the last MFMAs would expose its whole latency)
            const int4* ia = (const int4*)(s_nb + nbase + h * 16);  // (the lists are padded by 32 entries: no clamp on the address ...
            const int* ib = s_row + nbase + (tid >> 5);
            f32x4 b0 = Bp[0], b1 = Bp[32];
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                f32x4 n0 = b0, n1 = b1;
                if (q + 2 < 16) n0 = Bp[(q + 2) * 32], n1 = Bp[(q + 3) * 32];
                if (q < 8) {
                    const int4 ix = ia[q / 2];
                    nxt[2 * q] = Xc[(int64_t)ix.x * 128];
                    nxt[2 * q + 1] = Xc[(int64_t)ix.y * 128];
                    nxt[2 * q + 2] = Xc[(int64_t)ix.z * 128];
                    nxt[2 * q + 3] = Xc[(int64_t)ix.w * 128];
                    rb[q / 2] = D4[(int64_t)ib[q / 2 * 8] * 32 + (tid & 31)];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float a0 = q < live ? cur[q] : 0.f, a1 = q + 1 < live ? cur[q + 1] : 0.f;
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[n], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = n0, b1 = n1;
            }
            PROF_T(u1);
            buf ^= 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs4[buf][j * 256 + tid] = rb[j];
            PROF_T(u2);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
#ifdef DD3D_CONV_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PROF_T(u3);
            PROF_ADD(sg1, u0, u1);
            PROF_ADD(sg2, u1, u2);
            PROF_ADD(sg3, u2, u3);
        }
        PROF_T(q1);
        PROF_ADD(t_steps, q0, q1);
        PROF_ADD(n_steps, 0, steps);
    }
    if (!have_any) {
        if (tid == 0) slab_valid[(int64_t)c * K + k] = 0;  // every resolved (chunk, offset) writes its flag: no memset of the flags
        return;
    }
    // C/D layout of a 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); block n, lane col i32 <-> output channel 4*i32 + n
    f32x4* slab4 = (f32x4*)(slabs + ((int64_t)c * K + k) * (128 * 128));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        slab4[ci * 32 + i32] = v;
    }
    if (tid == 0) slab_valid[(int64_t)c * K + k] = 1;
#ifdef DD3D_CONV_PROF
    {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && wid == 0 && lin < 4096) {  // [start, end, time in step loops, steps, HW_ID, XCC_ID]
            long long* o = g_prof + (int64_t)lin * 8;
            o[0] = pt0, o[1] = wall_clock64(), o[2] = t_steps, o[3] = n_steps;
            o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11)), o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)), o[6] = sg1 | (sg2 << 32), o[7] = sg3;
        }
    }
#endif
}

// dW[g][k][e] = sum over the valid slabs of group g.  4 slot-lanes walk the slots c = first+lane, first+lane+4, ... (4 independent chains), then the lanes are combined in lane order: deterministic.
__global__ __launch_bounds__(256) void k_slab_reduce(const f32x4* __restrict__ slabs, const int* __restrict__ slab_valid,
                                                     const int* __restrict__ group_row_off, int G, int cin3, int K, int elems4, f32x4* __restrict__ dW,
                                                     const unsigned* __restrict__ active) {
    __shared__ f32x4 s_red[256];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int k = blockIdx.y, g = blockIdx.z;
    if (active && !((*active >> k) & 1u)) {  // an offset no row has: no slab of it was written (nor its flags)
        if (sl == 0 && e < elems4) dW[((int64_t)g * K + k) * elems4 + e] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int CH = cin3 ? chunk3(group_row_off[G]) : chunk128(group_row_off[G]);
    int first = 0;
    for (int j = 0; j < g; ++j) first += (group_row_off[j + 1] - group_row_off[j] + CH - 1) / CH;
    const int n = (group_row_off[g + 1] - group_row_off[g] + CH - 1) / CH;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s = zero;
    const int ec = min(e, elems4 - 1);
    if (cin3) {  // every slot of a group holds a slab: unconditional loads, several in flight
#pragma unroll 4
        for (int c = first + sl; c < first + n; c += 4) s += slabs[((int64_t)c * K + k) * elems4 + ec];
    } else {     // most (chunk, offset) slabs of flat clouds do not exist: never touch those
        for (int c = first + sl; c < first + n; c += 32) {  // eight of this lane's slots in flight (flags, then the slabs that exist), added in slot order
            int ok[8];
            f32x4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cq = c + 4 * q;
                ok[q] = slab_valid[(int64_t)min(cq, first + n - 1) * K + k];
                if (cq >= first + n) ok[q] = 0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = ok[q] ? slabs[((int64_t)(c + 4 * q) * K + k) * elems4 + ec] : zero;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (ok[q]) s += x[q];
        }
    }
    s_red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && e < elems4) dW[((int64_t)g * K + k) * elems4 + e] = s_red[el] + s_red[el + 64] + s_red[el + 128] + s_red[el + 192];
}

// ---------------------------------------------------------------------------------------------
// Cin = 3, Cout = 128 (first layer): bandwidth kernels on the VALU
struct __attribute__((packed, aligned(4))) f32x3 { float x, y, z; };
// ---------------------------------------------------------------------------------------------
// Y[o][:] = sum_k X[nbr[k][o]][0:3] . W[k][0:3][:] as a [64 x 96] x [96 x 128] product per tile on the fp32 matrix cores.  The
// im2col tile A[o][3 pos(k) + ci] is gathered into LDS with coalesced neighbour-table reads (64 consecutive rows per offset); the
// wave's slice of W sits in registers.  Columns are ordered so that the nine offsets with no displacement along the first spatial
// axis come first (pos < 9): clouds that are flat along that axis (MNIST: depth 0) then fill only the first of the three
// 32-column chunks, and chunks without any neighbour in the tile are skipped as a whole.
__device__ __forceinline__ int conv3_pos(int k) { return (k % 3 == 1 ? 0 : k % 3 == 0 ? 9 : 18) + k / 3; }   // offset -> column group
__device__ __forceinline__ int conv3_off(int p) { return (p % 9) * 3 + (p < 9 ? 1 : p < 18 ? 0 : 2); }       // column group -> offset

__global__ __launch_bounds__(256, 3) void k_conv3_fwd(const float* __restrict__ X, const float* __restrict__ W /*[K][3][128]*/, const int* __restrict__ nbr,
                                                   int64_t ld, int K, int M_bound, const int* __restrict__ m_dev, float* __restrict__ Y) {
    constexpr int LDA = 97;
    __shared__ float As[64 * LDA];
    __shared__ __align__(16) unsigned s_pm[4];  // column groups present in the tile, one word per wave (ballots: no LDS atomics)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int M_out = m_dev ? min(*m_dev, M_bound) : M_bound;
    if ((int64_t)blockIdx.x * 64 >= M_out) return;
    // The wave's slice of W (96 columns x its 32 output channels) lives in REGISTERS for the whole kernel: lane (i32, h) holds exactly
    // the B operands it feeds to the 48 MFMA steps.  48 coalesced loads per lane, issued together, once per workgroup.
    // Columns 0..63 go in two chunks of 16 steps (lane half h takes 16 consecutive columns of a chunk); the last 17 columns (+1 of
    // zeros) in a chunk of 9 steps, lane half h taking columns 64 + 9h ..: 41 MFMA steps per tile, not 48.
    float wv[41];
#pragma unroll
    for (int t = 0; t < 41; ++t) {
        const int j = t < 32 ? (t >> 4) * 32 + h * 16 + (t & 15) : 64 + h * 9 + (t - 32);  // column of step t for this lane half
        const int wrow = conv3_off(min(j, 80) / 3) * 3 + min(j, 80) % 3;
        wv[t] = W[wrow * 128 + wid * 32 + i32];
        if (j >= 81) wv[t] = 0.f;
    }
    for (int i = tid; i < 64 * LDA; i += 256) As[i] = 0.f;  // columns 81..96 stay zero
    const int sr = tid & 63, sk = __builtin_amdgcn_readfirstlane(tid >> 6);  // staging: row of the tile, offset lane (4 lanes walk the 27 offsets)
    // Two-stage software pipeline over the workgroup's tiles: while the matrix cores work on tile t, the gathers of tile t+1 (through the
    // table entries read one iteration ago) and the table reads of tile t+2 are in flight; nothing is waited for in the iteration
    // that issues it.
    int nb[7], nb2[7];
    float xv[7][3];
    const int64_t tstep = (int64_t)gridDim.x * 64;
    auto load_nb = [&](int64_t row0, int* dst) {  // clamped: every load has a valid address, no branches
        const int64_t row = min(row0 + sr, (int64_t)M_out - 1);
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) dst[kk] = nbr[(int64_t)min(sk + kk * 4, 26) * ld + row];
    };
    auto gather = [&]() {
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
            const f32x3 x = *(const f32x3*)(X + (int64_t)max(nb[kk], 0) * 3);  // one 12-B load, not three dwords
            xv[kk][0] = x.x; xv[kk][1] = x.y; xv[kk][2] = x.z;
        }
    };
    load_nb((int64_t)blockIdx.x * 64, nb);
    load_nb((int64_t)blockIdx.x * 64 + tstep, nb2);
    gather();
    for (int64_t row0 = (int64_t)blockIdx.x * 64; row0 < M_out; row0 += tstep) {  // persistent: tiles strided by the grid
        lds_barrier();  // previous tile's MFMA reads of As (and of s_pm) are done; zero fill visible.  LDS-only: the output stores stay in flight
        unsigned wmask = 0;  // wave-uniform: the wave stages offsets wid, wid + 4, ...
        const bool live = row0 + sr < M_out;
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
            const int k = sk + kk * 4;
            if (k < 27) {
                const bool ok = live && nb[kk] >= 0;
                const int p = conv3_pos(k);
                if (__ballot(ok)) wmask |= 1u << p;
                As[sr * LDA + p * 3 + 0] = ok ? xv[kk][0] : 0.f;
                As[sr * LDA + p * 3 + 1] = ok ? xv[kk][1] : 0.f;
                As[sr * LDA + p * 3 + 2] = ok ? xv[kk][2] : 0.f;
            }
        }
        if (lane == 0) s_pm[wid] = wmask;
        lds_barrier();
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) nb[kk] = nb2[kk];
        gather();                          // tile t+1
        load_nb(row0 + 2 * tstep, nb2);    // tile t+2
        const uint4 pm4 = *(const uint4*)s_pm;
        const unsigned pmask = __builtin_amdgcn_readfirstlane(pm4.x | pm4.y | pm4.z | pm4.w);  // wave-uniform: chunk skipping is a scalar branch
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            // column groups touched by chunk q: columns [32q, 32q+32) = groups [32q/3, (32q+31)/3]
            const unsigned groups = ((2u << ((q * 32 + 31) / 3)) - 1u) & ~((1u << (q * 32 / 3)) - 1u);
            if (!(pmask & groups)) continue;
            const int ns = q < 2 ? 16 : 9, c0 = q * 32 + h * ns;
#pragma unroll
            for (int s0 = 0; s0 < ns; s0 += 8) {  // 8 steps of operands at a time: 16 registers, three workgroups per CU
                float av[2][8];
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        if (s0 + s2 < ns) av[m][s2] = As[(m * 32 + i32) * LDA + c0 + s0 + s2];
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        if (s0 + s2 < ns) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s2], wv[q * 16 + s0 + s2], acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = row0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (orow < M_out) Y[orow * 128 + wid * 32 + i32] = acc[m][r];
            }
    }
}

// dX[i][ci] = sum_k sum_co dY[nbr[K-1-k][i]][co] * W[k][ci][co]
// Half a wave per row (32 lanes x float4 = the 128 channels of a gathered dY row), 8 rows per workgroup pass.  The launch is small (the
// synthetic prefix only: the real batch needs no input gradient of the first layer) and latency-bound, so the 27 gathers of a row are
// issued three at a time with clamped addresses (no branch between them) and a workgroup makes ONE pass when the grid allows it.
__global__ __launch_bounds__(256) void k_conv3_bwd_data(const float* __restrict__ dY, const float* __restrict__ W, const int* __restrict__ nbr, int64_t ld,
                                                        int K, int M_bound, const int* __restrict__ m_dev, int rows_per_block, float* __restrict__ dX) {
    __shared__ f32x4 Ws4[27 * 3 * 32];
    const int tid = threadIdx.x;
    const int M = m_dev ? min(*m_dev, M_bound) : M_bound;
    if ((int64_t)blockIdx.x * rows_per_block >= M) return;
    const f32x4* W4 = (const f32x4*)W;
    for (int i = tid; i < K * 3 * 32; i += 256) Ws4[i] = W4[i];
    __syncthreads();
    const int c4 = tid & 31, rl = tid >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
    const f32x4* D4 = (const f32x4*)dY;
    // A half-wave takes a row: lane c4 < 27 holds the row's neighbour at (mirrored) offset c4.  Only the offsets that EXIST are visited
    // (7 of 27 on average, at most 9 for flat clouds), nine at a time, their gathers issued together: one round trip per row for flat
    // clouds instead of nine.  Ascending offsets, present ones only: the sums are the same bits as a loop over all 27.
    auto load_nb = [&](int it) {
        const int64_t row = row0 + it + rl;
        return (it < rows_per_block && row < M && c4 < K) ? nbr[(int64_t)(K - 1 - c4) * ld + row] : -1;
    };
    int mynb = load_nb(0);
    for (int it = 0; it < rows_per_block; it += 8) {
        const int64_t row = row0 + it + rl;
        const bool live = row < M;
        const int nbn = load_nb(it + 8);  // the next row's table entries are in flight under this row's gathers
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        unsigned hm = (unsigned)(__ballot(mynb >= 0) >> (tid & 32));  // this half-wave's offsets
        while (hm) {
            int ks[9];
            f32x4 d[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                ks[j] = hm ? __ffs(hm) - 1 : -1;
                hm &= hm - 1;  // (0 stays 0)
                const int nb = __shfl(mynb, max(ks[j], 0), 32);
                d[j] = D4[(int64_t)max(nb, 0) * 32 + c4];
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (ks[j] >= 0) {
                    const int k = ks[j];
                    const f32x4 w0 = Ws4[(k * 3 + 0) * 32 + c4], w1 = Ws4[(k * 3 + 1) * 32 + c4], w2 = Ws4[(k * 3 + 2) * 32 + c4];
                    s0 += d[j][0] * w0[0] + d[j][1] * w0[1] + d[j][2] * w0[2] + d[j][3] * w0[3];
                    s1 += d[j][0] * w1[0] + d[j][1] * w1[1] + d[j][2] * w1[2] + d[j][3] * w1[3];
                    s2 += d[j][0] * w2[0] + d[j][1] * w2[1] + d[j][2] * w2[2] + d[j][3] * w2[3];
                }
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 32);
            s1 += __shfl_xor(s1, o, 32);
            s2 += __shfl_xor(s2, o, 32);
        }
        if (live && c4 == 0) {
            dX[row * 3 + 0] = s0;
            dX[row * 3 + 1] = s1;
            dX[row * 3 + 2] = s2;
        }
        mynb = nbn;
    }
}

// dW[g][k][ci][co] partial per row chunk on the fp32 matrix cores: out[81 (pad 96)][128] = A^T . dY with the im2col tile
// A[o][3 pos(k) + ci] = X[nbr[k][o]][ci] built in LDS, 32 rows at a time (same column order as k_conv3_fwd, so flat clouds only
// ever fill the first of the three 32-column output tiles and the other two are skipped).  Wave w owns output columns
// [32w, 32w+32) and reads ITS 128 bytes of every dY row straight from global memory in the MFMA operand layout (lane (i32, h) of
// step s2 holds dY[rb + 16 h + s2][32 w + i32]: whole 128-byte lines, each read by exactly one wave) -- dY never passes through
// LDS.  The im2col tile is double-buffered, so a 32-row step has ONE barrier; while the matrix cores work on rows rb.., the dY rows
// and gathers of rb+32.. and the neighbour-table reads of rb+64.. are in flight.  Even and odd MFMA steps of the first output
// tile use separate accumulators (no dependent chain).  <= 168 registers: three workgroups per CU.
__global__ __launch_bounds__(256, 3) void k_conv3_bww(const float* __restrict__ X, const float* __restrict__ dY, const int* __restrict__ nbr, int64_t ld,
                                                      const int* __restrict__ group_row_off, int G, float* __restrict__ slabs, int* __restrict__ slab_valid) {
    constexpr int LDA = 97;  // 96 + 1: the im2col writes walk rows (stride LDA) and must not land on one bank
    __shared__ float As[2][32 * LDA];
    __shared__ __align__(16) unsigned s_pm[2][4];  // column groups present in a buffer's rows, one word per wave (ballots: no LDS atomics)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int c = blockIdx.x;
    PROF_T(pt0);
#ifdef DD3D_CONV_PROF
    long long sg0 = 0, sg1 = 0, sg2 = 0, sg3 = 0, n_it = 0;
#endif
    int g, r0, r1;
    const int CH = chunk3(group_row_off[G]);
    if (!resolve_chunk(c, group_row_off, G, CH, &g, &r0, &r1)) return;
    for (int i = tid; i < 2 * 32 * LDA; i += 256) (&As[0][0])[i] = 0.f;  // pad columns 81..96 stay zero
    f32x16 acc[4];  // output tile 0 (the one flat clouds use) has two accumulators: [0] even steps, [3] odd steps
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int sr = tid & 31, sk = tid >> 5;  // staging: row within the sub-block, offset lane (8 lanes walk the 27 offsets; sk = 2 wid + h)
    const float* dYw = dY + wid * 32 + i32;
    // every load is clamped to the chunk (valid address, no branch); rows beyond r1 are zeroed before they are used
    auto load_nb = [&](int rb, int* nb) {
        const int rowc = min(rb + sr, r1 - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) nb[kk] = nbr[(int64_t)min(sk + kk * 8, 26) * ld + rowc];
    };
    auto load_x = [&](const int* nb, float (*xv)[3]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x3 x = *(const f32x3*)(X + (int64_t)max(nb[kk], 0) * 3);  // one 12-B load, not three dwords
            xv[kk][0] = x.x; xv[kk][1] = x.y; xv[kk][2] = x.z;
        }
    };
    auto load_b = [&](int rb, float* b) {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) b[s2] = dYw[(int64_t)min(rb + h * 16 + s2, r1 - 1) * 128];
    };
    auto store = [&](int rb, const int* nb, const float (*xv)[3], float* as, unsigned* pm) {
        unsigned wmask = 0;  // wave-uniform
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = sk + kk * 8;
            const bool ok = k < 27 && rb + sr < r1 && nb[kk] >= 0;
            const unsigned long long bal = __ballot(ok);  // low half: offset 2 wid + 8 kk, high half: the next one
            if ((unsigned)bal) wmask |= 1u << conv3_pos(2 * wid + kk * 8);
            if ((unsigned)(bal >> 32)) wmask |= 1u << conv3_pos(2 * wid + 1 + kk * 8);
            if (k < 27) {
                const int p = conv3_pos(k);
                as[sr * LDA + p * 3 + 0] = ok ? xv[kk][0] : 0.f;
                as[sr * LDA + p * 3 + 1] = ok ? xv[kk][1] : 0.f;
                as[sr * LDA + p * 3 + 2] = ok ? xv[kk][2] : 0.f;
            }
        }
        if (lane == 0) pm[wid] = wmask;
    };
    int nbA[4], nbB[4];
    float xv[4][3];
    float bv0[16], bv1[16];
    load_nb(r0, nbA);
    load_nb(r0 + 32, nbB);
    load_x(nbA, xv);
    load_b(r0, bv0);
    __syncthreads();  // zero fill visible
    store(r0, nbA, xv, As[0], s_pm[0]);
    __syncthreads();
    PROF_T(pt1);
    // One 32-row step.  `bv` holds the step's dY rows (loaded one step ago), `bvn` receives the next step's; `nbx` holds the table entries of
    // rows rb+32.. (read one step ago: their gathers go out now), `nby` receives those of rb+64...  The two calls per loop trip swap the
    // roles, so no loaded register is ever copied (a copy would wait for the load at the end of the step that issued it).
    auto step = [&](int rb, int par, float* bv, float* bvn, int* nbx, int* nby) {
        PROF_T(q0);
        const float* as = As[par];
        load_x(nbx, xv);
        load_b(rb + 32, bvn);
        load_nb(rb + 64, nby);
        const uint4 pm4 = *(const uint4*)s_pm[par];
        const unsigned pmask = __builtin_amdgcn_readfirstlane(pm4.x | pm4.y | pm4.z | pm4.w);
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            if (rb + h * 16 + s2 >= r1) bv[s2] = 0.f;  // the clamped load repeated the chunk's last row
        PROF_T(q1);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const unsigned groups = ((2u << ((m * 32 + 31) / 3)) - 1u) & ~((1u << (m * 32 / 3)) - 1u);  // column groups of output tile m
            if (!(pmask & groups)) continue;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const float av = as[(h * 16 + s2) * LDA + m * 32 + i32];
                const int ai = (m == 0 && (s2 & 1)) ? 3 : m;
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s2], acc[ai], 0, 0, 0);
            }
        }
        PROF_T(q2);
        if (rb + 32 < r1) store(rb + 32, nbx, xv, As[par ^ 1], s_pm[par ^ 1]);  // the other buffer: its readers passed the last barrier
        PROF_T(q3);
        __syncthreads();
        PROF_T(q4);
        PROF_ADD(sg0, q0, q1);
        PROF_ADD(sg1, q1, q2);
        PROF_ADD(sg2, q2, q3);
        PROF_ADD(sg3, q3, q4);
        PROF_ADD(n_it, 0, 1);
    };
    for (int rb = r0;;) {
        step(rb, 0, bv0, bv1, nbB, nbA);
        rb += 32;
        if (rb >= r1) break;
        step(rb, 1, bv1, bv0, nbA, nbB);
        rb += 32;
        if (rb >= r1) break;
    }
    float* slab = slabs + (int64_t)c * (81 * 128);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // permuted im2col column -> row (k*3 + ci) of the kernel gradient
            if (col < 81) slab[(conv3_off(col / 3) * 3 + col % 3) * 128 + wid * 32 + i32] = m == 0 ? acc[0][r] + acc[3][r] : acc[m][r];
        }
    if (tid == 0) slab_valid[c] = 1;
#ifdef DD3D_CONV_PROF
    if (lane == 0 && wid == 0 && c < 4096) {  // [start, end, loop start, iterations, HW_ID, XCC_ID, issue+dY wait | multiply, store | barrier]
        long long* o = g_prof + (int64_t)c * 8;
        o[0] = pt0, o[1] = wall_clock64(), o[2] = pt1, o[3] = n_it;
        o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11)), o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)), o[6] = sg0 | (sg1 << 32), o[7] = sg2 | (sg3 << 32);
    }
#endif
}

__global__ void k_transpose_kernel(const float* __restrict__ W, int K, int Cin, int Cout, float* __restrict__ Wt) {
    // Wt[k][co][ci] = W[k][ci][co]
    __shared__ float tile[32][33];
    int k = blockIdx.z;
    int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int ci = ci0 + j, co = co0 + tx;
        tile[j][tx] = (ci < Cin && co < Cout) ? W[((int64_t)k * Cin + ci) * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int co = co0 + j, ci = ci0 + tx;
        if (ci < Cin && co < Cout) Wt[((int64_t)k * Cout + co) * Cin + ci] = tile[tx][j];
    }
}

}  // namespace

DD3D_API int64_t dd3d_conv_bwd_weight_ws_bytes(int64_t M_out, int G, int K, int Cin, int Cout) {
    int64_t slots = Cin == 3 ? max_slots3(M_out, G) : max_slots128(M_out, G);
    int64_t nslab = (Cin == 3) ? slots : slots * K;
    int64_t per = (Cin == 3) ? (int64_t)K * 3 * Cout : (int64_t)Cin * Cout;
    const int64_t chunked = align_up64(nslab * per * 4, 256) + align_up64(nslab * 4, 256) + 256;  // slabs, their flags, the mask of offsets present anywhere
    if (Cin == 128 && G <= DD3D_BWW_MAX_GROUPS) {  // the stream-K launch (conv_bww_sk.hip); the larger of the two while both forms exist
        const int64_t sk = conv128_bww_sk_ws_bytes(M_out, G);
        return sk > chunked ? sk : chunked;
    }
    return chunked;
}

// Y[M_out,Cout] = sum_k X[nbr[k'][o]] . Wmat[k],  k' = mirror ? K-1-k : k.   Wmat is [K,Cin,Cout] row-major.
// m_dev (nullable): device-side row count; the launch then covers the bound M_out and rows >= *m_dev are not touched.
DD3D_API int dd3d_conv_fwd(const float* X, int64_t M_in, int Cin, const float* W, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                           const int32_t* m_dev, int mirror, float* Y, void* stream) {
    if (M_out < 0 || M_in < 0 || K != 27 || ld < M_out || Cout != 128 || (Cin != 3 && Cin != 128)) return DD3D_EARG;
    if (M_out >= (1ll << 31) / 128 * 16) return DD3D_ESIZE;
    if (M_out == 0) return 0;
    if (!X || !W || !nbr || !Y) return DD3D_EARG;
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 3) {
        if (mirror) return DD3D_EARG;
        const int64_t tiles = ceil_div64(M_out, 64);
        static const int cap = getenv("DD3D_CONV3_WGS") ? atoi(getenv("DD3D_CONV3_WGS")) : 512;
        k_conv3_fwd<<<dim3((unsigned)(tiles < cap ? tiles : cap)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, Y);
    } else {
        static const int dbg_tile = getenv("DD3D_CONV_TILE") ? atoi(getenv("DD3D_CONV_TILE")) : 0;
        static const int dbg_xcd = getenv("DD3D_CONV_XCD") ? atoi(getenv("DD3D_CONV_XCD")) : 1;  // XCD-aware tile order: same time, 10-50 % less L2 fill traffic (profiles/r01_pmc_conv128_rows.json)
        static const int dbg_rows = getenv("DD3D_CONV_ROWS") ? atoi(getenv("DD3D_CONV_ROWS")) : 1;
        const bool big = dbg_tile ? dbg_tile == 128 : M_out >= 128 * 512;
        if (big && dbg_rows) {
            const int64_t nt = ceil_div64(M_out, 128);
            if (mirror)
                k_conv128_rows<4, 1><<<dim3((unsigned)(ceil_div64(nt, 8) * 8)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, dbg_xcd, nullptr, nullptr,
                                                                                             nullptr, nullptr);
            else
                k_conv128_rows<4, 0><<<dim3((unsigned)(ceil_div64(nt, 8) * 8)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, dbg_xcd, nullptr, nullptr,
                                                                                             nullptr, nullptr);
        } else if (big) {
            const int64_t nt = ceil_div64(M_out, 128);
            k_conv128<128, 2, 2><<<dim3((unsigned)(ceil_div64(nt, 8) * 8)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, dbg_xcd);
        } else if (dbg_tile == 64) {  // measured (tools/convsweep.py): never the fastest -- 32-row tiles win below ~50k rows, 128-row tiles above
            const int64_t nt = ceil_div64(M_out, 64);
            k_conv128<64, 1, 4><<<dim3((unsigned)(ceil_div64(nt, 8) * 8)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, dbg_xcd);
        } else {
            // small launches (the synthetic batch): 32-row tiles so that every CU holds several workgroups whose load latencies overlap
            const int64_t nt = ceil_div64(M_out, 32);
            k_conv128<32, 1, 4><<<dim3((unsigned)(ceil_div64(nt, 8) * 8)), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, dbg_xcd);
        }
    }
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_conv_fwd for Cin = Cout = 128 along a schedule of dd3d_conv_schedule (same nbr / M_out / m_dev): rows in the order of `perm`,
// workgroup b on tile tile_map[b].  Same result as dd3d_conv_fwd, bit for bit; fewer all-zero rows through the matrix cores and an even
// load over the CUs.
DD3D_API int64_t dd3d_conv_schedule_blocks(int64_t M);
DD3D_API int dd3d_conv_schedule_tile_rows(int64_t M);
DD3D_API int dd3d_conv128_ordered(const float* X, int64_t M_in, const float* W, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* m_dev,
                                  int mirror, const int32_t* perm, const int32_t* tile_map, const int32_t* nbr_sorted, const uint32_t* tile_mask, float* Y,
                                  void* stream) {
    if (M_out < 0 || M_in < 0 || K != 27 || ld < M_out) return DD3D_EARG;
    if (M_out >= (1ll << 31) / 128 * 16) return DD3D_ESIZE;
    if (M_out == 0) return 0;
    if (!X || !W || !nbr || !Y || !perm || !tile_map || (nbr_sorted == nullptr) != (tile_mask == nullptr)) return DD3D_EARG;
    const unsigned nb = (unsigned)dd3d_conv_schedule_blocks(M_out);
    if (dd3d_conv_schedule_tile_rows(M_out) == 128 && mirror)
        k_conv128_rows<4, 1><<<dim3(nb), 256, 0, (hipStream_t)stream>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, 0, perm, tile_map, nbr_sorted, tile_mask);
    else if (dd3d_conv_schedule_tile_rows(M_out) == 128)
        k_conv128_rows<4, 0><<<dim3(nb), 256, 0, (hipStream_t)stream>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, 0, perm, tile_map, nbr_sorted, tile_mask);
    else  // small launches: 32-row tiles (see dd3d_conv_fwd), same schedule machinery
        k_conv128<32, 1, 4><<<dim3(nb), 256, 0, (hipStream_t)stream>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, 0, perm, tile_map);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dd3d_conv_fwd for Cin = Cout = 128 with the offsets of every tile dealt over up to 9 workgroups (launches of few rows, which would
// leave most CUs idle behind a serial chain per tile); ws holds the partial tiles.  Deterministic; the sum over the offsets is
// associated per split, so the result equals dd3d_conv_fwd's to fp32 rounding, not bit for bit.
DD3D_API int64_t dd3d_conv128_split_ws_bytes(int64_t M_out) { return 9 * (M_out > 0 ? M_out : 1) * 128 * (int64_t)sizeof(float); }
DD3D_API int dd3d_conv128_split(const float* X, int64_t M_in, const float* W, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* m_dev,
                                int mirror, float* Y, void* ws, int64_t ws_bytes, void* stream) {
    if (M_out < 0 || M_in < 0 || K != 27 || ld < M_out) return DD3D_EARG;
    if (M_out >= (1ll << 31) / 128 * 16 / 9) return DD3D_ESIZE;
    if (M_out == 0) return 0;
    if (!X || !W || !nbr || !Y) return DD3D_EARG;
    if (!ws || ws_bytes < dd3d_conv128_split_ws_bytes(M_out)) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nt = ceil_div64(M_out, 32);
    k_conv128<32, 1, 4, true><<<dim3((unsigned)nt, 9), 256, 0, st>>>(X, W, nbr, ld, K, (int)M_out, m_dev, mirror, Y, 0, nullptr, nullptr, (float*)ws, 9);
    k_split_sum<<<dim3((unsigned)ceil_div64(M_out, 8)), 256, 0, st>>>((const f32x4*)ws, (int)M_out, m_dev, 9, (f32x4*)Y);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dX[M,Cin] for a stride-1 convolution.  Cin==128: pass Wt = transposed kernel [K,Cout,Cin] (dd3d_transpose_kernel); Cin==3: pass W itself.
DD3D_API int dd3d_conv_bwd_data(const float* dY, int64_t M, const int32_t* m_dev, int Cout, const float* W_or_Wt, int Cin, const int32_t* nbr, int64_t ld,
                                int K, float* dX, void* stream) {
    if (Cin == 128) return dd3d_conv_fwd(dY, M, Cout, W_or_Wt, Cin, nbr, ld, K, M, m_dev, 1, dX, stream);
    if (M < 0 || K != 27 || ld < M || Cout != 128 || Cin != 3) return DD3D_EARG;
    if (M == 0) return 0;
    if (!dY || !W_or_Wt || !nbr || !dX) return DD3D_EARG;
    // every workgroup stages the 41-KB kernel in LDS first: rows per workgroup sized so that one round of workgroups (~768 slots) covers the launch
    int rpb = (int)((M / 768 + 7) / 8 * 8);
    rpb = rpb < 8 ? 8 : rpb > 64 ? 64 : rpb;
    k_conv3_bwd_data<<<dim3((unsigned)ceil_div64(M, rpb)), 256, 0, (hipStream_t)stream>>>(dY, W_or_Wt, nbr, ld, K, (int)M, m_dev, rpb, dX);
    DD3D_CHECK_LAUNCH();
    return 0;
}

// dW[G,K,Cin,Cout]; rows of group g are [group_row_off[g], group_row_off[g+1]) (device array, G+1 ints).  M_out may be a bound
// (>= group_row_off[G]): the row chunking follows the true row count, read on the device.
DD3D_API int dd3d_conv_bwd_weight(const float* X, int Cin, const float* dY, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                                  const int32_t* group_row_off, int G, float* dW, void* ws, int64_t ws_bytes, void* stream) {
    if (M_out < 0 || K != 27 || ld < M_out || Cout != 128 || (Cin != 3 && Cin != 128) || G <= 0 || !group_row_off || !dW) return DD3D_EARG;
    if (M_out >= (1ll << 31) / 128) return DD3D_ESIZE;
    if (ws_bytes < dd3d_conv_bwd_weight_ws_bytes(M_out, G, K, Cin, Cout) || !ws) return DD3D_EWS;
    hipStream_t st = (hipStream_t)stream;
    int64_t slots = Cin == 3 ? max_slots3(M_out, G) : max_slots128(M_out, G);
    int64_t nslab = (Cin == 3) ? slots : slots * K;
    int64_t per = (Cin == 3) ? (int64_t)K * 3 * Cout : (int64_t)Cin * Cout;
    float* slabs = (float*)ws;
    int* valid = (int*)((char*)ws + align_up64(nslab * per * 4, 256));
    unsigned* active = (unsigned*)((char*)valid + align_up64(nslab * 4, 256));
    if (M_out > 0 && (!X || !dY || !nbr)) return DD3D_EARG;
    static const int use_sk = getenv("DD3D_BWW_SK") ? atoi(getenv("DD3D_BWW_SK")) : 1;
    if (Cin == 128 && use_sk && G <= DD3D_BWW_MAX_GROUPS && M_out > 0) {
        // stream-K over the steps of all (group, offset) tiles: equal work per workgroup, accumulators resident across windows
        // persistent workgroups: two per CU; one per CU for the few groups of a shard (every workgroup leaves up to two partial slabs, and
        // with little work per workgroup their sum is what a launch waits for: measured 0.545 -> 0.533 ms per step with 6 groups, 0.436 ->
        // 0.429 with 4, 0.365 -> 0.359 with 2; from 10 groups on 512 is as fast or faster).  A function of the group count only, which the
        // captured and the eager launch share.
        static const int nwg_env = getenv("DD3D_BWW_NWG") ? atoi(getenv("DD3D_BWW_NWG")) : 0;
        const int nwg_pick = nwg_env > 0 ? nwg_env : (G <= 8 ? 256 : 512);
        const int nwg = nwg_pick < 8 ? 8 : nwg_pick > DD3D_BWW_MAX_NWG ? DD3D_BWW_MAX_NWG : nwg_pick / 8 * 8;
        return conv128_bww_sk(X, dY, nbr, ld, M_out, group_row_off, G, dW, ws, nwg, st);
    }
    if (M_out > 0) {
        if (Cin == 3)
            k_conv3_bww<<<dim3((unsigned)slots), 256, 0, st>>>(X, dY, nbr, ld, group_row_off, G, slabs, valid);
        else
            {
            DD3D_HIP(hipMemsetAsync(active, 0, 4, st));
            k_offset_mask<<<dim3((unsigned)(ceil_div64(M_out, 256) < 1024 ? ceil_div64(M_out, 256) : 1024)), 256, 0, st>>>(nbr, ld, K, group_row_off, G, active);  // one row per thread up to 262144 rows
            static const int order = getenv("DD3D_BWW_ORDER") ? atoi(getenv("DD3D_BWW_ORDER")) : 1;
            k_conv128_bww<<<dim3((unsigned)(ceil_div64(slots, 8) * 8), K), 256, 0, st>>>(X, dY, nbr, ld, K, group_row_off, G, slabs, valid, active, order);
        }
    }
    if (Cin == 3)
        k_slab_reduce<<<dim3((unsigned)ceil_div64(per / 4, 64), 1, G), 256, 0, st>>>((const f32x4*)slabs, valid, group_row_off, G, 1, 1, (int)(per / 4),
                                                                                     (f32x4*)dW, nullptr);
    else
        k_slab_reduce<<<dim3((unsigned)ceil_div64(per / 4, 64), K, G), 256, 0, st>>>((const f32x4*)slabs, valid, group_row_off, G, 0, K, (int)(per / 4),
                                                                                     (f32x4*)dW, M_out > 0 ? active : nullptr);
    DD3D_CHECK_LAUNCH();
    return 0;
}

DD3D_API int dd3d_transpose_kernel(const float* W, int K, int Cin, int Cout, float* Wt, void* stream) {
    if (K <= 0 || Cin <= 0 || Cout <= 0 || !W || !Wt) return DD3D_EARG;
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, K);
    k_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(W, K, Cin, Cout, Wt);
    DD3D_CHECK_LAUNCH();
    return 0;
}
