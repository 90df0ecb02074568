// knn.hip -- brute-force k-nearest-neighbour search for gfx950.
//
// Replaces get_neighbor_index / get_nearest_index (reference network/fs_net_repo/gcn3d.py:15-36).
// The (B,N,N) distance matrix of the reference is never written: distances live in registers
// (xyz path) or in f32-MFMA accumulators (feature path) and go straight into per-lane sorted lists.
//
// Bit-exactness contract (pinned by oracle/hsp_oracle.c against the reference's CPU path):
//   inner  = k-ordered fp32 fma chain from 0.  v_mfma_f32_32x32x2_f32 IS such a chain, so the
//            feature-space tiles reproduce torch.bmm bit for bit; the xyz path spells it out.
//   quad   = ATen row-sum order (quad_kernel below), dist = ((inner*-2)+quad[j])+quad[i]
//   select = ascending (distance, index); strict '<' keeps the lower index on exact ties.
#include "common.h"
#include "tie_pass.h"
#include <float.h>

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------------
// per-lane sorted list of the K1 smallest (distance, index) pairs seen so far
// ------------------------------------------------------------------------------------------------
template <int K1>
struct TopList {
    float d[K1];
    int i[K1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int p = 0; p < K1; ++p) { d[p] = INFINITY; i[p] = INT_MAX; }
    }
    // candidates arrive in ascending index order per lane, so strict '<' == lowest index first.
    // Sorted insert that drops the largest: new d[p] = clamp(v, d[p-1], d[p]) = v_med3_f32 (one op);
    // the index follows with one compare (shared between neighbouring slots) and two selects.
    __device__ __forceinline__ void insert(float v, int idx) {
        if (v < d[K1 - 1]) {
            bool lt_cur = true;                       // v < d[K1-1] holds inside the branch
#pragma unroll
            for (int p = K1 - 1; p > 0; --p) {
                const bool lt_prev = v < d[p - 1];
                i[p] = lt_prev ? i[p - 1] : (lt_cur ? idx : i[p]);
                d[p] = __builtin_amdgcn_fmed3f(d[p - 1], d[p], v);
                lt_cur = lt_prev;
            }
            i[0] = lt_cur ? idx : i[0];
            d[0] = fminf(d[0], v);
        }
    }
    // same update without the guard: a no-op for v = +inf (med3 keeps d[p], no compare passes), so a
    // wave can run it unconditionally with +inf in the lanes that have nothing to insert (no divergence).
    __device__ __forceinline__ void insert_always(float v, int idx) {
        bool lt_cur = v < d[K1 - 1];
#pragma unroll
        for (int p = K1 - 1; p > 0; --p) {
            const bool lt_prev = v < d[p - 1];
            i[p] = lt_prev ? i[p - 1] : (lt_cur ? idx : i[p]);
            d[p] = __builtin_amdgcn_fmed3f(d[p - 1], d[p], v);
            lt_cur = lt_prev;
        }
        i[0] = lt_cur ? idx : i[0];
        d[0] = fminf(d[0], v);
    }
    __device__ __forceinline__ void store(int2* dst) const {
#pragma unroll
        for (int p = 0; p < K1; ++p) dst[p] = make_int2(__float_as_int(d[p]), i[p]);
    }
};

// T-way tournament merge of sorted lists held in LDS.  The T lanes of one query are adjacent
// lanes of a wave; lane `sub` walks list `my_list`.  Writes ranks [drop, drop+k) of the union as
// indices to out_row, or (PAIRS) all k+drop ranks as (distance, index) pairs to out_pairs.
// A list slot that was never filled (non-finite distances are never inserted: `v < d` is false for NaN / +inf) still
// holds the sentinel index INT_MAX; it is replaced by `fallback` (a valid row, the query itself) before it is written,
// so a NaN / Inf input row can never turn into an out-of-range neighbour index downstream (the reference's topk also
// returns valid indices there; the NaN shows up in the loss and engine/train.py:91-95 skips the batch).
// ``msel`` > k + drop with ``tie_row``: the walk continues to rank msel - 1 (K1 >= msel) and bit 0 of *tie_row is set when two
// neighbouring ranks below msel hold EQUAL distances -- the only rows on which torch.topk's answer is not the
// (distance, index) order (csrc/knn_exact.hip); bit 1: the same among the first msel2 ranks (the window of a second, shorter list).
template <int K1, int T, bool PAIRS = false>
__device__ __forceinline__ void merge_write(const int2* __restrict__ lists, int my_list, int sub, int k,
                                            int drop, bool valid, int32_t* __restrict__ out_row,
                                            int nrows, int fallback, int2* __restrict__ out_pairs = nullptr,
                                            int msel = 0, uint8_t* __restrict__ tie_row = nullptr, int msel2 = 0,
                                            int32_t* __restrict__ out_row2 = nullptr, int k2 = 0) {
    const int2* mine = lists + (size_t)my_list * K1;
    int ptr = 0;
    int2 h = mine[0];
    float hd = __int_as_float(h.x);
    int hi = h.y;
    const int m = k + drop;
    const int mm = tie_row && msel > m ? msel : m;
    float prev = 0.f;
    bool tie = false, tie2 = false;                   // tie2: among the first msel2 ranks (the short list's window)
    for (int r = 0; r < mm; ++r) {
        float bd = hd;
        int bi = hi;
#pragma unroll
        for (int s = 1; s < T; s <<= 1) {
            const float od = __shfl_xor(bd, s, T);
            const int oi = __shfl_xor(bi, s, T);
            const bool take = (od < bd) || (od == bd && oi < bi);
            bd = take ? od : bd;
            bi = take ? oi : bi;
        }
        if (hi == bi) {  // this lane owned the winner: advance its head
            ++ptr;
            if (ptr < K1) {
                h = mine[ptr];
                hd = __int_as_float(h.x);
                hi = h.y;
            } else {
                hd = INFINITY;
                hi = INT_MAX;
            }
        }
        const bool eq = r > 0 && bd == prev;
        tie = tie || eq;
        tie2 = tie2 || (eq && r < msel2);
        prev = bd;
        if (PAIRS) {
            if (sub == 0) out_pairs[r] = make_int2(__float_as_int(bd), bi);
        } else if (valid && sub == 0 && r >= drop && r < m) {
            const int o = (unsigned)bi < (unsigned)nrows ? bi : fallback;
            out_row[r - drop] = o;
            if (out_row2 && r - drop < k2) out_row2[r - drop] = o;          // the short list: the prefix (final unless flagged)
        }
    }
    if (tie_row && valid && sub == 0) *tie_row = (tie ? 1 : 0) | (tie2 ? 2 : 0);
}

// ------------------------------------------------------------------------------------------------
// xyz path (C == 3): the cloud (x,y,z,|p|^2) sits in LDS; T lanes share one query, each scanning
// every T-th candidate (one broadcast ds_read_b128 per candidate); lists are merged by tournament.
// grid (ceil(N / (256/T)), B), block 256, dynamic LDS = max(chunk*16, 256*K1*8)
// ------------------------------------------------------------------------------------------------
template <int K1, int T>
__global__ __launch_bounds__(256) void knn3_kernel(const float* __restrict__ x, int N, int k, int drop,
                                                   int32_t* __restrict__ idx, int chunk, int msel,
                                                   uint8_t* __restrict__ tie, int msel2, int32_t* __restrict__ idx2, int k2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* pts = reinterpret_cast<float4*>(smem);
    int2* lists = reinterpret_cast<int2*>(smem);          // aliases pts once the scan is over
    constexpr int Q = 256 / T;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int q = blockIdx.x * Q + tid / T;
    const int t = tid % T;
    const float* xb = x + (size_t)b * N * 3;
    const bool valid = q < N;

    TopList<K1> top;
    top.init();
    float qx = 0.f, qy = 0.f, qz = 0.f, qq = 0.f;
    if (valid) {
        qx = xb[q * 3 + 0]; qy = xb[q * 3 + 1]; qz = xb[q * 3 + 2];
        qq = quad3(qx, qy, qz);
    }
    for (int c0 = 0; c0 < N; c0 += chunk) {
        const int cn = min(chunk, N - c0);
        __syncthreads();
        for (int j = tid; j < cn; j += 256) {
            const float px = xb[(c0 + j) * 3 + 0], py = xb[(c0 + j) * 3 + 1], pz = xb[(c0 + j) * 3 + 2];
            pts[j] = make_float4(px, py, pz, quad3(px, py, pz));
        }
        __syncthreads();
        if (valid) {
            for (int j = t; j < cn; j += T) {
                const float4 c = pts[j];
                const float inner = dot3_chain(qx, qy, qz, c.x, c.y, c.z);
                const float d = add_rn(add_rn(mul_rn(inner, -2.0f), c.w), qq);
                top.insert(d, c0 + j);
            }
        }
    }
    __syncthreads();                       // every lane is done with pts: reuse the LDS for the lists
    top.store(lists + (size_t)tid * K1);   // each lane re-reads only its own list: no further barrier
    merge_write<K1, T>(lists, tid, t, k, drop, valid, idx + ((size_t)b * N + (valid ? q : 0)) * k, N, valid ? q : 0, nullptr,
                       msel, tie ? tie + (size_t)b * N + (valid ? q : 0) : nullptr, msel2,
                       idx2 ? idx2 + ((size_t)b * N + (valid ? q : 0)) * k2 : nullptr, k2);
}

#define KNN_TAU_LOW_BIT 18     // radix select of the pruning bound: key bits 31..18 (sign, exponent, 5 mantissa bits)

__device__ __forceinline__ unsigned sortable_key(float f) {
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// ------------------------------------------------------------------------------------------------
// xyz path, one WAVE per query (64 <= N <= 64*S): lane l owns candidates l, l+64, ... and keeps their S
// distances in registers; the k + drop nearest are then selected exactly as in knn_select_wave (a bound
// from the 64 lane minima by radix select over ballots, compaction of the ~k+4 survivors, ranking by
// (distance, index)) -- about 600 wave instructions per query, where 16 lanes x 64 sorted-list inserts
// cost ~2000.  Each wave takes QW consecutive queries.
// grid (ceil(N / (4*QW)), B), block 256, LDS = N*16 (cloud) + 4 * KNN3W_CAP*8 (survivor scratch per wave)
// ------------------------------------------------------------------------------------------------
#define KNN3W_QW 4
#define KNN3W_CAP 128   // survivor scratch entries per wave (typically ~k + 4 are used)
#define KNN3W_SV (KNN3W_CAP + 4)   // ... + four sentinel entries behind the last survivor (the ranking loop reads four at a time)

// ``tie`` (may be null) with ``msel`` = k + drop + 1: the selection runs one rank past the answer and tie[row] says whether two of
// those msel nearest hold EQUAL distances (see merge_write).
// The body serves two launches: knn3_wave_kernel (a whole (B,N,3) tensor) and geometry_levels_kernel (the coarse levels of the
// stack, whose points are ROWS sel[j] of the finer cloud: ``sel`` / ``sel_outer`` / ``vout``).  xb: the cloud the rows come from; row0: this cloud's
// first row in idx / idx2 / tie; qblock: which 4 * QW queries of the cloud this workgroup takes.
template <int S>
__device__ __forceinline__ void knn3_wave_body(char* smem, const float* __restrict__ xb, const int32_t* __restrict__ sel,
                                               const int32_t* __restrict__ sel_outer, float* __restrict__ vout, int N, int k, int drop, int32_t* __restrict__ idx,
                                               int msel, uint8_t* __restrict__ tie, int msel2, int32_t* __restrict__ idx2, int k2,
                                               int tie_inline, size_t row0, int qblock, int QW) {
    float4* pts = reinterpret_cast<float4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int2* sv = reinterpret_cast<int2*>(pts + N) + (size_t)wave * KNN3W_SV;
    float* sd = reinterpret_cast<float*>(reinterpret_cast<int2*>(pts + N) + 4 * KNN3W_SV) + wave * 64;   // ranked distances (tie flags)
    // tie_inline: a row whose flags come out non-zero is replayed HERE through libstdc++'s routines (tie_pass.h) from the distances
    // this wave already holds in registers, instead of being left to a second launch (knn_xyz_ties_kernel: ~5 us per call even when
    // no row is flagged).  A scratch row per wave, compiled in for the small clouds only (S <= 9: N <= 576): the replay costs
    // the kernel ~50 VGPRs (1.24 -> 2.15 ms at B = 64, N = 4096 when every variant carried it), and at N = 1028 it bought nothing
    // -- one scratch row per workgroup under an LDS lock ran the bench cloud's 4 flagged rows no sooner than the second launch
    // does (54.7 vs 33.8 + 20.6 us) and a tiled cloud 3x slower (994 vs 335 us: the workgroup's waves queue on the lock).
    char* tie_mem = reinterpret_cast<char*>(reinterpret_cast<float*>(reinterpret_cast<int2*>(pts + N) + 4 * KNN3W_SV) + 4 * 64);
    TkE* tq = reinterpret_cast<TkE*>(tie_mem + (size_t)wave * 16 * N);
    for (int j = tid; j < N; j += 256) {
        int r = sel ? sel[j] : j;                             // row j of this level = row sel[j] of the level it was drawn from,
        r = sel_outer ? sel_outer[r] : r;                     // which is row sel_outer[.] of the cloud xb
        const float px = xb[r * 3 + 0], py = xb[r * 3 + 1], pz = xb[r * 3 + 2];
        pts[j] = make_float4(px, py, pz, quad3(px, py, pz));
        if (vout && qblock == 0) { vout[j * 3] = px; vout[j * 3 + 1] = py; vout[j * 3 + 2] = pz; }   // the level's vertices
    }
    __syncthreads();
    const int m = k + drop;
    const int ms = tie && msel > m ? msel : m;                       // ranks looked at
    for (int qi = 0; qi < QW; ++qi) {
        const int q = (qblock * 4 + wave) * QW + qi;                 // wave-uniform
        if (q >= N) break;
        const float4 qp = pts[q];
        float d[S];
        float lmin = INFINITY;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int j = lane + 64 * s;
            const float4 c = pts[j < N ? j : N - 1];
            const float inner = dot3_chain(qp.x, qp.y, qp.z, c.x, c.y, c.z);
            const float dv = add_rn(add_rn(mul_rn(inner, -2.0f), c.w), qp.w);
            d[s] = j < N ? fminf(dv, FLT_MAX) : INFINITY;       // NaN / +inf of a real row -> FLT_MAX: still selectable, so
            lmin = fminf(lmin, d[s]);                           // every output slot is written with a valid index
        }
        // the ms-th smallest of the 64 lane minima: ms distinct candidates are <= tau
        const unsigned key = sortable_key(lmin);
        unsigned prefix = 0;
        int need = ms;
        // (the bound only has to be >= the ms-th smallest minimum: the descent stops after the sign, the exponent and five mantissa
        // bits and fills the rest with ones -- tau up to 3 % high, a survivor or two more for the ranking, 18 ballot rounds fewer)
        for (int bit = 31; bit >= KNN_TAU_LOW_BIT; --bit) {
            const bool zero = (key ^ prefix) < (1u << bit);       // bits 31..bit+1 equal the prefix (whose lower bits are 0), bit `bit` is 0
            const int c0 = __popcll(__ballot(zero));
            if (need > c0) { need -= c0; prefix |= 1u << bit; }
        }
        prefix |= (1u << KNN_TAU_LOW_BIT) - 1u;
        const float tau = __uint_as_float(prefix ^ ((prefix >> 31) ? 0x80000000u : 0xffffffffu));
        int n = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool keep = d[s] <= tau;                    // +inf padding never passes a finite tau
            const unsigned long long bal = __ballot(keep);
            if (bal == 0ull) continue;                        // (wave-uniform: most slots of a dense cloud hold no survivor)
            const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && pos < KNN3W_CAP) sv[pos] = make_int2(__float_as_int(d[s]), lane + 64 * s);
            n += __popcll(bal);
        }
        if (lane < 4 && n <= KNN3W_CAP) sv[n + lane] = make_int2(__float_as_int(INFINITY), INT_MAX);   // sentinels: rank nothing
        __builtin_amdgcn_wave_barrier();
        int32_t* out = idx + (row0 + q) * k;
        int32_t* out2 = idx2 ? idx2 + (row0 + q) * k2 : nullptr;   // the short list: the prefix (final unless flagged)
        bool tied = false, tied2 = false;                     // tied2: inside the first msel2 ranks
        if (n <= KNN3W_CAP) {
            for (int e = lane; e < n; e += 64) {
                const int2 me = sv[e];
                const float de = __int_as_float(me.x);
                int rank = 0;
                for (int f = 0; f < n; f += 4) {               // four entries per round trip (the loop waits on LDS, not on the ALU)
                    const int4 o01 = *reinterpret_cast<const int4*>(sv + f), o23 = *reinterpret_cast<const int4*>(sv + f + 2);
                    const float d0 = __int_as_float(o01.x), d1 = __int_as_float(o01.z), d2 = __int_as_float(o23.x), d3 = __int_as_float(o23.z);
                    rank += (d0 < de || (d0 == de && o01.y < me.y)) ? 1 : 0;
                    rank += (d1 < de || (d1 == de && o01.w < me.y)) ? 1 : 0;
                    rank += (d2 < de || (d2 == de && o23.y < me.y)) ? 1 : 0;
                    rank += (d3 < de || (d3 == de && o23.w < me.y)) ? 1 : 0;
                }
                if (rank >= drop && rank < m) out[rank - drop] = me.y;
                if (out2 && rank >= drop && rank - drop < k2) out2[rank - drop] = me.y;
                if (tie && rank < ms) sd[rank] = de;           // the ms nearest distances in rank order (ranks are a permutation)
            }
            if (tie) {
                // equal distances occupy neighbouring ranks: one compare per rank instead of two more tests per ranked pair
                __builtin_amdgcn_wave_barrier();
                const bool eq = lane + 1 < ms && sd[lane] == sd[lane + 1];
                tied = eq;
                tied2 = eq && lane + 1 < msel2;
            }
        } else {
            // more survivors than scratch (heavily duplicated points): extract the m smallest (distance, index)
            // pairs one at a time -- lane-local minimum above the previous pick, then a bitwise descent over
            // ballots for the wave minimum.  Slow (~150 ballots per pick) but exact and scratch-free.
            unsigned pk = 0;                                  // previous pick: sortable distance key, index
            int pj = -1;
            for (int r = 0; r < ms; ++r) {
                unsigned bk = 0xffffffffu;
                int bj = 0x7fffffff;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const unsigned ks = sortable_key(d[s]);
                    const int j = lane + 64 * s;
                    const bool after = r == 0 || ks > pk || (ks == pk && j > pj);
                    const bool better = ks < bk || (ks == bk && j < bj);
                    if (j < N && after && better) { bk = ks; bj = j; }
                }
                unsigned long long live = __ballot(bj != 0x7fffffff);
                unsigned wk = 0;
                for (int bit = 31; bit >= 0; --bit) {         // smallest key among the live lanes
                    const unsigned long long z = __ballot(((bk >> bit) & 1u) == 0) & live;
                    if (z) live = z; else wk |= 1u << bit;
                }
                int wj = 0;
                for (int bit = 15; bit >= 0; --bit) {         // smallest index among the lanes holding that key
                    const unsigned long long z = __ballot(((bj >> bit) & 1) == 0) & live;
                    if (z) live = z; else wj |= 1 << bit;
                }
                if (lane == 0 && r >= drop && r < m) out[r - drop] = wj;
                if (lane == 0 && out2 && r >= drop && r - drop < k2) out2[r - drop] = wj;
                tied = tied || (r > 0 && wk == pk);
                tied2 = tied2 || (r > 0 && r < msel2 && wk == pk);
                pk = wk; pj = wj;
            }
        }
        if (tie) {
            const int flags = (__ballot(tied) != 0ull ? 1 : 0) | (__ballot(tied2) != 0ull ? 2 : 0);
            if (!tie_inline) {
                if (lane == 0) tie[row0 + q] = (uint8_t)flags;
            } else if (S <= 9 && flags) {                     // (wave-uniform)
                int* LA = reinterpret_cast<int*>(tq + N);
                int* LB = LA + N;
                // (as knn_xyz_ties_kernel: the list whose search leaves the row untouched -- partial_sort -- goes first)
                const int m2 = k2 + drop;
                const bool want2 = out2 && (flags & 2);
                const bool second_first = want2 && !tkw_topk_destroys(m2, N) && tkw_topk_destroys(m, N);
                bool filled = false;
                for (int pass = 0; pass < (want2 ? 2 : 1); ++pass) {
                    const bool short_list = (pass == 0) == second_first && want2;
                    if (!filled) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const int j = lane + 64 * s;
                            TkE e;
                            e.v = d[s]; e.i = j;
                            if (j < N) tq[j] = e;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    const int mm = short_list ? m2 : m;
                    const LaneAcc H = tkw_topk(tq, LA, LB, mm, N, lane);
                    filled = !tkw_topk_destroys(mm, N);
                    int32_t* o = short_list ? out2 : out;
                    if (lane >= drop && lane < mm) o[lane - drop] = H.i;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                      // sv is reused by the next query
    }
}

template <int S>
__global__ __launch_bounds__(256) void knn3_wave_kernel(const float* __restrict__ x, int N, int k, int drop,
                                                        int32_t* __restrict__ idx, int msel, uint8_t* __restrict__ tie,
                                                        int msel2, int32_t* __restrict__ idx2, int k2, int tie_inline, int QW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    knn3_wave_body<S>(smem, x + (size_t)b * N * 3, nullptr, nullptr, nullptr, N, k, drop, idx, msel, tie, msel2, idx2, k2, tie_inline,
                      (size_t)b * N, (int)blockIdx.x, QW);
}

// ------------------------------------------------------------------------------------------------
// quad[r] = sum_c x[r][c]^2 in ATen's CPU order (oracle/hsp_oracle.c aten_row_sum): 8-lane vector
// partials with 4-way ILP, then a sequential horizontal sum.  One thread per row.
// C < 8 : 4 interleaved scalar partials.  Cascade levels (only reached when C >= 512) included.
// ------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void multi_row_sum(const float* __restrict__ row, long long size, float (&out)[4][W]) {
    // size < 2^20 => level_step == 16
    int lp = 0;
    {
        long long v = 1; int r = 0;
        while (v < size) { v <<= 1; ++r; }
        lp = r / 4;
        if (lp < 4) lp = 4;
    }
    const long long level_step = 1ll << lp, level_mask = level_step - 1;
    float acc[4][4][W];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < W; ++l) acc[a][k][l] = 0.f;
    long long i = 0;
    for (; i + level_step <= size;) {
        for (long long j = 0; j < level_step; ++j, ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int l = 0; l < W; ++l) {
                    const float v = row[(i * 4 + k) * W + l];
                    acc[0][k][l] = add_rn(acc[0][k][l], mul_rn(v, v));
                }
#pragma unroll
        for (int j = 1; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int l = 0; l < W; ++l) { acc[j][k][l] = add_rn(acc[j][k][l], acc[j - 1][k][l]); acc[j - 1][k][l] = 0.f; }
            const long long mask = level_mask << (j * lp);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < W; ++l) {
                const float v = row[(i * 4 + k) * W + l];
                acc[0][k][l] = add_rn(acc[0][k][l], mul_rn(v, v));
            }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < W; ++l) acc[0][k][l] = add_rn(acc[0][k][l], acc[j][k][l]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < W; ++l) out[k][l] = acc[0][k][l];
}

// the same sum for 8 <= C < 512, C % 8 == 0 (every feature width of the stack), with 32 lanes per row: lane
// 8k + l IS ATen's partial accumulator (ILP slot k, vector lane l) -- it adds elements 32 i + 8k + l in ascending i,
// so a row is read with fully coalesced 128-byte requests instead of one strided row per thread -- then the same
// fixed-order combination: leftover 8-wide chunks into slot 0, slots 1..3 into slot 0, vector lanes 0..7 in order.
__global__ __launch_bounds__(256) void quad32_kernel(const float* __restrict__ x, long long rows, int C,
                                                     float* __restrict__ quad) {
    const int lane = threadIdx.x & 63, sub = lane & 31, base = lane & 32;
    const long long r = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool live = r < rows;
    const float* row = x + (live ? r : 0) * C;
    const int vec_size = C >> 3, size_ilp = vec_size >> 2;
    float a = 0.f;
    for (int i = 0; i < size_ilp; ++i) { const float v = row[i * 32 + sub]; a = add_rn(a, mul_rn(v, v)); }
    if (sub < 8)
        for (int m = size_ilp * 4; m < vec_size; ++m) { const float v = row[m * 8 + sub]; a = add_rn(a, mul_rn(v, v)); }
    // slots k = 1, 2, 3 into slot 0 (lanes 0..7 of the half-wave)
    float t = a;
    t = add_rn(t, __shfl(a, base | ((sub + 8) & 31)));
    t = add_rn(t, __shfl(a, base | ((sub + 16) & 31)));
    t = add_rn(t, __shfl(a, base | ((sub + 24) & 31)));
    float fin = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) fin = add_rn(fin, __shfl(t, base | l));
    if (live && sub == 0) quad[r] = fin;
}

__global__ __launch_bounds__(256) void quad_kernel(const float* __restrict__ x, long long rows, int C,
                                                   float* __restrict__ quad) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* row = x + r * C;
    if (C < 8) {
        float ps[4][1];
        const long long size_ilp = C / 4;
        multi_row_sum<1>(row, size_ilp, ps);
        for (long long i = size_ilp * 4; i < C; ++i) ps[0][0] = add_rn(ps[0][0], mul_rn(row[i], row[i]));
        for (int k = 1; k < 4; ++k) ps[0][0] = add_rn(ps[0][0], ps[k][0]);
        quad[r] = ps[0][0];
        return;
    }
    constexpr int W = 8;
    float ps[4][W];
    const long long vec_size = C / W, size_ilp = vec_size / 4;
    multi_row_sum<W>(row, size_ilp, ps);
    for (long long m = size_ilp * 4; m < vec_size; ++m)
#pragma unroll
        for (int l = 0; l < W; ++l) { const float v = row[m * W + l]; ps[0][l] = add_rn(ps[0][l], mul_rn(v, v)); }
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < W; ++l) ps[0][l] = add_rn(ps[0][l], ps[k][l]);
    float fin = 0.f;
    for (long long c = vec_size * W; c < C; ++c) fin = add_rn(fin, mul_rn(row[c], row[c]));
#pragma unroll
    for (int l = 0; l < W; ++l) fin = add_rn(fin, ps[0][l]);
    quad[r] = fin;
}

// ------------------------------------------------------------------------------------------------
// exact selection of the m = k + drop nearest out of N distances held in LDS, by ONE wave:
//   1. tau = the m-th smallest of the 64 lane-local minima (radix select over ballots): m distinct
//      candidates are <= tau, so everything above it is out;
//   2. the survivors (about m + 3) are compacted in index order and ranked by (distance, index) against
//      each other; ranks [drop, m) are the answer, already in order.
// dl: N floats (N >= 64), sv: N int2 of scratch, both LDS
// ------------------------------------------------------------------------------------------------

// tie (may be null): *tie = 1 when two of the m + 1 nearest hold equal distances (the selection then runs to rank m), else 0
__device__ __forceinline__ void knn_select_wave(const float* dl, int2* sv, int N, int k, int drop,
                                                int32_t* __restrict__ out, uint8_t* __restrict__ tie = nullptr) {
    const int lane = threadIdx.x & 63;
    const int m = k + drop;
    const int ms = tie && m + 1 <= N ? m + 1 : m;
    float lmin = INFINITY;
    // real rows with a NaN / +inf distance count as FLT_MAX (ties by index): m valid candidates always exist, every
    // output slot is written (see merge_write)
    for (int j = lane; j < N; j += 64) lmin = fminf(lmin, fminf(dl[j], FLT_MAX));
    // radix select, most significant bit first: the m-th smallest of the 64 keys
    const unsigned key = sortable_key(lmin);
    unsigned prefix = 0;
    int need = ms;
    for (int bit = 31; bit >= KNN_TAU_LOW_BIT; --bit) {     // (a bound, not the exact value: see knn3_wave_body)
        const bool zero = (key ^ prefix) < (1u << bit);
        const int c0 = __popcll(__ballot(zero));
        if (need > c0) { need -= c0; prefix |= 1u << bit; }
    }
    prefix |= (1u << KNN_TAU_LOW_BIT) - 1u;
    const float tau = __uint_as_float(prefix ^ ((prefix >> 31) ? 0x80000000u : 0xffffffffu));   // key -> float
    int n = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const float d = j < N ? fminf(dl[j], FLT_MAX) : INFINITY;
        const bool keep = j < N && d <= tau;
        const unsigned long long bal = __ballot(keep);
        if (keep) sv[n + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(__float_as_int(d), j);
        n += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();
    bool tied = false;
    for (int e = lane; e < n; e += 64) {
        const int2 me = sv[e];
        const float de = __int_as_float(me.x);
        int rank = 0;
        bool eq_lo = false, eq_hi = false;                    // an equal distance at a lower / a higher index
        for (int f = 0; f < n; ++f) {
            const int2 o = sv[f];
            const float df = __int_as_float(o.x);
            const bool eq = df == de;
            rank += (df < de || (eq && o.y < me.y)) ? 1 : 0;
            eq_lo = eq_lo || (eq && o.y < me.y);
            eq_hi = eq_hi || (eq && o.y > me.y);
        }
        if (rank >= drop && rank < m) out[rank - drop] = me.y;
        // (equal distances occupy neighbouring ranks: the pair lies below ms when this entry does and its partner is the previous
        // rank, or the next one and that is still below ms)
        tied = tied || (rank < ms && (eq_lo || (eq_hi && rank + 1 < ms)));
    }
    if (tie) {
        const bool any_tied = __ballot(tied) != 0ull;
        if (lane == 0) *tie = any_tied ? 1 : 0;
    }
}

// The same selection with the candidates' distances in REGISTERS (S per lane, N <= 64 S) and the survivors ranked lane-against-lane:
// knn_select_wave walks the survivor list in LDS once per survivor (a dependent ds_read per step: ~100 clocks x ~28 survivors, and
// ~10 000 clocks a query measured with nine waves selecting side by side); here lane e holds survivor e and the list is broadcast
// from registers (v_readlane_b32 with a scalar index: no memory round trip), one 64-bit compare per pair.  The radix descent is a
// chain of vector-compare -> scalar-count -> scalar-select steps that waits on itself, so a wave runs NQ (1 or 2) queries through it
// side by side: two independent chains interleave and take about the time of one.  Same bound, same (distance, index) ranks,
// same tie flag as knn_select_wave; more than 64 survivors (heavily duplicated rows) fall back on it.
// dl[u], out[u], tie[u]: query u's distances / output row / flag byte; sv: max(64 * NQ, N) int2 of scratch.
template <int S, int NQ, bool TIE>
__device__ __forceinline__ void knn_select_wave_regs(const float* const (&dl)[NQ], int2* sv, int N, int k, int drop,
                                                     int32_t* const (&out)[NQ], uint8_t* const (&tie)[NQ]) {
    const int lane = threadIdx.x & 63;
    const int m = k + drop;
    const int ms = TIE && m + 1 <= N ? m + 1 : m;
    float v[NQ][S];
    unsigned key[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        float lmin = INFINITY;
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int j = lane + 64 * s_;
            const float raw = dl[u][j < N ? j : N - 1];               // (clamped address: all S loads of both queries in flight together)
            // NaN / +inf of a real row -> FLT_MAX (see knn_select_wave); -0.0 -> +0.0 (equal as floats, not as key bits)
            v[u][s_] = j < N ? add_rn(fminf(raw, FLT_MAX), 0.0f) : INFINITY;
            lmin = fminf(lmin, v[u][s_]);
        }
        key[u] = sortable_key(lmin);
    }
    unsigned prefix[NQ];
    int need[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) { prefix[u] = 0; need[u] = ms; }
    for (int bit = 31; bit >= KNN_TAU_LOW_BIT; --bit) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const bool zero = (key[u] ^ prefix[u]) < (1u << bit);    // bits 31..bit+1 equal the prefix and bit `bit` is 0
            const int c0 = __popcll(__ballot(zero));
            if (need[u] > c0) { need[u] -= c0; prefix[u] |= 1u << bit; }
        }
    }
    int n[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const unsigned pf = prefix[u] | ((1u << KNN_TAU_LOW_BIT) - 1u);
        const float tau = __uint_as_float(pf ^ ((pf >> 31) ? 0x80000000u : 0xffffffffu));
        n[u] = 0;
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const bool keep = v[u][s_] <= tau;                    // +inf padding never passes a finite tau
            const unsigned long long bal = __ballot(keep);
            const int pos = n[u] + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && pos < 64) sv[u * 64 + pos] = make_int2(__float_as_int(v[u][s_]), lane + 64 * s_);
            n[u] += __popcll(bal);
        }
    }
    __builtin_amdgcn_wave_barrier();
    int2 me[NQ];
    unsigned khi[NQ];
    unsigned long long ke[NQ];
    int rank[NQ];
    bool eq_lo[NQ], eq_hi[NQ];
    int nmax = 0;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int nu = n[u] <= 64 ? n[u] : 0;                      // (an overflowing query is redone below)
        me[u] = lane < nu ? sv[u * 64 + lane] : make_int2(__float_as_int(INFINITY), INT_MAX);
        khi[u] = sortable_key(__int_as_float(me[u].x));
        ke[u] = ((unsigned long long)khi[u] << 32) | (unsigned)me[u].y;
        rank[u] = 0; eq_lo[u] = false; eq_hi[u] = false;
        nmax = nu > nmax ? nu : nmax;
    }
    // ranks by ONE 64-bit compare per pair: key = (sortable distance bits, index).  Lanes past a query's survivors hold the largest
    // key, so running both queries to the longer list changes no rank.
    // (four list entries per trip: the list is padded with largest keys up to lane 63, which rank nothing)
    for (int f0 = 0; f0 < nmax; f0 += 4) {
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const int f = f0 + df;
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const unsigned fh = (unsigned)__builtin_amdgcn_readlane((int)khi[u], f);
                const unsigned fl = (unsigned)__builtin_amdgcn_readlane(me[u].y, f);
                const unsigned long long kf = ((unsigned long long)fh << 32) | fl;
                rank[u] += kf < ke[u] ? 1 : 0;
                if (TIE) {
                    eq_lo[u] = eq_lo[u] || (fh == khi[u] && fl < (unsigned)me[u].y);
                    eq_hi[u] = eq_hi[u] || (fh == khi[u] && fl > (unsigned)me[u].y);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (n[u] > 64) {                                           // (wave-uniform)
            __builtin_amdgcn_wave_barrier();
            knn_select_wave(dl[u], sv, N, k, drop, out[u], TIE ? tie[u] : nullptr);   // (the survivor slots are dead: every lane holds its entry)
            continue;
        }
        if (lane < n[u] && rank[u] >= drop && rank[u] < m) out[u][rank[u] - drop] = me[u].y;
        if (TIE) {
            const bool tied = lane < n[u] && rank[u] < ms && (eq_lo[u] || (eq_hi[u] && rank[u] + 1 < ms));
            const bool any_tied = __ballot(tied) != 0ull;
            if (lane == 0) *tie[u] = any_tied ? 1 : 0;
        }
    }
}

// one query, N <= 1088 in registers (17 per lane), else the LDS walk
__device__ __forceinline__ void knn_select_wave_any(const float* dl, int2* sv, int N, int k, int drop, int32_t* __restrict__ out,
                                                    uint8_t* __restrict__ tie) {
    if (N > 64 * 17) { knn_select_wave(dl, sv, N, k, drop, out, tie); return; }
    const float* const dls[1] = {dl};
    int32_t* const outs[1] = {out};
    uint8_t* const ties[1] = {tie};
    if (N <= 64 * 5) {
        if (tie) knn_select_wave_regs<5, 1, true>(dls, sv, N, k, drop, outs, ties);
        else knn_select_wave_regs<5, 1, false>(dls, sv, N, k, drop, outs, ties);
    } else {
        if (tie) knn_select_wave_regs<17, 1, true>(dls, sv, N, k, drop, outs, ties);
        else knn_select_wave_regs<17, 1, false>(dls, sv, N, k, drop, outs, ties);
    }
}

// ------------------------------------------------------------------------------------------------
// remainder queries of the feature path.  N = 1028 = 32*32 + 4 leaves 4 queries per cloud that would
// cost a whole MFMA workgroup (33 instead of 32 tiles: 528 workgroups on 256 CUs, a 3:2 imbalance).
// They are handled here: one workgroup per query; thread t owns candidates t, t+256, ... and runs four
// of their k-ordered fma chains at a time (same chain as the MFMA path: bit-identical distances) into
// an LDS array; wave 0 then selects (knn_select_wave).
// Runs as extra workgroups of knn_feat_kernel's grid (the lowest block ids), concurrently with the MFMA tiles.
// LDS = C*4 + 12*(N+3)
// ------------------------------------------------------------------------------------------------
template <int K1>
__device__ __forceinline__ void knn_feat_tail_body(char* smem, const float* __restrict__ x,
                                                   const float* __restrict__ quad, int N, int C, int k, int drop,
                                                   int q, int32_t* __restrict__ idx, uint8_t* __restrict__ tie,
                                                   float* __restrict__ dmat) {
    const int Np = (N + 3) & ~3;                              // keeps sv / sq 16-byte aligned
    float* dl = reinterpret_cast<float*>(smem);               // N distances
    int2* sv = reinterpret_cast<int2*>(dl + Np);              // selection scratch
    float* sq = reinterpret_cast<float*>(sv + Np);            // the query row
    const int tid = threadIdx.x;
    const int b = blockIdx.y;                                 // q < N by construction
    const float* xb = x + (size_t)b * N * C;
    const float* quadb = quad + (size_t)b * N;
    for (int e = tid; e < C; e += 256) sq[e] = xb[(size_t)q * C + e];
    __syncthreads();
    const float qn = quadb[q];
    for (int j0 = tid; j0 < N; j0 += 4 * 256) {
        // four candidates per pass (clamped rows: a clamped duplicate is discarded below), independent chains
        const int j1 = j0 + 256, j2 = j0 + 512, j3 = j0 + 768;
        const float* r0 = xb + (size_t)j0 * C;
        const float* r1 = xb + (size_t)min(j1, N - 1) * C;
        const float* r2 = xb + (size_t)min(j2, N - 1) * C;
        const float* r3 = xb + (size_t)min(j3, N - 1) * C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int c = 0;
        for (; c + 3 < C; c += 4) {                         // 16-byte row segments (C % 4 == 0 keeps them aligned)
            const float4 qv = *reinterpret_cast<const float4*>(sq + c);
            const float4 v0 = *reinterpret_cast<const float4*>(r0 + c);
            const float4 v1 = *reinterpret_cast<const float4*>(r1 + c);
            const float4 v2 = *reinterpret_cast<const float4*>(r2 + c);
            const float4 v3 = *reinterpret_cast<const float4*>(r3 + c);
            a0 = __fmaf_rn(qv.w, v0.w, __fmaf_rn(qv.z, v0.z, __fmaf_rn(qv.y, v0.y, __fmaf_rn(qv.x, v0.x, a0))));
            a1 = __fmaf_rn(qv.w, v1.w, __fmaf_rn(qv.z, v1.z, __fmaf_rn(qv.y, v1.y, __fmaf_rn(qv.x, v1.x, a1))));
            a2 = __fmaf_rn(qv.w, v2.w, __fmaf_rn(qv.z, v2.z, __fmaf_rn(qv.y, v2.y, __fmaf_rn(qv.x, v2.x, a2))));
            a3 = __fmaf_rn(qv.w, v3.w, __fmaf_rn(qv.z, v3.z, __fmaf_rn(qv.y, v3.y, __fmaf_rn(qv.x, v3.x, a3))));
        }
        for (; c < C; ++c) {
            const float qc = sq[c];
            a0 = __fmaf_rn(qc, r0[c], a0);
            a1 = __fmaf_rn(qc, r1[c], a1);
            a2 = __fmaf_rn(qc, r2[c], a2);
            a3 = __fmaf_rn(qc, r3[c], a3);
        }
        dl[j0] = add_rn(add_rn(mul_rn(a0, -2.0f), quadb[j0]), qn);
        if (j1 < N) dl[j1] = add_rn(add_rn(mul_rn(a1, -2.0f), quadb[j1]), qn);
        if (j2 < N) dl[j2] = add_rn(add_rn(mul_rn(a2, -2.0f), quadb[j2]), qn);
        if (j3 < N) dl[j3] = add_rn(add_rn(mul_rn(a3, -2.0f), quadb[j3]), qn);
    }
    __syncthreads();
    if (dmat) for (int j = tid; j < N; j += 256) dmat[((size_t)b * N + j) * N + q] = dl[j];
    if (tid < 64) knn_select_wave_any(dl, sv, N, k, drop, idx + ((size_t)b * N + q) * k, tie ? tie + (size_t)b * N + q : nullptr);
}

// ------------------------------------------------------------------------------------------------
// remainder queries when the MFMA grid already fills the chip (B * tiles >= 2 workgroups per CU): extra
// workgroups in that launch would only slow their CUs down.  The inner products are symmetric -- the fma
// chain of (t, j) and of (j, t) multiplies the same pairs in the same order -- so the MFMA workgroups,
// which meet the remainder rows t as CANDIDATES of their 32 queries j, also write
// d(t, j) = ((inner * -2) + |x_j|^2) + |x_t|^2 to dtail[b][t][j]; this kernel adds the rem x rem
// remainder-remainder pairs (rows staged through LDS with coalesced loads, same k-ordered chain) and
// selects, one wave per remainder query.
// grid (ceil(rem/4), B), block 256, LDS = 4 * (12*(N+3) + (1+rem)*C*4)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_feat_sym_tail_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ quad,
                                                                const float* __restrict__ dtail, int N, int C,
                                                                int k, int drop, int nfull,
                                                                int32_t* __restrict__ idx, uint8_t* __restrict__ tie,
                                                                float* __restrict__ dmat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rem = N - nfull, b = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;
    if (t >= rem) return;
    const int Np = (N + 3) & ~3;
    const size_t per_wave = (size_t)3 * Np + (size_t)(1 + rem) * C;
    float* dl = reinterpret_cast<float*>(smem) + wave * per_wave;
    int2* sv = reinterpret_cast<int2*>(dl + Np);
    float* rows = reinterpret_cast<float*>(sv + Np);          // [query row | rem candidate rows]
    const int q = nfull + t;
    const float* xb = x + (size_t)b * N * C;
    const float* quadb = quad + (size_t)b * N;
    const float* drow = dtail + ((size_t)b * rem + t) * nfull;
    // all global loads first: the distances written by the MFMA workgroups and the 1 + rem rows
    for (int j0 = lane; j0 < nfull; j0 += 8 * 64) {
        float tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tv[u] = drow[min(j0 + u * 64, nfull - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + u * 64 < nfull) dl[j0 + u * 64] = tv[u];
    }
    for (int e = lane; e < (1 + rem) * C; e += 64) {
        const int r = e / C, c = e - r * C;
        rows[e] = xb[(size_t)(r == 0 ? q : nfull + r - 1) * C + c];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < rem) {
        const float* rc = rows + (size_t)(1 + lane) * C;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = __fmaf_rn(rows[c], rc[c], a);
        dl[nfull + lane] = add_rn(add_rn(mul_rn(a, -2.0f), quadb[nfull + lane]), quadb[q]);
    }
    __builtin_amdgcn_wave_barrier();
    if (dmat) for (int j = lane; j < N; j += 64) dmat[((size_t)b * N + j) * N + q] = dl[j];
    knn_select_wave_any(dl, sv, N, k, drop, idx + ((size_t)b * N + q) * k, tie ? tie + (size_t)b * N + q : nullptr);
}

// ------------------------------------------------------------------------------------------------
// feature path (any C that is a multiple of 64, or any even C via zero padding of the last chunk):
// one block = 32 queries of one cloud, 4 waves; wave w takes candidate tiles w, w+4, ... (32 rows
// each).  Distances of a 32x32 (candidate x query) tile come from v_mfma_f32_32x32x2_f32 over
// K-chunks of 64 staged through wave-private LDS (k de-interleaved into even|odd halves so every
// lane fetches its operands for 4 consecutive MFMA steps with one ds_read_b128).
//   lane l: query column j = l & 31, half h = l >> 5; acc[r] = inner(query j, candidate row
//   (r&3) + 8*(r>>2) + 4*h of the tile).
// Each lane keeps a sorted list for (query j, its half of the rows); the 8 lists per query
// (4 waves x 2 halves) are merged by tournament at the end.
// grid (ceil(N/32), B), block 256.
// ------------------------------------------------------------------------------------------------
#define KF_CT_STRIDE 68   // candidate chunk row stride in floats (64 + 4: 16B aligned, odd # of 16B slots)

// FULLK: C is a multiple of 64 (every shape of the HS stack).  Candidate chunks are then fetched with raw
// buffer loads: one descriptor per cloud, a loop-invariant per-lane voffset and a scalar soffset per
// element, so the 16 loads of a chunk cost no vector address arithmetic, and rows past N read as 0
// (their |c|^2 is staged as +inf, which makes the distance +inf without a per-candidate select).
template <int K1, bool FULLK>
__global__ __launch_bounds__(256, 3) void knn_feat_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ quad, int N, int C, int k,
                                                       int drop, int32_t* __restrict__ idx, int full_tiles,
                                                       int ntail, float* __restrict__ dtail, int msel,
                                                       uint8_t* __restrict__ tie, float* __restrict__ dmat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // remainder queries: one workgroup each.  They take the LOWEST block ids so that they are dispatched
    // first and run alongside the MFMA tiles instead of after them.
    if ((int)blockIdx.x < ntail) {
        knn_feat_tail_body<K1>(smem, x, quad, N, C, k, drop, full_tiles * 32 + (int)blockIdx.x, idx, tie, dmat);
        return;
    }
    const int Cp = (C + 63) & ~63;          // K padded to a multiple of 64 with zeros (adds exact 0s)
    const int QS = Cp + 4;                  // query row stride (floats)
    float* qtile = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    float* ctile = qtile + 32 * QS + wave * 32 * KF_CT_STRIDE;
    float* qsm = qtile + 32 * QS + 4 * 32 * KF_CT_STRIDE + wave * 32;   // |c|^2 of the current candidate tile
    // shared pruning bound.  List l = (wave, half) of query q publishes its a_l-th smallest distance so far,
    // sum_l a_l = K1 >= k + drop: the K1 candidates behind those entries are all <= tau_q = max_l pub[q][l], so
    // a candidate with d > tau_q has k + drop strictly closer ones and can never be selected.  Entries only
    // decrease, so a stale read is still a valid bound: no barrier, plain 32-bit LDS stores / loads.
    float* pub = qtile + 32 * QS + 4 * 32 * KF_CT_STRIDE + 4 * 32;     // [32 queries][8 lists]
    constexpr bool SHARE_OK = K1 >= 8;
    constexpr int A_HI = SHARE_OK ? K1 / 8 : 0, A_LO = SHARE_OK ? K1 / 8 - 1 : 0;   // a_l - 1 for l < K1 % 8, else
    const bool SHARE = SHARE_OK && ((N + 31) >> 5) >= 4;    // all four waves own a tile
    pub[tid] = INFINITY;
    const int b = blockIdx.y;
    const int q0 = ((int)blockIdx.x - ntail) * 32;
    const float* xb = x + (size_t)b * N * C;
    const float* quadb = quad + (size_t)b * N;

    // ---- stage the 32 query rows, de-interleaved per 64-chunk: [chunk][even 32 | odd 32]
    const int halfC = Cp >> 1;
    for (int e = tid; e < 32 * halfC; e += 256) {
        const int row = e / halfC, pair = e - row * halfC;
        const int kk = pair * 2;
        float2 v = make_float2(0.f, 0.f);
        if (q0 + row < N) {
            if (kk + 1 < C) v = *reinterpret_cast<const float2*>(xb + (size_t)(q0 + row) * C + kk);
            else if (kk < C) v.x = xb[(size_t)(q0 + row) * C + kk];
        }
        const int chunk = kk >> 6, within = pair & 31;
        qtile[row * QS + chunk * 64 + within] = v.x;
        qtile[row * QS + chunk * 64 + 32 + within] = v.y;
    }
    __syncthreads();

    const int q = q0 + col;
    const bool qvalid = q < N;
    const float qq = qvalid ? quadb[q] : 0.f;
    TopList<K1> top;
    top.init();

    const int ntiles = (N + 31) >> 5;
    const int nchunks = Cp >> 6;
    // register prefetch of one candidate chunk: 32 rows x 32 float2 = 16 float2 per lane; lane (h, pair)
    // takes rows h, h+2, ... of the tile, so every half-wave reads 256 contiguous bytes of one row.
    float2 pre[16];
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)((size_t)N * C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(quadb), 0, N * 4, 0x00020000);
    const int lane_off = (h * C + 2 * col) * 4;              // loop-invariant voffset (bytes)
    const int row2 = 2 * C * 4;                              // two rows down (bytes)
    const bool evenC = (C & 1) == 0;
    auto prefetch = [&](int tile, int chunk) {
        if (FULLK) {
            const int s0 = (tile * 32 * C + chunk * 64) * 4;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(xrs, lane_off, s0 + it * row2, 0);
                pre[it] = make_float2(__int_as_float(v[0]), __int_as_float(v[1]));
            }
        } else {
            // generic C, BRANCH-FREE on purpose: a per-element "in range ? load : 0" makes hipcc branch
            // around every load and wait vmcnt(0) before the next one.  Rows past N are clamped (their
            // |c|^2 is +inf), columns past C are clamped then zeroed by a select (exact 0s in the chain).
            const int c0 = tile * 32, kc = chunk * 64;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = it * 64 + lane;
                const int row = min(c0 + (e >> 5), N - 1), pair = e & 31;
                const int kk = kc + pair * 2;
                const float* rp = xb + (size_t)row * C;
                float2 v;
                if (evenC) {                                   // wave-uniform: 8-byte aligned pair loads
                    v = *reinterpret_cast<const float2*>(rp + min(kk, C - 2));
                } else {
                    v.x = rp[min(kk, C - 1)];
                    v.y = rp[min(kk + 1, C - 1)];
                }
                v.x = kk < C ? v.x : 0.f;
                v.y = kk + 1 < C ? v.y : 0.f;
                pre[it] = v;
            }
        }
    };

    int tile = wave, chunk = 0;
    if (tile < ntiles) prefetch(tile, 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    unsigned qraw = 0;                                      // |c|^2 bits of candidate tile*32 + col

    while (tile < ntiles) {
        // registers -> wave-private LDS chunk (in-order LDS ops of one wave: no barrier required)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int e = it * 64 + lane;
            const int row = e >> 5, pair = e & 31;
            ctile[row * KF_CT_STRIDE + pair] = pre[it].x;
            ctile[row * KF_CT_STRIDE + 32 + pair] = pre[it].y;
        }
        // |candidate|^2 of the tile's 32 rows (one per lane pair), +inf past N: in flight under the MFMAs
        if (chunk == 0) {
            qraw = __builtin_amdgcn_raw_buffer_load_b32(qrs, (tile * 32 + col) * 4, 0, 0);   // reads 0 past N
        }
        // next (tile, chunk) of this wave
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == nchunks) { nchunk = 0; ntile = tile + 4; }
        if (ntile < ntiles) prefetch(ntile, nchunk);
        __builtin_amdgcn_wave_barrier();

        const float* arow = ctile + col * KF_CT_STRIDE + h * 32;
        const float* brow = qtile + col * QS + chunk * 64 + h * 32;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
            const float4 b4 = *reinterpret_cast<const float4*>(brow + 4 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();

        if (chunk == nchunks - 1) {
            // tile finished: fold the 16 distances of this lane into its list (rows ascend with r).
            // A candidate can only matter if d <= tau_q, the shared bound below, and d < this list's worst;
            // the survivors (few once the lists have warmed up) are inserted by a drain loop that runs
            // max-over-lanes(#survivors) times instead of once per candidate.
            qsm[col] = tile * 32 + col < N ? __uint_as_float(qraw) : INFINITY;   // both halves: same value
            __builtin_amdgcn_wave_barrier();
            if (dtail && tile == ntiles - 1) {
                // symmetric remainder path (knn_feat_sym_tail_kernel): the rows of this partial tile are the
                // remainder QUERIES t; their distance to this lane's query j, in the association of query t
                const int nfull = full_tiles * 32, rem = N - nfull;
                if (h * 4 < rem) {
                    const float4 qc = *reinterpret_cast<const float4*>(qsm + 4 * h);
                    const float qcv[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (h * 4 + u < rem)
                            dtail[((size_t)b * rem + h * 4 + u) * nfull + q] =
                                add_rn(add_rn(mul_rn(acc[u], -2.0f), qq), qcv[u]);
                }
            }
            float thr = top.d[K1 - 1];
            if (SHARE) {
                const float4 t0 = *reinterpret_cast<const float4*>(pub + col * 8);
                const float4 t1 = *reinterpret_cast<const float4*>(pub + col * 8 + 4);
                const float tau = fmaxf(fmaxf(fmaxf(t0.x, t0.y), fmaxf(t0.z, t0.w)),
                                        fmaxf(fmaxf(t1.x, t1.y), fmaxf(t1.z, t1.w)));
                thr = fminf(thr, tau);
            }
            float* stash = ctile;                           // the chunk is consumed: 16 x 64 floats fit
            unsigned m = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 qc = *reinterpret_cast<const float4*>(qsm + 8 * g + 4 * h);
                const float qcv[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * g + u;
                    const float d = add_rn(add_rn(mul_rn(acc[r], -2.0f), qcv[u]), qq);   // +inf past N
                    stash[r * 64 + lane] = d;
                    m |= d <= thr ? (1u << r) : 0u;          // conservative: insert() re-checks d < worst
                    acc[r] = 0.f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int c0 = tile * 32 + 4 * h;
            if (dmat && qvalid) {                           // (exact scope, small batches) the distance matrix, [candidate][query]:
#pragma unroll
                for (int r = 0; r < 16; ++r) {              // 32 queries = 128 contiguous bytes per store
                    const int cand = c0 + (r & 3) + 8 * (r >> 2);
                    if (cand < N) dmat[((size_t)b * N + cand) * N + q] = stash[r * 64 + lane];
                }
            }
            // software-pipelined: the next survivor is fetched from the stash while the current one is inserted
            bool has = m != 0;
            int r = has ? __builtin_ctz(m) : 0;              // ascending r == ascending candidate index
            m &= m - 1;                                      // 0 stays 0
            float v = stash[r * 64 + lane];
#pragma unroll
            for (int it = 0; it < 16; ++it) {                // wave-uniform early-out, straight-line bodies
                if (!__any(has)) break;
                const bool has_n = m != 0;
                const int r_n = has_n ? __builtin_ctz(m) : 0;
                m &= m - 1;
                const float v_n = stash[r_n * 64 + lane];
                top.insert_always(has ? v : INFINITY, c0 + (r & 3) + 8 * (r >> 2));
                has = has_n; r = r_n; v = v_n;
            }
            if (SHARE) pub[col * 8 + wave * 2 + h] = (wave * 2 + h < (K1 & 7)) ? top.d[A_HI] : top.d[A_LO];
            __builtin_amdgcn_wave_barrier();
        }
        tile = ntile;
        chunk = nchunk;
    }

    // ---- merge the 8 lists of every query
    __syncthreads();                                   // tiles are dead: alias the region with the lists
    int2* lists = reinterpret_cast<int2*>(smem);
    top.store(lists + (size_t)tid * K1);
    __syncthreads();
    const int mq = tid >> 3, ml = tid & 7;             // merge thread -> (query, list)
    const int src = (ml >> 1) * 64 + (ml & 1) * 32 + mq;
    const int oq = q0 + mq;
    const bool ovalid = oq < N;
    merge_write<K1, 8>(lists, src, ml, k, drop, ovalid, idx + ((size_t)b * N + (ovalid ? oq : 0)) * k, N, ovalid ? oq : 0, nullptr,
                       msel, tie ? tie + (size_t)b * N + (ovalid ? oq : 0) : nullptr);
}

// ------------------------------------------------------------------------------------------------
// feature path, SMALL clouds (64 <= N <= 320: the two coarse levels of the stack, N = 257 and N = 64; C % 64 == 0).
// knn_feat_kernel gives each of its four waves every fourth candidate tile and a sorted list per lane; at N = 257 that is 144
// workgroups whose wave 0 walks three tiles one after the other, inserts every candidate of the first one and then merges eight
// 21-entry lists: 25 / 41 us (C = 128 / 256) for 1.7 / 3.5 us of matrix work.  Here a workgroup is QT queries of one cloud and ONE
// WAVE PER CANDIDATE TILE (9 waves at N = 257): every wave runs a single 32 x 32 chain of k-ordered MFMA steps (the same chain:
// identical distance bits), the QT x N distances go to LDS and the queries are then selected by knn_select_wave (radix bound over
// ballots + ranking of the ~k + 4 survivors, ~600 wave instructions a query), QT / waves rounds per wave.
//   |x|^2 is computed in the kernel (quad_in == nullptr): each wave has its 32 candidate rows in LDS anyway and adds their squares
//   in ATen's order (quad32_kernel's: partial accumulator s = 8 k + l takes the elements 32 i + s in ascending i, then slots 1..3
//   into slot 0, then the eight vector lanes in order) -- lane (row, hh) keeps the sixteen accumulators s = 16 hh .. 16 hh + 15 of
//   its row.  One launch per search instead of two (the |x|^2 kernel was ~5 us of pure latency in front of each).
// QT <= 32 is chosen so that the batch's workgroups just fill the 256 CUs (QT = 17 at B = 16, N = 257: 16 x 16 workgroups; the
// unused query columns of the MFMA repeat the last query): the matrix work of a workgroup does not depend on QT -- one chain per
// wave -- and the selection, which is bound by instruction issue, shrinks with it.  A remainder of at most 4 rows (N = 257) is not
// given an MFMA tile but one wave's fma chains.  grid (ceil(N / QT), B), block 64 * max(tiles, 8) (waves beyond the tiles only select).
// ------------------------------------------------------------------------------------------------
#ifdef HSP_KNN_PROF
static __device__ long long* g_knn_prof = nullptr;     // tools/prof_knn_small.py: clock64 stamps of workgroup (0, 0), lane 0 of every wave
#define KNN_STAMP(slot) do { if (g_knn_prof && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_knn_prof[(threadIdx.x >> 6) * 8 + (slot)] = clock64(); } while (0)
#else
#define KNN_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(640) void knn_feat_small_kernel(const float* __restrict__ x, const float* __restrict__ quad_in,
                                                             float* __restrict__ quad_out, int N, int C, int k, int drop,
                                                             int32_t* __restrict__ idx, uint8_t* __restrict__ tie,
                                                             float* __restrict__ dmat, int mtiles, int QT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nw = (int)blockDim.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int ntiles = (N + 31) >> 5, nchunks = C >> 6;
    const int rem = N - mtiles * 32;          // > 0: the last rows (at most 4) are not a candidate tile but extra QUERY columns (below)
    const int QR = QT + (rem > 0 ? rem : 0);  // rows of the query tile: the workgroup's queries, then the remainder rows
    const int QS = C + 4;                     // query row stride (floats)
    const int NP = ntiles * 32, DS = NP + 1;  // distance row stride: odd, so the 32 query columns of a store hit 32 banks
    float* qtile = reinterpret_cast<float*>(smem);                 // QR x QS
    float* ctiles = qtile + QR * QS;                               // mtiles x 32 x KF_CT_STRIDE (wave-private chunks)
    float* quadl = ctiles + (size_t)ntiles * 32 * KF_CT_STRIDE;    // NP: |c|^2, +inf past N
    float* dl = quadl + NP;                                        // QT x DS distances
    const int b = blockIdx.y, q0 = (int)blockIdx.x * QT;
    const float* xb = x + (size_t)b * N * C;
    const bool mma = wave < mtiles;
    const bool remw = rem > 0 && wave == mtiles;
    float* ctile = ctiles + (size_t)(mma ? wave : 0) * 32 * KF_CT_STRIDE;

    KNN_STAMP(0);
    // this wave's candidate tile, chunk 0: in flight under the query staging
    float2 pre[16];
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)((size_t)N * C * 4), 0x00020000);
    const int lane_off = (h * C + 2 * col) * 4, row2 = 2 * C * 4;
    auto prefetch = [&](int chunk) {
        const int s0 = (wave * 32 * C + chunk * 64) * 4;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(xrs, lane_off, s0 + it * row2, 0);   // rows past N read 0
            pre[it] = make_float2(__int_as_float(v[0]), __int_as_float(v[1]));
        }
    };
    if (mma) prefetch(0);
    // ---- the QR query rows, de-interleaved per 64-chunk: [chunk][even 32 | odd 32]; four 16-byte loads in flight per thread
    {
        const int C4 = C >> 2, total = QR * C4, bd = (int)blockDim.x;
        for (int e0 = tid; e0 < total; e0 += 4 * bd) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * bd;
                const int row = e / C4, p = e - row * C4;
                const int src = row < QT ? q0 + row : mtiles * 32 + (row - QT);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < total && src < N) v[u] = *reinterpret_cast<const float4*>(xb + (size_t)src * C + 4 * p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * bd;
                if (e < total) {
                    const int row = e / C4, p = e - row * C4;
                    float* dst = qtile + row * QS + (p >> 4) * 64 + 2 * (p & 15);     // elements 4p .. 4p+3: pairs 2p, 2p+1 of the row
                    *reinterpret_cast<float2*>(dst) = make_float2(v[u].x, v[u].z);
                    *reinterpret_cast<float2*>(dst + 32) = make_float2(v[u].y, v[u].w);
                }
            }
        }
    }
    __syncthreads();
    KNN_STAMP(1);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ra = 0.f;                                          // remainder wave: inner product of (remainder query, remainder row) of this lane
    if (mma) {
        float sq[16];                                       // ATen's partial accumulators 16 h .. 16 h + 15 of candidate row `col`
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) sq[s_] = 0.f;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = it * 64 + lane;
                const int row = e >> 5, pair = e & 31;
                ctile[row * KF_CT_STRIDE + pair] = pre[it].x;
                ctile[row * KF_CT_STRIDE + 32 + pair] = pre[it].y;
            }
            if (chunk + 1 < nchunks) prefetch(chunk + 1);
            __builtin_amdgcn_wave_barrier();
            const float* arow = ctile + col * KF_CT_STRIDE + h * 32;
            const float* brow = qtile + min(col, QR - 1) * QS + chunk * 64 + h * 32;
            if (!quad_in) {
                // elements 64 c + s (i = 2 c) then 64 c + 32 + s (i = 2 c + 1) of row `col`, s = 16 h + 2 j (+ 1): the even element of a
                // pair sits at [s / 2], the odd one at [32 + s / 2] of the staged chunk
                const float* sr = ctile + col * KF_CT_STRIDE + 8 * h;
                const float4 e0 = *reinterpret_cast<const float4*>(sr), e1 = *reinterpret_cast<const float4*>(sr + 4);
                const float4 o0 = *reinterpret_cast<const float4*>(sr + 32), o1 = *reinterpret_cast<const float4*>(sr + 36);
                const float4 f0 = *reinterpret_cast<const float4*>(sr + 16), f1 = *reinterpret_cast<const float4*>(sr + 20);
                const float4 p0 = *reinterpret_cast<const float4*>(sr + 48), p1 = *reinterpret_cast<const float4*>(sr + 52);
                const float ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                const float od[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                const float ev2[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                const float od2[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sq[2 * j] = add_rn(sq[2 * j], mul_rn(ev[j], ev[j]));
                    sq[2 * j + 1] = add_rn(sq[2 * j + 1], mul_rn(od[j], od[j]));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sq[2 * j] = add_rn(sq[2 * j], mul_rn(ev2[j], ev2[j]));
                    sq[2 * j + 1] = add_rn(sq[2 * j + 1], mul_rn(od2[j], od2[j]));
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
                const float4 b4 = *reinterpret_cast<const float4*>(brow + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        KNN_STAMP(2);
        const int cand = wave * 32 + col;
        float qv;
        if (quad_in) {
            qv = cand < N ? quad_in[(size_t)b * N + cand] : INFINITY;
        } else {
            // slots 1..3 into slot 0: t_l = ((a_l + a_{l+8}) + a_{l+16}) + a_{l+24}; this half holds a_{16 h + 0..15}
            float t[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const float lo = add_rn(sq[l], sq[l + 8]);                   // h == 0: a_l + a_{l+8}
                const float p16 = __shfl_xor(sq[l], 32), p24 = __shfl_xor(sq[l + 8], 32);   // the other half's a_{16 + l}, a_{24 + l}
                t[l] = add_rn(add_rn(lo, p16), p24);                          // (meaningful in the h == 0 lanes)
            }
            float fin = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) fin = add_rn(fin, t[l]);
            qv = cand < N ? fin : INFINITY;
            if (quad_out && blockIdx.x == 0 && h == 0 && cand < N) quad_out[(size_t)b * N + cand] = fin;
        }
        if (h == 0) quadl[cand] = qv;
        KNN_STAMP(3);
    } else if (remw) {
        // ---- the N % 32 <= 4 last rows.  As CANDIDATES of the workgroup's queries they cost no tile: they ride as query columns
        // QT .. QT + rem - 1 of every MFMA, and the chain of (candidate row j, column of remainder row t) IS the chain of (query j,
        // candidate t) -- the same products in the same k order -- so the tile that holds the workgroup's own queries as candidate
        // rows delivers d(q, t) (epilogue below).  What no tile holds are the rem x rem pairs of remainder QUERIES (only the last
        // workgroups of a cloud own any): spelled-out fma chains, lane = (own remainder query, remainder row), both rows from the
        // staged query tile.  This wave also computes the remainder rows' |x|^2 (quad32_kernel's order, 32 lanes per row).
        const int tq = lane / rem, tc = lane - tq * rem;             // query mtiles*32 + tq against candidate mtiles*32 + tc
        const int qrow = mtiles * 32 + tq - q0;                       // its row of the query tile, if this workgroup owns it
        if (tq < rem && qrow >= 0 && qrow < QT) {
            const float* qa = qtile + qrow * QS;
            const float* qb = qtile + (QT + tc) * QS;
            float a = 0.f;
            for (int c8 = 0; c8 < C; c8 += 8) {
                const int o = (c8 >> 6) * 64 + ((c8 & 63) >> 1);
                const float4 ae = *reinterpret_cast<const float4*>(qa + o), ao = *reinterpret_cast<const float4*>(qa + o + 32);
                const float4 be = *reinterpret_cast<const float4*>(qb + o), bo = *reinterpret_cast<const float4*>(qb + o + 32);
                a = __fmaf_rn(ao.y, bo.y, __fmaf_rn(ae.y, be.y, __fmaf_rn(ao.x, bo.x, __fmaf_rn(ae.x, be.x, a))));
                a = __fmaf_rn(ao.w, bo.w, __fmaf_rn(ae.w, be.w, __fmaf_rn(ao.z, bo.z, __fmaf_rn(ae.z, be.z, a))));
            }
            ra = a;
        }
        for (int t = h; t < 32; t += 2) {
            const int cand = mtiles * 32 + t;
            float fin = INFINITY;
            if (t < rem) {                                    // (uniform per half-wave)
                if (quad_in) {
                    fin = quad_in[(size_t)b * N + cand];
                } else {
                    const float* row = xb + (size_t)cand * C;
                    float a = 0.f;
                    for (int i = 0; i < (C >> 5); ++i) { const float v = row[i * 32 + col]; a = add_rn(a, mul_rn(v, v)); }
                    const int base = lane & 32;
                    float tt = a;
                    tt = add_rn(tt, __shfl(a, base | ((col + 8) & 31)));
                    tt = add_rn(tt, __shfl(a, base | ((col + 16) & 31)));
                    tt = add_rn(tt, __shfl(a, base | ((col + 24) & 31)));
                    fin = 0.f;
#pragma unroll
                    for (int l = 0; l < 8; ++l) fin = add_rn(fin, __shfl(tt, base | l));
                    if (quad_out && blockIdx.x == 0 && col == 0) quad_out[(size_t)b * N + cand] = fin;
                }
            }
            if (col == 0) quadl[cand] = fin;
        }
        KNN_STAMP(3);
    }
    __syncthreads();
    KNN_STAMP(4);
    if (mma) {
        // (wave-uniform) does this tile hold some of the workgroup's own queries as candidate rows?  Only then do the remainder
        // rows' columns carry anything (at most two waves of a workgroup)
        const bool holds_own = rem > 0 && wave * 32 < q0 + QT && wave * 32 + 32 > q0;
        if (col < QT) {                                     // own queries against this wave's candidate tile
            const int q = q0 + col;
            const float qq = q < N ? quadl[q] : 0.f;
            float* drow = dl + col * DS + wave * 32 + 4 * h;
            float* mrow = dmat && q < N ? dmat + ((size_t)b * N + wave * 32 + 4 * h) * N + q : nullptr;   // [candidate][query], as knn_feat_kernel
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 qc = *reinterpret_cast<const float4*>(quadl + wave * 32 + 8 * g + 4 * h);
                const float qcv[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = add_rn(add_rn(mul_rn(acc[4 * g + u], -2.0f), qcv[u]), qq);
                    drow[8 * g + u] = d;                                                  // (rows past N: +inf, never read)
                    if (mrow && wave * 32 + 8 * g + 4 * h + u < N) mrow[(size_t)(8 * g + u) * N] = d;
                }
            }
        } else if (holds_own && col < QR) {                 // a remainder row's column: rows of this tile that are OWN QUERIES
            const int c = mtiles * 32 + (col - QT);
            const float qcand = quadl[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (q >= q0 && q < q0 + QT) {
                    const float d = add_rn(add_rn(mul_rn(acc[r], -2.0f), qcand), quadl[q]);
                    dl[(q - q0) * DS + c] = d;
                    if (dmat) dmat[((size_t)b * N + c) * N + q] = d;
                }
            }
        }
    } else if (remw) {
        const int tq = lane / rem, tc = lane - tq * rem;
        const int q = mtiles * 32 + tq, c = mtiles * 32 + tc;
        if (tq < rem && q >= q0 && q < q0 + QT) {
            const float d = add_rn(add_rn(mul_rn(ra, -2.0f), quadl[c]), quadl[q]);
            dl[(q - q0) * DS + c] = d;
            if (dmat) dmat[((size_t)b * N + c) * N + q] = d;
        }
    }
    __syncthreads();                                        // distances complete; the candidate chunks are dead: selection scratch
    KNN_STAMP(5);
    // two queries per wave and pass (knn_select_wave_regs): scratch = 128 survivor slots + N for the fall-back
    int2* sv = reinterpret_cast<int2*>(ctiles) + (size_t)wave * (128 + N);
    const int nq = min(QT, N - q0);                         // queries of this workgroup
    if (nq <= nw) {                                         // a query per wave is enough
        if (wave < nq) {
            const float* const dls[1] = {dl + wave * DS};
            int32_t* const outs[1] = {idx + ((size_t)b * N + q0 + wave) * k};
            uint8_t* const ties[1] = {tie ? tie + (size_t)b * N + q0 + wave : nullptr};
            if (tie) knn_select_wave_regs<5, 1, true>(dls, sv, N, k, drop, outs, ties);
            else knn_select_wave_regs<5, 1, false>(dls, sv, N, k, drop, outs, ties);
        }
    } else {
        for (int ql = 2 * wave; ql < nq; ql += 2 * nw) {
            const int q = q0 + ql;
            const int ql1 = ql + 1 < nq ? ql + 1 : ql;      // (an odd count: the last query is simply done twice)
            const float* const dls[2] = {dl + ql * DS, dl + ql1 * DS};
            int32_t* const outs[2] = {idx + ((size_t)b * N + q) * k, idx + ((size_t)b * N + q0 + ql1) * k};
            uint8_t* const ties[2] = {tie ? tie + (size_t)b * N + q : nullptr, tie ? tie + (size_t)b * N + q0 + ql1 : nullptr};
            if (tie) knn_select_wave_regs<5, 2, true>(dls, sv, N, k, drop, outs, ties);
            else knn_select_wave_regs<5, 2, false>(dls, sv, N, k, drop, outs, ties);
            __builtin_amdgcn_wave_barrier();
        }
    }
    KNN_STAMP(6);
}

// ------------------------------------------------------------------------------------------------
// top-1 nearest source row per target row (C == 3); d = (s2[j] + t2[i]) - 2*inner   (gcn3d.py:34)
// grid (ceil(Nt/256), B), block 256, dynamic LDS = Ns*16
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void nn1_body(char* smem, const float* __restrict__ tb, int Nt, const float* __restrict__ sb,
                                         const int32_t* __restrict__ sel, const int32_t* __restrict__ sel_outer, int Ns,
                                         int32_t* __restrict__ idx_b, int tblock) {
    float4* pts = reinterpret_cast<float4*>(smem);
    for (int j = threadIdx.x; j < Ns; j += 256) {
        int r = sel ? sel[j] : j;                                  // (source row j as a row of the cloud sb: see knn3_wave_body)
        r = sel_outer ? sel_outer[r] : r;
        const float px = sb[r * 3], py = sb[r * 3 + 1], pz = sb[r * 3 + 2];
        pts[j] = make_float4(px, py, pz, quad3(px, py, pz));
    }
    __syncthreads();
    const int i = tblock * 256 + threadIdx.x;
    if (i >= Nt) return;
    const float* tp = tb + (size_t)i * 3;
    const float tx = tp[0], ty = tp[1], tz = tp[2];
    const float t2 = quad3(tx, ty, tz);
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < Ns; ++j) {
        const float4 c = pts[j];
        const float inner = dot3_chain(tx, ty, tz, c.x, c.y, c.z);
        const float d = sub_rn(add_rn(c.w, t2), mul_rn(2.0f, inner));
        if (j == 0 || d < best) { best = d; bi = j; }
    }
    idx_b[i] = bi;
}

__global__ __launch_bounds__(256) void nn1_kernel(const float* __restrict__ tgt, int Nt,
                                                  const float* __restrict__ src, int Ns,
                                                  int32_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    nn1_body(smem, tgt + (size_t)b * Nt * 3, Nt, src + (size_t)b * Ns * 3, nullptr, nullptr, Ns, idx + (size_t)b * Nt, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The geometry of the two coarse levels of the stack in ONE launch.  Pool_layer keeps a random subset of its input points
// (gcn3d.py:243-245), so the level-1 cloud is rows sel1 of the input cloud and the level-2 cloud rows sel2 of level 1 -- both known
// before the forward starts (the draws are host-side).  Everything the forward later asks of them depends on coordinates only:
//   level 1: get_neighbor_index(v1, k1) (RF-P + the ORL branches of conv_2 / conv_3) and Pool_layer's own k = kpool list,
//   level 2: get_neighbor_index(v2, k2) (conv_4),
//   get_nearest_index(vertices, v1), get_nearest_index(vertices, v2)                         (FaceRecon.py:100-101),
// four mutually independent small searches that were four launches of 6-13 us spread over the forward (each mostly latency);
// as block ranges of one grid they run side by side.  Same bodies as knn3_wave_kernel / nn1_kernel: identical lists, tie replay
// included.  1-D grid, block 256.
// ------------------------------------------------------------------------------------------------
struct GeoArgs {
    const float* xyz; int N0;
    const int32_t* sel1; int N1;
    const int32_t* sel2; int N2;
    int k1, kpool, k2, drop;
    float* v1; float* v2;
    int32_t* idx1; int32_t* idx1p; int32_t* idx2; int32_t* up1; int32_t* up2;
    int nb1, nb2, nbt;
    // optional rider (hsp_geometry_all_f32): the tie pass of the LEVEL-0 search, which depends on that search's flags only and
    // otherwise costs its own launch right behind it (one flagged row = 14 us of latency with the chip idle)
    const uint8_t* tie0; int B, k0, kpool0; int32_t* idx0; int32_t* idx0p; int nbtie;
};

template <int S>
__global__ __launch_bounds__(256) void geometry_levels_kernel(GeoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid: [tie workers of the whole batch | per cloud: level-1 search, level-2 search, the two nearest-point maps].  The tie
    // workers come FIRST in dispatch order: a flagged row is the longest job of the launch.
    int t = (int)blockIdx.x;
    if (t < a.nbtie) {
        const int wave = threadIdx.x >> 6;                    // (four waves per workgroup, each with its own scratch row)
        tie_rows_xyz(smem + (size_t)wave * 16 * a.N0, a.xyz, a.tie0, a.B, a.N0, a.k0, a.kpool0, a.drop, a.idx0, a.idx0p, nullptr,
                     t * 4 + wave, a.nbtie * 4, threadIdx.x & 63);
        return;
    }
    t -= a.nbtie;
    const int per_cloud = a.nb1 + a.nb2 + 2 * a.nbt;
    const int b = t / per_cloud;
    t -= b * per_cloud;
    const float* xb = a.xyz + (size_t)b * a.N0 * 3;
    if (t < a.nb1) {
        const int m = a.k1 + a.drop;
        knn3_wave_body<S>(smem, xb, a.sel1, nullptr, a.v1 + (size_t)b * a.N1 * 3, a.N1, a.k1, a.drop, a.idx1, m + 1 < a.N1 ? m + 1 : a.N1,
                          reinterpret_cast<uint8_t*>(smem) /* non-null: detect ties */, a.kpool > 0 ? a.kpool + a.drop + 1 : 0, a.idx1p,
                          a.kpool, 1, (size_t)b * a.N1, t, 1);
        return;
    }
    t -= a.nb1;
    if (t < a.nb2) {
        const int m = a.k2 + a.drop;
        knn3_wave_body<S>(smem, xb, a.sel2, a.sel1, a.v2 + (size_t)b * a.N2 * 3, a.N2, a.k2, a.drop, a.idx2, m + 1 < a.N2 ? m + 1 : a.N2,
                          reinterpret_cast<uint8_t*>(smem), 0, nullptr, 0, 1, (size_t)b * a.N2, t, 1);
        return;
    }
    t -= a.nb2;
    if (t < a.nbt) {
        nn1_body(smem, xb, a.N0, xb, a.sel1, nullptr, a.N1, a.up1 + (size_t)b * a.N0, t);
        return;
    }
    t -= a.nbt;
    nn1_body(smem, xb, a.N0, xb, a.sel2, a.sel1, a.N2, a.up2 + (size_t)b * a.N0, t);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int pick_k1(int m) {
    if (m <= 3) return 3;
    if (m <= 5) return 5;
    if (m <= 6) return 6;               // (6, 10, 22: one entry past the network's 5 / 9 / 21 -- the exact search's boundary check)
    if (m <= 9) return 9;
    if (m <= 10) return 10;
    if (m <= 17) return 17;
    if (m <= 21) return 21;
    if (m <= 22) return 22;
    if (m <= 33) return 33;
    return 0;
}

template <int K1, int T>
static int launch_knn3(const float* x, int B, int N, int k, int drop, int32_t* idx, hipStream_t st, int msel, uint8_t* tie,
                       int msel2, int32_t* idx2, int k2) {
    const int chunk = N < 4096 ? N : 4096;
    size_t lds = (size_t)chunk * 16;
    if (lds < (size_t)256 * K1 * 8) lds = (size_t)256 * K1 * 8;
    auto kern = knn3_kernel<K1, T>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    constexpr int Q = 256 / T;
    dim3 grid((N + Q - 1) / Q, B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, N, k, drop, idx, chunk, msel, tie, msel2, idx2, k2);
    return check_launch();
}

// small clouds replay their flagged rows inside the selection kernel; larger ones keep the separate pass (see knn3_wave_kernel)
static int knn3_wave_tie_inline(int N) { return N <= 64 * 9 ? 1 : 0; }

template <int S>
static int launch_knn3_wave(const float* x, int B, int N, int k, int drop, int32_t* idx, hipStream_t st, int msel, uint8_t* tie,
                            int msel2, int32_t* idx2, int k2) {
    const int tie_inline = tie ? knn3_wave_tie_inline(N) : 0;
    const size_t lds = (size_t)N * 16 + (size_t)4 * KNN3W_SV * 8 + 4 * 64 * 4 +
                       (tie_inline ? (size_t)16 * N * 4 : 0);
    auto kern = knn3_wave_kernel<S>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    // queries per wave: four amortise the cloud's staging; a small batch (the coarse levels) takes one so that its few thousand
    // queries spread over the chip instead of queueing four deep behind 272 workgroups
    const int QW = (long long)B * N >= 8192 ? KNN3W_QW : 1;
    hipLaunchKernelGGL(kern, dim3((N + 4 * QW - 1) / (4 * QW), B), dim3(256), lds, st, x, N, k, drop, idx, msel, tie, msel2, idx2, k2,
                       tie_inline, QW);
    return check_launch();
}

template <int K1>
static int launch_knn3_t(const float* x, int B, int N, int k, int drop, int32_t* idx, hipStream_t st, int msel = 0,
                         uint8_t* tie = nullptr, int msel2 = 0, int32_t* idx2 = nullptr, int k2 = 0) {
    const long long nq = (long long)B * N;
    // one wave per query while the per-lane distances fit in registers (<= 65 per lane: N <= 4160, the dense clouds of
    // BASELINE configs[3]; 1.8 ms with the per-lane-list kernel below at B=64 N=4096)
    if (N >= 64 && (nq < 131072 || N > 64 * 17)) {
        if (N <= 64 * 5) return launch_knn3_wave<5>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        if (N <= 64 * 9) return launch_knn3_wave<9>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        if (N <= 64 * 17) return launch_knn3_wave<17>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        if (N <= 64 * 33) return launch_knn3_wave<33>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        if (N <= 64 * 65) return launch_knn3_wave<65>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
    }
    if (nq >= 131072) return launch_knn3<K1, 1>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
    if (nq >= 32768) return launch_knn3<K1, 4>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
    return launch_knn3<K1, 16>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
}

// how the N % 32 remainder queries of the feature path are handled
enum { KF_REM_TILE = 0,    // as a 33rd, nearly empty MFMA query tile
       KF_REM_WG = 1,      // one extra workgroup per query in the same launch (grid below 2 workgroups / CU: they run on idle CUs)
       KF_REM_SYM = 2 };   // symmetric path + knn_feat_sym_tail_kernel (grid already fills the chip)

static size_t knn_feat_lds(int C, int K1) {
    const int Cp = (C + 63) & ~63;
    const size_t lds = (size_t)(32 * (Cp + 4) + 4 * 32 * KF_CT_STRIDE + 4 * 32 + 32 * 8) * 4;
    const size_t lds_lists = (size_t)256 * K1 * 8;
    return lds_lists > lds ? lds_lists : lds;
}

static int knn_feat_rem_mode(int B, int N, int C) {
    const int rem = N & 31;
    if (rem == 0 || rem > 8 || N < 64) return KF_REM_TILE;
    const long long full = (long long)(N / 32) * B;
    if (full >= 512) {
        const size_t lds_s = (size_t)4 * (12 * ((size_t)N + 3) + (size_t)(1 + rem) * C * 4);
        return lds_s <= 160 * 1024 ? KF_REM_SYM : KF_REM_TILE;
    }
    const size_t lds_t = (size_t)C * 4 + (size_t)12 * (N + 3);
    return ((C & 3) == 0 && lds_t <= knn_feat_lds(C, 3)) ? KF_REM_WG : KF_REM_TILE;
}

// knn_feat_small_kernel: LDS bytes for QT queries per workgroup, and the QT that makes the batch's workgroups fill the chip once
static size_t knn_feat_small_lds(int N, int C, int QT) {
    const int ntiles = (N + 31) / 32, rem = (N & 31) <= 4 ? (N & 31) : 0;
    const size_t fl = (size_t)(QT + rem) * (C + 4) + (size_t)ntiles * 32 * KF_CT_STRIDE + (size_t)ntiles * 32 + (size_t)QT * (ntiles * 32 + 1);
    // (the selection scratch aliases the candidate chunks: (128 + N) int2 per wave)
    const int nw = ntiles > 8 ? ntiles : 8;
    const size_t scratch = (size_t)nw * (128 + N) * 2, chunks = (size_t)ntiles * 32 * KF_CT_STRIDE;
    return (fl + (scratch > chunks ? scratch - chunks : 0)) * 4;
}
static int knn_feat_small_qt(int B, int N, int C) {           // 0: not this kernel's shape
    if (N < 64 || N > 320 || (C & 63) || C >= 512 ||   // (C >= 512: ATen sums |x|^2 in cascade levels, quad_kernel)
        (size_t)N * C * 4 >= ((size_t)1 << 31)) return 0;
    const int rem = (N & 31) <= 4 ? (N & 31) : 0;
    const int per_cloud = HSP_NUM_CU / B > 0 ? HSP_NUM_CU / B : 1;      // workgroups a cloud may have
    int qt = (N + per_cloud - 1) / per_cloud;
    if (qt < 2) qt = 2;
    if (qt > 32 - rem) qt = 32 - rem;                          // (the remainder rows ride as query columns)
    while (qt > 2 && knn_feat_small_lds(N, C, qt) > 160 * 1024) --qt;
    return knn_feat_small_lds(N, C, qt) <= 160 * 1024 ? qt : 0;
}

static int launch_knn_feat_small(int QT, const float* x, const float* quad_in, float* quad_out, int B, int N, int C, int k, int drop,
                                 int32_t* idx, hipStream_t st, uint8_t* tie, float* dmat) {
    const size_t lds = knn_feat_small_lds(N, C, QT);
    auto kern = knn_feat_small_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const int ntiles = (N + 31) / 32;
    const int rem = N & 31;
    const int mtiles = (rem > 0 && rem <= 4) ? N / 32 : ntiles;      // MFMA tiles; a short remainder goes to one wave's fma chains
    const int nw = ntiles > 8 ? ntiles : 8;
    hipLaunchKernelGGL(kern, dim3((N + QT - 1) / QT, B), dim3(64 * nw), lds, st, x, quad_in, quad_out, N, C, k, drop, idx, tie, dmat,
                       mtiles, QT);
    return check_launch();
}

template <int K1>
static int launch_knn_feat(const float* x, const float* quad, float* dtail, int B, int N, int C, int k, int drop,
                           int32_t* idx, hipStream_t st, int msel = 0, uint8_t* tie = nullptr, float* dmat = nullptr) {
    const size_t lds = knn_feat_lds(C, K1);
    const bool fullk = (C & 63) == 0 && (size_t)N * C * 4 < ((size_t)1 << 31);
    auto kern = fullk ? knn_feat_kernel<K1, true> : knn_feat_kernel<K1, false>;
    const int mode = knn_feat_rem_mode(B, N, C);
    const int rem = mode == KF_REM_TILE ? 0 : (N & 31);
    const int full_tiles = (N - rem + 31) / 32;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const int ntail = mode == KF_REM_WG ? rem : 0;
    hipLaunchKernelGGL(kern, dim3(full_tiles + ntail, B), dim3(256), lds, st, x, quad, N, C, k, drop, idx, full_tiles,
                       ntail, mode == KF_REM_SYM ? dtail : nullptr, msel, tie, dmat);
    int rc = check_launch();
    if (rc || mode != KF_REM_SYM) return rc;
    const size_t lds_s = (size_t)4 * (12 * ((size_t)N + 3) + (size_t)(1 + rem) * C * 4);
    if (lds_s > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_feat_sym_tail_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(knn_feat_sym_tail_kernel, dim3((rem + 3) / 4, B), dim3(256), lds_s, st, x, quad, dtail, N, C, k,
                       drop, N - rem, idx, tie, dmat);
    return check_launch();
}

// ------------------------------------------------------------------------------------------------
// feature path, bf16 rows (BASELINE configs[3]): inner products on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are
// exact in fp32, accumulation fp32), |x|^2 in fp32 (quad_bf16_kernel), d = ((inner * -2) + |c|^2) + |q|^2 as in the fp32
// path; selection (shared pruning bound, drain loop, tournament merge, lowest index on ties) is the fp32 kernel's.  Rows
// are staged whole (C <= 256) with a pitch of 2C + 16 bytes: an odd number of 16-byte slots, so the 16 lanes of a
// ds_read_b128 group land on 16 different slots; one ds_read_b128 per operand per MFMA.  Not bit-identical with an fp32
// evaluation of the same bf16 rows in principle (the MFMA adds 16 products per step in its own order):
// tests/test_gpu_bf16.py reports the neighbour-set agreement (measured: identical sets on every tested shape).  C % 32 == 0.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void quad_bf16_kernel(const bf16_t* __restrict__ x, long long rows, int C,
                                                        float* __restrict__ quad) {
    // 8 lanes per row, 16-byte loads, fixed-order fold (deterministic)
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int l8 = threadIdx.x & 7;
    float s = 0.f;
    if (row < rows) {
        const bf16_t* r = x + (size_t)row * C;
        for (int c = l8 * 8; c < C; c += 64) {
            const uint4 v = *reinterpret_cast<const uint4*>(r + c);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __uint_as_float(w[e] << 16), b = __uint_as_float(w[e] & 0xffff0000u);
                s = __fmaf_rn(a, a, s);
                s = __fmaf_rn(b, b, s);
            }
        }
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (row < rows && l8 == 0) quad[row] = s;
}

// 128 queries per workgroup: wave w owns queries 32w .. 32w+31 against EVERY candidate tile; the 32-row candidate tiles
// are staged once per workgroup (double-buffered LDS, one barrier per tile) and shared by the four waves.  Against one
// 32-query workgroup per candidate sweep this reads the cloud from L2 a quarter as often (measured round 2: that kernel
// waited on memory 62 % of its wave-cycles, 8.6 GB of L2 requests per launch at B=64 N=4096) and keeps TWO lists per
// query (the two row halves of a tile) instead of eight, so ~3x fewer list insertions in total.
// LDS: 128 x (2C+16) query rows + 2 x 32 x (2C+16) candidate rows + 4 x 4 KB selection scratch + bounds.
template <int K1>
__global__ __launch_bounds__(256, 2) void knn_feat_bf16_kernel(const bf16_t* __restrict__ x,
                                                               const float* __restrict__ quad, int N, int C, int k,
                                                               int drop, int32_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
    const int RP = C * 2 + 16;                                  // row pitch (bytes): (C/8 + 1) sixteen-byte slots, odd
    char* qtile = smem;                                         // 128 rows
    char* ctile = qtile + 128 * RP;                             // 2 x 32 rows
    float* stash_all = reinterpret_cast<float*>(ctile + 2 * 32 * RP);      // 4 waves x 16 x 64
    float* qsm = stash_all + 4 * 16 * 64;                       // 2 x 32: |c|^2 of the staged tiles (+inf past N)
    float* pub = qsm + 2 * 32;                                  // [128 queries][2 lists]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    float* stash = stash_all + wave * 16 * 64;
    constexpr bool SHARE = K1 >= 2;
    constexpr int A0 = (K1 + 1) / 2, A1 = K1 / 2;               // entries the two lists of a query vouch for: A0 + A1 = K1
    pub[tid] = INFINITY;
    const int b = blockIdx.y;
    const int q0 = (int)blockIdx.x * 128;
    const bf16_t* xb = x + (size_t)b * N * C;
    const float* quadb = quad + (size_t)b * N;
    const int cpr = C >> 3;                                     // 16-byte pieces per row
    for (int e = tid; e < 128 * cpr; e += 256) {
        const int row = e / cpr, ch = e - row * cpr;
        *reinterpret_cast<uint4*>(qtile + row * RP + ch * 16) =
            *reinterpret_cast<const uint4*>(xb + (size_t)min(q0 + row, N - 1) * C + ch * 8);
    }
    const int q = q0 + wave * 32 + col;
    const float qq = quadb[min(q, N - 1)];
    TopList<K1> top;
    top.init();
    const int ntiles = (N + 31) >> 5;
    const int npc = (32 * cpr + 255) / 256;                     // staging pieces per thread and tile (<= 4: C <= 256)
    uint4 pre[4];
    float preq = 0.f;
    auto fetch = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const int row = e / cpr, ch = e - row * cpr;
            if (j < npc) pre[j] = *reinterpret_cast<const uint4*>(xb + (size_t)min(tile * 32 + min(row, 31), N - 1) * C + ch * 8);
        }
        if (tid < 32) preq = tile * 32 + tid < N ? quadb[tile * 32 + tid] : INFINITY;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const int row = e / cpr, ch = e - row * cpr;
            if (j < npc && row < 32) *reinterpret_cast<uint4*>(ctile + (buf * 32 + row) * RP + ch * 16) = pre[j];
        }
        if (tid < 32) qsm[buf * 32 + tid] = preq;
    };
    fetch(0);
    stage(0);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ksteps = C >> 4;
    const char* brow = qtile + (wave * 32 + col) * RP + h * 16;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) fetch(tile + 1);                  // in flight under this tile's MFMAs and selection
        const char* arow = ctile + (buf * 32 + col) * RP + h * 16;
        for (int s_ = 0; s_ < ksteps; ++s_) {
            const uint4 a = *reinterpret_cast<const uint4*>(arow + s_ * 32);
            const uint4 bq = *reinterpret_cast<const uint4*>(brow + s_ * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq), acc, 0, 0, 0);
        }
        float thr = top.d[K1 - 1];
        if (SHARE) thr = fminf(thr, fmaxf(pub[(wave * 32 + col) * 2], pub[(wave * 32 + col) * 2 + 1]));
        unsigned m = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 qc = *reinterpret_cast<const float4*>(qsm + buf * 32 + 8 * g + 4 * h);
            const float qcv[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float d = add_rn(add_rn(mul_rn(acc[r], -2.0f), qcv[u]), qq);   // +inf past N
                stash[r * 64 + lane] = d;
                m |= d <= thr ? (1u << r) : 0u;
                acc[r] = 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int c0 = tile * 32 + 4 * h;
        bool has = m != 0;
        int r = has ? __builtin_ctz(m) : 0;
        m &= m - 1;
        float v = stash[r * 64 + lane];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (!__any(has)) break;
            const bool has_n = m != 0;
            const int r_n = has_n ? __builtin_ctz(m) : 0;
            m &= m - 1;
            const float v_n = stash[r_n * 64 + lane];
            top.insert_always(has ? v : INFINITY, c0 + (r & 3) + 8 * (r >> 2));
            has = has_n; r = r_n; v = v_n;
        }
        if (SHARE) pub[(wave * 32 + col) * 2 + h] = h == 0 ? top.d[A0 - 1] : top.d[A1 > 0 ? A1 - 1 : 0];
        if (tile + 1 < ntiles) stage(buf ^ 1);                   // the other buffer: last read in the previous iteration
        __syncthreads();
    }
    // merge the two lists of every query
    int2* lists = reinterpret_cast<int2*>(smem);
    top.store(lists + (size_t)tid * K1);
    __syncthreads();
    const int mq = tid >> 1, ml = tid & 1;
    const int src = (mq >> 5) * 64 + ml * 32 + (mq & 31);
    const int oq = q0 + mq;
    const bool ovalid = oq < N;
    merge_write<K1, 2>(lists, src, ml, k, drop, ovalid, idx + ((size_t)b * N + (ovalid ? oq : 0)) * k, N, ovalid ? oq : 0);
}

template <int K1>
static int launch_knn_feat_bf16(const bf16_t* x, const float* quad, int B, int N, int C, int k, int drop, int32_t* idx,
                                hipStream_t st) {
    if (C > 256) return HSP_ERR_UNSUPPORTED;                   // staging registers: 4 pieces per thread and tile
    size_t lds = (size_t)(128 + 64) * (C * 2 + 16) + (size_t)(4 * 16 * 64 + 2 * 32 + 128 * 2) * 4;
    const size_t lds_lists = (size_t)256 * K1 * 8;
    if (lds_lists > lds) lds = lds_lists;
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    auto kern = knn_feat_bf16_kernel<K1>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(kern, dim3((N + 127) / 128, B), dim3(256), lds, st, x, quad, N, C, k, drop, idx);
    return check_launch();
}

// xyz search of csrc/knn_exact.hip: ranks [drop, k + drop) by (distance, index) into idx (B,N,k) and, per row (tie, B*N bytes),
// bit 0: two of the k + drop + 1 nearest hold equal distances; bit 1: two of the k2 + drop + 1 nearest do
int knn3_select_flags(const float* x, int B, int N, int k, int drop, int k2, int32_t* idx, int32_t* idx2, uint8_t* tie,
                      hipStream_t st, bool* needs_tie_pass) {
    // the wave kernel replays its flagged rows itself (tie_inline); the per-lane-list kernel and the dense clouds leave flags
    *needs_tie_pass = !(N >= 64 && (long long)B * N < 131072 && knn3_wave_tie_inline(N) != 0);
    const int m = k + drop;
    const int msel = m + 1 < N ? m + 1 : N;
    const int msel2 = k2 > 0 ? k2 + drop + 1 : 0;              // (k2 < k: inside msel)
    switch (pick_k1(msel)) {
        case 3: return launch_knn3_t<3>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 5: return launch_knn3_t<5>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 6: return launch_knn3_t<6>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 9: return launch_knn3_t<9>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 10: return launch_knn3_t<10>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 17: return launch_knn3_t<17>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 21: return launch_knn3_t<21>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 22: return launch_knn3_t<22>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        case 33: return launch_knn3_t<33>(x, B, N, k, drop, idx, st, msel, tie, msel2, idx2, k2);
        default: return HSP_ERR_UNSUPPORTED;
    }
}

}  // namespace hsp

using namespace hsp;

#ifdef HSP_KNN_PROF
extern "C" int hsp_debug_set_knn_prof(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(hsp::g_knn_prof), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t hsp_knn_workspace_bytes(int B, int N, int C, int k) {
    (void)k;
    if (C == 3 || B <= 0 || N <= 0) return 0;
    // |x|^2 per row; in the symmetric remainder mode also the remainder-query distances d(t, j)
    const int rows_extra = knn_feat_rem_mode(B, N, C) == KF_REM_SYM ? (N & 31) : 0;
    return (size_t)B * N * sizeof(float) * (size_t)(1 + rows_extra);
}

extern "C" int hsp_quad_outer_f32(const float* x, int B, int N, int C, float* quad, hspStream_t stream);

// quad_mode 0: |x|^2 in ATen's order for a CONTIGUOUS (B,N,C) tensor; 1: for the transposed view of a (B,C,N) tensor (exact.hip)
// tie (feature path only, may be null): B*N flag bytes -- the selection runs one rank past the answer and says per row whether two
// of the k + drop + 1 nearest hold equal distances (csrc/knn_exact.hip replays those rows in torch.topk's order)
static int knn_f32_impl(const float* x, int B, int N, int C, int k, int drop_first, int32_t* idx, void* ws,
                        size_t ws_bytes, int quad_mode, hspStream_t stream, uint8_t* tie = nullptr, float* dmat = nullptr) {
    if (!x || !idx || B <= 0 || N <= 0 || C <= 0 || k <= 0) return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    const int m = k + drop;
    if (m > N || k > HSP_MAX_K) return HSP_ERR_BAD_ARG;
    const int msel = tie && C != 3 ? (m + 1 < N ? m + 1 : N) : 0;
    const int K1 = pick_k1(msel > m ? msel : m);
    if (!K1) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define HSP_K1_SWITCH(CALL)                                  \
    switch (K1) {                                            \
        case 3: return CALL(3);                              \
        case 5: return CALL(5);                              \
        case 6: return CALL(6);                              \
        case 9: return CALL(9);                              \
        case 10: return CALL(10);                            \
        case 17: return CALL(17);                            \
        case 21: return CALL(21);                            \
        case 22: return CALL(22);                            \
        default: return CALL(33);                            \
    }
    if (C == 3) {
#define CALL3(K) launch_knn3_t<K>(x, B, N, k, drop, idx, st)
        HSP_K1_SWITCH(CALL3)
#undef CALL3
    }
    if (hsp_knn_workspace_bytes(B, N, C, k) > ws_bytes || !ws) return HSP_ERR_WORKSPACE;
    float* quad = reinterpret_cast<float*>(ws);
    const long long rows = (long long)B * N;
    int rc;
    const int small_qt = knn_feat_small_qt(B, N, C);
    if (small_qt) {
        // the coarse levels' searches: one launch, |x|^2 inside it (contiguous rows; the transposed view's sum order keeps its own
        // kernel).  The exact search's replay pass reads |x|^2 from ws: the kernel leaves it there when flags are asked for.
        const float* qin = nullptr;
        if (quad_mode == 1) {
            rc = hsp_quad_outer_f32(x, B, N, C, quad, stream);
            if (rc) return rc;
            qin = quad;
        }
        float* qout = (tie && !qin) ? quad : nullptr;
        return launch_knn_feat_small(small_qt, x, qin, qout, B, N, C, k, drop, idx, st, tie, dmat);
    }
    if (quad_mode == 1) rc = hsp_quad_outer_f32(x, B, N, C, quad, stream);
    else {
        if (C >= 8 && C < 512 && (C & 7) == 0)
            hipLaunchKernelGGL(quad32_kernel, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, st, x, rows, C, quad);
        else
            hipLaunchKernelGGL(quad_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, rows, C, quad);
        rc = check_launch();
    }
    if (rc) return rc;
#define CALLF(K) launch_knn_feat<K>(x, quad, quad + rows, B, N, C, k, drop, idx, st, msel, tie, dmat)
    HSP_K1_SWITCH(CALLF)
#undef CALLF
#undef HSP_K1_SWITCH
}

extern "C" int hsp_knn_f32(const float* x, int B, int N, int C, int k, int drop_first, int32_t* idx, void* ws,
                           size_t ws_bytes, hspStream_t stream) {
    return knn_f32_impl(x, B, N, C, k, drop_first, idx, ws, ws_bytes, 0, stream);
}
extern "C" int hsp_knn_quadmode_f32(const float* x, int B, int N, int C, int k, int drop_first, int32_t* idx, void* ws,
                                    size_t ws_bytes, int quad_mode, hspStream_t stream) {
    return knn_f32_impl(x, B, N, C, k, drop_first, idx, ws, ws_bytes, quad_mode, stream);
}

namespace hsp {
// knn_exact.hip: the feature-space search by (distance, index) + a flag byte per row (ws as hsp_knn_workspace_bytes: |x|^2 first)
int knn_feat_select_flags(const float* x, int B, int N, int C, int k, int drop, int quad_mode, int32_t* idx, void* ws, size_t ws_bytes,
                          uint8_t* tie, float* dmat, hspStream_t stream) {
    return knn_f32_impl(x, B, N, C, k, drop, idx, ws, ws_bytes, quad_mode, stream, tie, dmat);
}
}  // namespace hsp

extern "C" int hsp_knn_bf16(const hsp_bf16_t* x, int B, int N, int C, int k, int drop_first, int32_t* idx, void* ws,
                            size_t ws_bytes, hspStream_t stream) {
    if (!x || !idx || B <= 0 || N <= 0 || C <= 0 || k <= 0) return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    const int m = k + drop;
    if (m > N || k > HSP_MAX_K) return HSP_ERR_BAD_ARG;
    if ((C & 31) || (reinterpret_cast<size_t>(x) & 15)) return HSP_ERR_UNSUPPORTED;
    const int K1 = pick_k1(m);
    if (!K1) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)B * N * sizeof(float)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    float* quad = reinterpret_cast<float*>(ws);
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(quad_bf16_kernel, dim3((unsigned)((rows * 8 + 255) / 256)), dim3(256), 0, st, x, rows, C, quad);
    int rc = check_launch();
    if (rc) return rc;
    switch (K1 == 6 ? 9 : K1 == 10 ? 17 : K1 == 22 ? 33 : K1) {
        case 3: return launch_knn_feat_bf16<3>(x, quad, B, N, C, k, drop, idx, st);
        case 5: return launch_knn_feat_bf16<5>(x, quad, B, N, C, k, drop, idx, st);
        case 9: return launch_knn_feat_bf16<9>(x, quad, B, N, C, k, drop, idx, st);
        case 17: return launch_knn_feat_bf16<17>(x, quad, B, N, C, k, drop, idx, st);
        case 21: return launch_knn_feat_bf16<21>(x, quad, B, N, C, k, drop, idx, st);
        default: return launch_knn_feat_bf16<33>(x, quad, B, N, C, k, drop, idx, st);
    }
}

static int geometry_impl(const float* xyz, int B, int N0, const int32_t* sel1, int N1, const int32_t* sel2, int N2, int k1, int kpool,
                         int k2, int drop, float* v1, float* v2, int32_t* idx1, int32_t* idx1_pool, int32_t* idx2, int32_t* up1,
                         int32_t* up2, const uint8_t* tie0, int k0, int kpool0, int32_t* idx0, int32_t* idx0_pool, hipStream_t st) {
    GeoArgs a;
    a.xyz = xyz; a.N0 = N0; a.sel1 = sel1; a.N1 = N1; a.sel2 = sel2; a.N2 = N2; a.k1 = k1; a.kpool = kpool; a.k2 = k2; a.drop = drop;
    a.v1 = v1; a.v2 = v2; a.idx1 = idx1; a.idx1p = idx1_pool; a.idx2 = idx2; a.up1 = up1; a.up2 = up2;
    a.nb1 = (N1 + 3) / 4; a.nb2 = (N2 + 3) / 4; a.nbt = (N0 + 255) / 256;
    a.tie0 = tie0; a.B = B; a.k0 = k0; a.kpool0 = kpool0; a.idx0 = idx0; a.idx0p = idx0_pool;
    a.nbtie = tie0 ? (int)(((long long)B * N0 + 31) / 32) : 0;  // one tie wave per 8 rows of the level-0 search
    const int Nm = N1 > N2 ? N1 : N2;
    size_t lds = (size_t)Nm * 16 + (size_t)4 * KNN3W_SV * 8 + 4 * 64 * 4 + (size_t)16 * Nm * 4;
    if (tie0 && lds < (size_t)4 * 16 * N0) lds = (size_t)4 * 16 * N0;
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    const dim3 grid(a.nbtie + (a.nb1 + a.nb2 + 2 * a.nbt) * B);
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        return check_launch();
    };
    return Nm <= 64 * 5 ? launch(geometry_levels_kernel<5>) : launch(geometry_levels_kernel<9>);
}

static bool geometry_levels_ok(int N0, int N1, int N2, int k1, int k2, int drop) {
    // the wave-per-query search with its in-kernel tie replay: 64 <= N <= 576 points per level, lists of at most 31 + drop entries
    return !(N1 < 64 || N1 > 64 * 9 || N2 < 64 || N2 > 64 * 9 || k1 + drop + 1 > 33 || k2 + drop + 1 > 33 || k1 + drop > N1 ||
             k2 + drop > N2 || N2 > N1 || N1 > N0);
}

extern "C" int hsp_geometry_levels_f32(const float* xyz, int B, int N0, const int32_t* sel1, int N1, const int32_t* sel2, int N2,
                                       int k1, int kpool, int k2, int drop_first, float* v1, float* v2, int32_t* idx1,
                                       int32_t* idx1_pool, int32_t* idx2, int32_t* up1, int32_t* up2, hspStream_t stream) {
    if (!xyz || !sel1 || !sel2 || !v1 || !v2 || !idx1 || !idx2 || !up1 || !up2 || B <= 0 || N0 <= 0 || k1 <= 0 || k2 <= 0 || kpool < 0 ||
        (kpool > 0) != (idx1_pool != nullptr) || kpool > k1)
        return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    if (!geometry_levels_ok(N0, N1, N2, k1, k2, drop)) return HSP_ERR_UNSUPPORTED;
    return geometry_impl(xyz, B, N0, sel1, N1, sel2, N2, k1, kpool, k2, drop, v1, v2, idx1, idx1_pool, idx2, up1, up2, nullptr, 0, 0,
                         nullptr, nullptr, as_stream(stream));
}

extern "C" size_t hsp_geometry_all_workspace_bytes(int B, int N0) {
    if (B <= 0 || N0 <= 0) return 0;
    return ((size_t)B * N0 + 255) & ~(size_t)255;
}

extern "C" int hsp_geometry_all_f32(const float* xyz, int B, int N0, int k0, int kpool0, const int32_t* sel1, int N1,
                                    const int32_t* sel2, int N2, int k1, int kpool, int k2, int drop_first, int32_t* idx0,
                                    int32_t* idx0_pool, float* v1, float* v2, int32_t* idx1, int32_t* idx1_pool, int32_t* idx2,
                                    int32_t* up1, int32_t* up2, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!xyz || !sel1 || !sel2 || !v1 || !v2 || !idx0 || !idx1 || !idx2 || !up1 || !up2 || B <= 0 || N0 <= 0 || k0 <= 0 || k1 <= 0 ||
        k2 <= 0 || kpool < 0 || kpool0 < 0 || (kpool > 0) != (idx1_pool != nullptr) || (kpool0 > 0) != (idx0_pool != nullptr) ||
        kpool > k1 || kpool0 > k0)
        return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    // level 0 on the wave search WITHOUT the in-kernel replay (its tie pass rides here): 576 < N0 <= 1088, a batch below 131072 rows
    if (!geometry_levels_ok(N0, N1, N2, k1, k2, drop) || N0 <= 64 * 9 || N0 > 64 * 17 || (long long)B * N0 >= 131072 ||
        k0 + drop + 1 > 33 || k0 + drop > N0)
        return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_geometry_all_workspace_bytes(B, N0)) return HSP_ERR_WORKSPACE;
    uint8_t* tie = reinterpret_cast<uint8_t*>(ws);
    bool needs_pass = true;
    int rc = knn3_select_flags(xyz, B, N0, k0, drop, kpool0, idx0, idx0_pool, tie, as_stream(stream), &needs_pass);
    if (rc) return rc;
    if (!needs_pass) return HSP_ERR_UNSUPPORTED;               // (cannot happen inside the range above)
    return geometry_impl(xyz, B, N0, sel1, N1, sel2, N2, k1, kpool, k2, drop, v1, v2, idx1, idx1_pool, idx2, up1, up2, tie, k0, kpool0,
                         idx0, idx0_pool, as_stream(stream));
}

extern "C" int hsp_nn1_f32(const float* tgt, int Nt, const float* src, int Ns, int B, int32_t* idx,
                           hspStream_t stream) {
    if (!tgt || !src || !idx || B <= 0 || Nt <= 0 || Ns <= 0) return HSP_ERR_BAD_ARG;
    const size_t lds = (size_t)Ns * 16;
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nn1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(nn1_kernel, dim3((Nt + 255) / 256, B), dim3(256), lds, as_stream(stream), tgt, Nt, src, Ns, idx);
    return check_launch();
}
