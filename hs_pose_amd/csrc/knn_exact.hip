// knn_exact.hip -- get_neighbor_index (gcn3d.py:15-24) INCLUDING torch.topk's order among exactly equal distances.
//
// The expanded fp32 distance of post-ReLU feature rows is coarse (|x|^2 ~ 10^2 against d ~ 10^-1: steps of 3e-5), so exact
// ties are common on real activations: on the reference-initialised stack 1.8 % of the rows of conv_1's neighbour search hold
// a tie among their 21 nearest or at the boundary.  hsp_knn_f32 breaks ties by the lowest index; ATen's CPU topk
// (aten/src/ATen/native/cpu/TopKImpl.h) breaks them by whatever libstdc++'s std::nth_element + std::sort (m * 64 > N) or
// std::partial_sort (m * 64 <= N) do with a comparator that only looks at the value -- deterministic, and restated here
// statement by statement (as in oracle/hsp_oracle.c, which is pinned against torch.topk itself on tie-rich rows:
// tests/golden/exact_topk_ties.npz).  Eval-mode forward only (ops.exact_scope): with the feature rows already carrying the
// reference's bits (exact.hip, gemm_wave.hip) this makes every neighbour list the reference's list.
//
//   1. hsp_knn_f32 selects the m + 1 nearest by (distance, index)                         (m = k + drop_first)
//   2. knn_ties_kernel, one wave per query: the m + 1 candidates' distances again (same arithmetic: k-ordered fma chain,
//      ((inner * -2) + |c|^2) + |q|^2); no two equal neighbours in that sorted list -> the selection is unique, copy it;
//      otherwise all N distances of the row go to LDS and the wave runs libstdc++'s algorithm on them: the partition passes
//      of nth_element as ballot sweeps (tkw_partition_pivot), the short tails (median of three, insertion sort, the final
//      std::sort of m - 1 entries, the heap forms) by one lane.
#include "common.h"
#include <stdlib.h>
#include "tie_pass.h"

namespace hsp {

// ---- feature rows (C != 3): the tie pass over FLAGGED rows only ------------------------------------------------------------------
// knn_feat_select_flags has written every row's (distance, index)-ordered list and a flag per row (two of the k + drop + 1 nearest
// equally far).  A fixed grid of single-wave workgroups reads the flags 64 rows at a time; a flagged row gets its N distances again
// -- the k-ordered fma chain of torch.bmm, ((inner * -2) + |c|^2) + |q|^2, four candidate rows per lane in flight, the query row
// broadcast from LDS -- and libstdc++'s algorithm (tie_pass.h).  (Round 4 visited EVERY row with one wave that recomputed its 22
// candidates' distances to find out: 31 us per call whatever the data held.)
__global__ __launch_bounds__(64) void knn_feat_ties_rows_kernel(const float* __restrict__ x, const float* __restrict__ quad,
                                                                const uint8_t* __restrict__ tie, const float* __restrict__ dmat, int B,
                                                                int N, int C, int k, int drop, int32_t* __restrict__ idx,
                                                                int* __restrict__ nties) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TkE* q = reinterpret_cast<TkE*>(smem);                     // N entries
    int* LA = reinterpret_cast<int*>(q + N);                   // N: partition scratch
    int* LB = LA + N;
    float* sq = reinterpret_cast<float*>(LB + N);              // the query row (C floats, 16-byte aligned: N * 16 bytes precede it)
    const int lane = threadIdx.x;
    const int rows = B * N, G = gridDim.x, w = blockIdx.x;
    const int m = k + drop;
    for (int base = 0; base < rows; base += 64 * G) {
        const int row_l = base + lane * G + w;
        const int fl = row_l < rows ? (int)tie[row_l] : 0;
        unsigned long long todo = __ballot(fl != 0);
        while (todo) {
            const int tl = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int row = base + tl * G + w;
            const int b = row / N, i = row - b * N;
            const float* xb = x + (size_t)b * N * C;
            const float* qb = quad + (size_t)b * N;
            for (int c = lane; c < C; c += 64) sq[c] = xb[(size_t)i * C + c];
            __builtin_amdgcn_wave_barrier();
            const float qi = qb[i];
            if (lane == 0 && nties) atomicAdd(nties, 1);
            if (dmat) {
                // the selection kernel left the row's distances (as [candidate][query]: a strided column, every load in flight at
                // once) -- recomputing them is ~0.5 MB of feature rows streamed by ONE wave, ~30 us of the 40 a flagged row cost
                const float* col = dmat + (size_t)b * N * N + i;
                for (int j0 = lane; j0 < N; j0 += 8 * 64) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(j0 + 64 * u < N ? j0 + 64 * u : N - 1) * N];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + 64 * u;
                        if (j < N) { TkE e; e.v = fminf(v[u], 3.402823466e+38f); e.i = j; q[j] = e; }
                    }
                }
            } else
            for (int j0 = lane; j0 < N; j0 += 4 * 64) {
                const float* r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = xb + (size_t)(j0 + 64 * u < N ? j0 + 64 * u : N - 1) * C;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if ((C & 15) == 0) {
                    for (int c = 0; c < C; c += 16) {                    // torch.bmm: k-ordered chain from 0; 16 loads in flight per lane
                        float4 v[4][4];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[t][u] = *reinterpret_cast<const float4*>(r[u] + c + 4 * t);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float4 a = *reinterpret_cast<const float4*>(sq + c + 4 * t);
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                acc[u] = __fmaf_rn(a.w, v[t][u].w, __fmaf_rn(a.z, v[t][u].z, __fmaf_rn(a.y, v[t][u].y, __fmaf_rn(a.x, v[t][u].x, acc[u]))));
                        }
                    }
                } else if ((C & 3) == 0) {
                    for (int c = 0; c < C; c += 4) {
                        const float4 a = *reinterpret_cast<const float4*>(sq + c);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 v = *reinterpret_cast<const float4*>(r[u] + c);
                            acc[u] = __fmaf_rn(a.w, v.w, __fmaf_rn(a.z, v.z, __fmaf_rn(a.y, v.y, __fmaf_rn(a.x, v.x, acc[u]))));
                        }
                    }
                } else {
                    for (int c = 0; c < C; ++c) {
                        const float a = sq[c];
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] = __fmaf_rn(a, r[u][c], acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 64 * u;
                    if (j < N) {
                        TkE e;                                           // gcn3d.py:21, left to right; NaN / +inf -> FLT_MAX (a total order)
                        e.v = fminf(add_rn(add_rn(mul_rn(acc[u], -2.0f), qb[j]), qi), 3.402823466e+38f);
                        e.i = j;
                        q[j] = e;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const LaneAcc H = tkw_topk(q, LA, LB, m, N, lane);
            int32_t* out = idx + (size_t)row * k;
            if (lane >= drop && lane < m) out[lane - drop] = H.i;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- coordinates (C == 3): the tie pass over FLAGGED rows only -------------------------------------------------------------------
// knn3_select_flags has written every row's (distance, index)-ordered list and a flag per row; rows whose k + drop + 1 nearest are
// pairwise different are final (torch.topk's answer is unique there, and the k2-list is the prefix of the k-list).  A fixed grid of
// single-wave workgroups walks the rows (hipGraph-friendly: the launch does not depend on how many rows are flagged; a tie-free
// batch costs one byte read per row) and replays flagged rows through libstdc++'s algorithm, once per requested list length --
// on a tiled cloud (datasets/load_data.py:314-316) that is every row, and the k = 4 list of Pool_layer (gcn3d.py:236: partial_sort
// at N = 1028) is NOT the prefix of the layers' k = 20 list (nth_element + sort).
__global__ __launch_bounds__(64) void knn_xyz_ties_kernel(const float* __restrict__ x, const uint8_t* __restrict__ tie, int B, int N,
                                                          int k, int k2, int drop, int32_t* __restrict__ idx,
                                                          int32_t* __restrict__ idx2, int* __restrict__ nties) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tie_rows_xyz(smem, x, tie, B, N, k, k2, drop, idx, idx2, nties, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}

}  // namespace hsp

using namespace hsp;

#ifdef HSP_TIE_PROF
extern "C" int hsp_debug_set_tie_prof(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tie_prof), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t hsp_knn_xyz_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return ((size_t)B * N + 255) & ~(size_t)255;
}

extern "C" int hsp_knn_xyz_f32(const float* xyz, int B, int N, int k, int k2, int drop_first, int32_t* idx, int32_t* idx2, void* ws,
                               size_t ws_bytes, int* tie_rows, hspStream_t stream) {
    if (!xyz || !idx || B <= 0 || N <= 0 || k <= 0 || k2 < 0 || k2 > k || (k2 > 0) != (idx2 != nullptr)) return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    const int m = k + drop;
    if (m > N || k > HSP_MAX_K) return HSP_ERR_BAD_ARG;
    if (m + 1 > 33) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_knn_xyz_workspace_bytes(B, N)) return HSP_ERR_WORKSPACE;
    uint8_t* tie = reinterpret_cast<uint8_t*>(ws);
    // the tie pass keeps a whole row of candidates in LDS: clouds beyond 10 240 points are refused BEFORE anything is launched (the
    // caller then takes hsp_knn_f32's (distance, index) order, which has no such bound -- ops.knn_xyz does)
    const size_t lds = (size_t)N * (sizeof(TkE) + 2 * sizeof(int));
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    bool needs_pass = true;
    int rc = knn3_select_flags(xyz, B, N, k, drop, k2, idx, idx2, tie, as_stream(stream), &needs_pass);
    if (rc || !needs_pass) return rc;                      // (tie_rows is only counted by the separate pass)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_xyz_ties_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const long long rows = (long long)B * N;
    const int per_cu = (int)(160 * 1024 / (lds > 16 * 1024 ? lds : 16 * 1024));      // resident single-wave workgroups per CU
    const long long cap = (long long)HSP_NUM_CU * (per_cu < 1 ? 1 : per_cu);
    // one wave per 8 rows at most: a tie-free batch then dispatches a few hundred workgroups, not thousands (6 us -> 3 us at N = 257)
    const long long want = (rows + 7) / 8;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(knn_xyz_ties_kernel, dim3(grid), dim3(64), lds, as_stream(stream), xyz, tie, B, N, k, k2, drop, idx, idx2,
                       tie_rows);
    return check_launch();
}

// the selection kernel leaves the (B, N, N) distances for the tie pass while that is a small buffer (an image's instances at
// N = 1028: 4 MB each); beyond 512 MB a flagged row's distances are recomputed
// (64 MB until round 6: at B = 16, N = 1028 -- 67.6 MB, the exact_train step -- the replay then recomputed 1028 dot products per
// flagged row: 2.127 -> 2.015 ms per training step with the matrix kept)
#ifndef HSP_KNN_DMAT_MB
#define HSP_KNN_DMAT_MB 512
#endif
static bool knn_exact_keeps_distances(int B, int N) { return (size_t)B * N * N * sizeof(float) <= ((size_t)HSP_KNN_DMAT_MB << 20); }

extern "C" size_t hsp_knn_exact_workspace_bytes(int B, int N, int C, int k, int drop_first) {
    if (B <= 0 || N <= 0 || C <= 0 || k <= 0) return 0;
    const int m = k + (drop_first ? 1 : 0);
    const int mc = m + 1 < N ? m + 1 : N;
    const size_t inner = hsp_knn_workspace_bytes(B, N, C, mc);
    size_t bytes = ((inner + 255) & ~(size_t)255) + (((size_t)B * N + 255) & ~(size_t)255) + 256;   // |x|^2 (+ remainder rows), row flags
    if (C != 3 && knn_exact_keeps_distances(B, N)) bytes += (size_t)B * N * N * sizeof(float);       // + the distance matrix
    return bytes;
}

extern "C" int hsp_knn_quadmode_f32(const float* x, int B, int N, int C, int k, int drop_first, int32_t* idx, void* ws,
                                    size_t ws_bytes, int quad_mode, hspStream_t stream);

extern "C" int hsp_knn_exact_f32(const float* x, int B, int N, int C, int k, int drop_first, int quad_mode, int32_t* idx, void* ws,
                                 size_t ws_bytes, int* tie_rows, hspStream_t stream) {
    if (!x || !idx || B <= 0 || N <= 0 || C <= 0 || k <= 0) return HSP_ERR_BAD_ARG;
    const int drop = drop_first ? 1 : 0;
    const int m = k + drop;
    if (m > N) return HSP_ERR_BAD_ARG;
    const int mc = m + 1 < N ? m + 1 : N;
    if (mc > 33 || mc > 64) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_knn_exact_workspace_bytes(B, N, C, k, drop_first)) return HSP_ERR_WORKSPACE;
    if (C == 3) return hsp_knn_xyz_f32(x, B, N, k, 0, drop_first, idx, nullptr, ws, ws_bytes, tie_rows, stream);   // flags + flagged rows
    const size_t inner = (hsp_knn_workspace_bytes(B, N, C, mc) + 255) & ~(size_t)255;
    uint8_t* tie = reinterpret_cast<uint8_t*>(ws) + inner;
    float* dmat = knn_exact_keeps_distances(B, N)
                      ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + inner + ((((size_t)B * N + 255) & ~(size_t)255) + 256))
                      : nullptr;
    int rc = knn_feat_select_flags(x, B, N, C, k, drop, quad_mode, idx, ws, inner, tie, dmat, stream);
    if (rc) return rc;
    const size_t lds = (size_t)N * (sizeof(TkE) + 2 * sizeof(int)) + (size_t)((C + 3) & ~3) * sizeof(float);
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_feat_ties_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const float* quad = reinterpret_cast<const float*>(ws);                           // knn_feat_select_flags left |x|^2 there
    const long long rows = (long long)B * N;
    const int per_cu = (int)(160 * 1024 / (lds > 16 * 1024 ? lds : 16 * 1024));
    // (as many waves as fit: ~2 % of the rows of real activations are flagged, and two of them in one wave double the pass)
    const long long cap = (long long)HSP_NUM_CU * (per_cu < 1 ? 1 : per_cu);
    const int grid = (int)(rows < cap ? rows : cap);
    hipLaunchKernelGGL(knn_feat_ties_rows_kernel, dim3(grid), dim3(64), lds, as_stream(stream), x, quad, tie, dmat, B, N, C, k, drop, idx,
                       tie_rows);
    return check_launch();
}
