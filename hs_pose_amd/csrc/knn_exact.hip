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

// one wave per query row.  cand (B,N,mc): the mc = min(m + 1, N) nearest by (distance, index) from hsp_knn_f32 (no drop).
__global__ __launch_bounds__(64) void knn_ties_kernel(const float* __restrict__ x, const float* __restrict__ quad,
                                                      const int32_t* __restrict__ cand, int N, int C, int k, int drop, int mc,
                                                      int32_t* __restrict__ idx, int* __restrict__ nties) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TkE* q = reinterpret_cast<TkE*>(smem);                     // N entries (tie rows only)
    int* LA = reinterpret_cast<int*>(q + N);                   // N: partition scratch
    int* LB = LA + N;
    const int lane = threadIdx.x;
    const int b = blockIdx.y, i = blockIdx.x;
    const int m = k + drop;
    const float* xb = x + (size_t)b * N * C;
    const float* xi = xb + (size_t)i * C;
    const float* qb = quad ? quad + (size_t)b * N : nullptr;
    auto quad_of = [&](int j) {
        if (qb) return qb[j];
        const float* p = xb + (size_t)j * 3;
        return quad3(p[0], p[1], p[2]);
    };
    auto dist_to = [&](int j, float qi) {
        const float* xj = xb + (size_t)j * C;
        float acc = 0.f;
        if ((C & 3) == 0) {
            for (int c = 0; c < C; c += 4) {                                 // torch.bmm: k-ordered chain from 0
                const float4 a = *reinterpret_cast<const float4*>(xi + c), bq = *reinterpret_cast<const float4*>(xj + c);
                acc = __fmaf_rn(a.x, bq.x, acc); acc = __fmaf_rn(a.y, bq.y, acc);
                acc = __fmaf_rn(a.z, bq.z, acc); acc = __fmaf_rn(a.w, bq.w, acc);
            }
        } else {
            for (int c = 0; c < C; ++c) acc = __fmaf_rn(xi[c], xj[c], acc);
        }
        return add_rn(add_rn(mul_rn(acc, -2.0f), quad_of(j)), qi);           // gcn3d.py:21, left to right
    };
    const float qi = quad_of(i);
    const int32_t* cr = cand + ((size_t)b * N + i) * mc;
    const int cj = lane < mc ? cr[lane] : 0;
    const float dc = lane < mc ? dist_to(cj, qi) : INFINITY;
    const float dn = __shfl_down(dc, 1);
    const bool tie = lane + 1 < mc && dc == dn;
    int32_t* out = idx + ((size_t)b * N + i) * k;
    if (__ballot(tie) == 0ull) {
        if (lane >= drop && lane < m) out[lane - drop] = cj;
        return;
    }
    for (int j = lane; j < N; j += 64) { q[j].v = dist_to(j, qi); q[j].i = j; }
    __syncthreads();
    if (lane == 0 && nties) atomicAdd(nties, 1);
    const LaneAcc H = tkw_topk(q, LA, LB, m, N, lane);
    if (lane >= drop && lane < m) out[lane - drop] = H.i;
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
    bool needs_pass = true;
    int rc = knn3_select_flags(xyz, B, N, k, drop, k2, idx, idx2, tie, as_stream(stream), &needs_pass);
    if (rc || !needs_pass) return rc;                      // (tie_rows is only counted by the separate pass)
    const size_t lds = (size_t)N * (sizeof(TkE) + 2 * sizeof(int));
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
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

extern "C" size_t hsp_knn_exact_workspace_bytes(int B, int N, int C, int k, int drop_first) {
    if (B <= 0 || N <= 0 || C <= 0 || k <= 0) return 0;
    const int m = k + (drop_first ? 1 : 0);
    const int mc = m + 1 < N ? m + 1 : N;
    const size_t inner = hsp_knn_workspace_bytes(B, N, C, mc);
    return ((inner + 255) & ~(size_t)255) + (size_t)B * N * mc * sizeof(int32_t) + 256;
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
    int32_t* cand = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + inner);
    int rc = hsp_knn_quadmode_f32(x, B, N, C, mc, 0, cand, ws, inner, C == 3 ? 0 : quad_mode, stream);
    if (rc) return rc;
    const size_t lds = (size_t)N * (sizeof(TkE) + 2 * sizeof(int));
    if (lds > 160 * 1024) return HSP_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_ties_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const float* quad = C == 3 ? nullptr : reinterpret_cast<const float*>(ws);      // hsp_knn_f32 left |x|^2 there
    hipLaunchKernelGGL(knn_ties_kernel, dim3(N, B), dim3(64), lds, as_stream(stream), x, quad, cand, N, C, k, drop, mc, idx, tie_rows);
    return check_launch();
}
