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

namespace hsp {

struct TkE { float v; int i; };
#define TKD_LT(a, b) ((a).v < (b).v)

__device__ __forceinline__ void tkd_swap(TkE* a, TkE* b) { const TkE t = *a; *a = *b; *b = t; }
__device__ __forceinline__ int tkd_lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

__device__ void tkd_move_median_to_first(TkE* result, TkE* a, TkE* b, TkE* c) {
    if (TKD_LT(*a, *b)) {
        if (TKD_LT(*b, *c)) tkd_swap(result, b);
        else if (TKD_LT(*a, *c)) tkd_swap(result, c);
        else tkd_swap(result, a);
    } else if (TKD_LT(*a, *c)) tkd_swap(result, a);
    else if (TKD_LT(*b, *c)) tkd_swap(result, c);
    else tkd_swap(result, b);
}
__device__ TkE* tkd_partition_pivot(TkE* first, TkE* last) {
    TkE* mid = first + (last - first) / 2;
    tkd_move_median_to_first(first, first + 1, mid, last - 1);
    TkE* pivot = first;
    ++first;
    const float pv = pivot->v;                    // (the pivot slot is never swapped inside the loop)
    for (;;) {
        while (first->v < pv) ++first;
        --last;
        while (pv < last->v) --last;
        if (!(first < last)) return first;
        tkd_swap(first, last);
        ++first;
    }
}
__device__ void tkd_unguarded_linear_insert(TkE* last) {
    const TkE val = *last;
    TkE* next = last - 1;
    while (TKD_LT(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
__device__ void tkd_insertion_sort(TkE* first, TkE* last) {
    if (first == last) return;
    for (TkE* i = first + 1; i != last; ++i) {
        if (TKD_LT(*i, *first)) {
            const TkE val = *i;
            for (TkE* p = i; p != first; --p) *p = *(p - 1);          // move_backward(first, i, i + 1)
            *first = val;
        } else tkd_unguarded_linear_insert(i);
    }
}
__device__ void tkd_push_heap(TkE* first, int hole, int top, TkE value) {
    int parent = (hole - 1) / 2;
    while (hole > top && TKD_LT(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
__device__ void tkd_adjust_heap(TkE* first, int hole, int len, TkE value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (TKD_LT(first[child], first[child - 1])) --child;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    tkd_push_heap(first, hole, top, value);
}
__device__ void tkd_make_heap(TkE* first, TkE* last) {
    const int len = (int)(last - first);
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        tkd_adjust_heap(first, parent, len, first[parent]);
        if (parent == 0) return;
    }
}
__device__ void tkd_pop_heap(TkE* first, TkE* last, TkE* result) {
    const TkE value = *result;
    *result = *first;
    tkd_adjust_heap(first, 0, (int)(last - first), value);
}
__device__ void tkd_heap_select(TkE* first, TkE* middle, TkE* last) {
    tkd_make_heap(first, middle);
    for (TkE* i = middle; i < last; ++i)
        if (TKD_LT(*i, *first)) tkd_pop_heap(first, middle, i);
}
__device__ void tkd_sort_heap(TkE* first, TkE* last) {
    while (last - first > 1) { --last; tkd_pop_heap(first, last, last); }
}
__device__ void tkd_introselect(TkE* first, TkE* nth, TkE* last, int depth_limit) {
    while (last - first > 3) {
        if (depth_limit == 0) { tkd_heap_select(first, nth + 1, last); tkd_swap(first, nth); return; }
        --depth_limit;
        TkE* cut = tkd_partition_pivot(first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    tkd_insertion_sort(first, last);
}
// std::sort of a short range: __introsort_loop (recursion on the upper part turned into a small explicit stack) + final insertion sort
__device__ void tkd_sort(TkE* first, TkE* last) {
    if (first == last) return;
    struct { TkE* f; TkE* l; int d; } st[40];
    int sp = 0;
    st[sp].f = first; st[sp].l = last; st[sp].d = 2 * tkd_lg((int)(last - first)); ++sp;
    while (sp) {
        --sp;
        TkE* f = st[sp].f; TkE* l = st[sp].l; int d = st[sp].d;
        while (l - f > 16) {
            if (d == 0) { tkd_heap_select(f, l, l); tkd_sort_heap(f, l); break; }
            --d;
            TkE* cut = tkd_partition_pivot(f, l);
            // the reference recurses into [cut, l) FIRST and then continues with [f, cut): the two ranges are disjoint, so the
            // order in which they are finished does not change the result
            if (sp < 40) { st[sp].f = cut; st[sp].l = l; st[sp].d = d; ++sp; }
            l = cut;
        }
    }
    if (last - first > 16) {
        tkd_insertion_sort(first, first + 16);
        for (TkE* i = first + 16; i != last; ++i) tkd_unguarded_linear_insert(i);
    } else tkd_insertion_sort(first, last);
}

// ---- the same partition step by a whole wave -------------------------------------------------------------------------------------
// libstdc++'s __unguarded_partition walks two pointers towards each other over elements the other pointer has not touched yet, so
// its swaps are exactly: the t-th element from the LEFT that is not below the pivot <-> the t-th element from the RIGHT that is not
// above it, for as long as the former lies left of the latter (T pairs); it returns min(position of the (T+1)-th left element,
// position of the T-th right element) -- checked against the sequential form on 20 000 tie-rich arrays.  Both lists come out of one
// ballot / popcount sweep, the swaps are independent: N / 64 wave steps per pass instead of ~N dependent LDS round trips (a row of
// 1028 distances: ~0.15 ms sequentially, the pace of the whole kernel).
__device__ int tkw_partition_pivot(TkE* q, int* LA, int* LB, int first, int last, int lane) {
    if (lane == 0) tkd_move_median_to_first(q + first, q + first + 1, q + first + (last - first) / 2, q + last - 1);
    __syncthreads();
    const float pv = q[first].v;
    const int lo = first + 1, hi = last;
    const unsigned long long below = (1ull << lane) - 1ull;
    int nA = 0, nB = 0;
    for (int base = lo; base < hi; base += 64) {
        const int p = base + lane;
        const bool valid = p < hi;
        const float v = valid ? q[p].v : 0.f;
        const bool a = valid && !(v < pv), b = valid && !(pv < v);
        const unsigned long long ba = __ballot(a), bb = __ballot(b);
        if (a) LA[nA + __popcll(ba & below)] = p;
        if (b) LB[nB + __popcll(bb & below)] = p;
        nA += __popcll(ba); nB += __popcll(bb);
    }
    __syncthreads();
    const int nmin = nA < nB ? nA : nB;
    int T = 0;
    for (int base = 0; base < nmin; base += 64) {
        const int t = base + lane;
        const bool ok = t < nmin && LA[t] < LB[nB - 1 - t];
        const int c = __popcll(__ballot(ok));
        T += c;
        if (c < 64) break;                                  // (the pairs that swap are a prefix)
    }
    for (int t = lane; t < T; t += 64) {
        const int a = LA[t], b = LB[nB - 1 - t];
        const TkE x = q[a], y = q[b];
        q[a] = y; q[b] = x;
    }
    const int aT = T < nA ? LA[T] : 0x7fffffff, bp = T > 0 ? LB[nB - T] : hi;
    __syncthreads();
    return aT < bp ? aT : bp;
}
// std::nth_element(q, q + nth, q + n): __introselect, the partition steps by the wave, the rest by lane 0
__device__ void tkw_nth_element(TkE* q, int* LA, int* LB, int nth, int n, int lane) {
    int first = 0, last = n, depth = 2 * tkd_lg(n);
    while (last - first > 3) {
        if (depth == 0) {
            if (lane == 0) { tkd_heap_select(q + first, q + nth + 1, q + last); tkd_swap(q + first, q + nth); }
            __syncthreads();
            return;
        }
        --depth;
        const int cut = tkw_partition_pivot(q, LA, LB, first, last, lane);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0) tkd_insertion_sort(q + first, q + last);
    __syncthreads();
}

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
    if ((long long)m * 64 <= N) {                              // std::partial_sort
        if (lane == 0) { tkd_heap_select(q, q + m, q + N); tkd_sort_heap(q, q + m); }
    } else {
        if (m - 1 != N) tkw_nth_element(q, LA, LB, m - 1, N, lane);
        if (lane == 0) tkd_sort(q, q + (m - 1));
    }
    __syncthreads();
    if (lane >= drop && lane < m) out[lane - drop] = q[lane].i;
}

}  // namespace hsp

using namespace hsp;

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
