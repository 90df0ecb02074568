// tie_pass.h -- torch.topk's order among exactly equal distances, as device code shared by csrc/knn_exact.hip (the tie passes) and
// csrc/knn.hip (the coordinate search replays a tie-ridden row in place): libstdc++'s nth_element / sort / partial_sort restated
// (ATen TopKImpl.h takes them under a comparator that only sees the value), see knn_exact.hip's header comment.
#pragma once
#include "common.h"

namespace hsp {

struct TkE { float v; int i; };

#ifdef HSP_TIE_PROF
// stamps of ONE flagged row (tools/prof_tie_pass.py builds a private copy of the library with this switch): core clocks
static __device__ long long* g_tie_prof = nullptr;        // (one per translation unit; the tool sets knn_exact.hip's)
#define TIE_STAMP(slot) do { if (g_tie_prof && (threadIdx.x & 63) == 0) g_tie_prof[slot] = clock64(); } while (0)
#else
#define TIE_STAMP(slot) do { } while (0)
#endif

// The libstdc++ routines below are written once, in INDEX form, over an accessor: LdsAcc keeps the array in LDS (the full row of
// N distances), LaneAcc keeps a short array (<= 64 entries) with element p in LANE p of two registers and runs the sequential
// algorithm as wave-uniform scalar code over v_readlane / v_writelane -- a dependent step costs ~10 clocks instead of an LDS round
// trip (~120): the final std::sort of the 20 nearest took ~10 us per row by one lane on LDS, and Pool_layer's partial_sort
// (heap_select over all N entries, one dependent LDS read each) ~40 us; on lanes ~1 us and ~2 us.
struct LdsAcc {
    TkE* q;
    __device__ __forceinline__ TkE get(int p) const { return q[p]; }
    __device__ __forceinline__ void set(int p, TkE e) const { q[p] = e; }
};
struct LaneAcc {
    float v;
    int i;
    __device__ __forceinline__ TkE get(int p) const {
        p = __builtin_amdgcn_readfirstlane(p);
        TkE e;
        e.v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), p));
        e.i = __builtin_amdgcn_readlane(i, p);
        return e;
    }
    __device__ __forceinline__ void set(int p, TkE e) {
        const bool me = (int)(threadIdx.x & 63) == p;          // (a compare + two selects: no v_writelane builtin in this hipcc)
        v = me ? e.v : v;
        i = me ? e.i : i;
    }
};

__device__ __forceinline__ int tkd_lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

template <class A> __device__ __forceinline__ void tk_swap(A& a, int x, int y) {
    const TkE t = a.get(x);
    a.set(x, a.get(y));
    a.set(y, t);
}
template <class A> __device__ inline void tk_move_median_to_first(A& a, int result, int x, int y, int z) {
    const float va = a.get(x).v, vb = a.get(y).v, vc = a.get(z).v;
    if (va < vb) {
        if (vb < vc) tk_swap(a, result, y);
        else if (va < vc) tk_swap(a, result, z);
        else tk_swap(a, result, x);
    } else if (va < vc) tk_swap(a, result, x);
    else if (vb < vc) tk_swap(a, result, z);
    else tk_swap(a, result, y);
}
template <class A> __device__ inline int tk_partition_pivot(A& a, int first, int last) {
    const int mid = first + (last - first) / 2;
    tk_move_median_to_first(a, first, first + 1, mid, last - 1);
    const float pv = a.get(first).v;              // (the pivot slot is never swapped inside the loop)
    ++first;
    for (;;) {
        while (a.get(first).v < pv) ++first;
        --last;
        while (pv < a.get(last).v) --last;
        if (!(first < last)) return first;
        tk_swap(a, first, last);
        ++first;
    }
}
template <class A> __device__ inline void tk_unguarded_linear_insert(A& a, int last) {
    const TkE val = a.get(last);
    int next = last - 1;
    for (;;) {
        const TkE nx = a.get(next);
        if (!(val.v < nx.v)) break;
        a.set(last, nx);
        last = next;
        --next;
    }
    a.set(last, val);
}
template <class A> __device__ inline void tk_insertion_sort(A& a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const TkE val = a.get(i);
        if (val.v < a.get(first).v) {
            for (int p = i; p != first; --p) a.set(p, a.get(p - 1));          // move_backward(first, i, i + 1)
            a.set(first, val);
        } else tk_unguarded_linear_insert(a, i);
    }
}
template <class A> __device__ inline void tk_push_heap(A& a, int first, int hole, int top, TkE value) {
    int parent = (hole - 1) / 2;
    while (hole > top) {
        const TkE pe = a.get(first + parent);
        if (!(pe.v < value.v)) break;
        a.set(first + hole, pe);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.set(first + hole, value);
}
template <class A> __device__ inline void tk_adjust_heap(A& a, int first, int hole, int len, TkE value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        const TkE c1 = a.get(first + child), c0 = a.get(first + child - 1);
        TkE take = c1;
        if (c1.v < c0.v) { --child; take = c0; }
        a.set(first + hole, take);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.set(first + hole, a.get(first + child - 1));
        hole = child - 1;
    }
    tk_push_heap(a, first, hole, top, value);
}
template <class A> __device__ inline void tk_make_heap(A& a, int first, int last) {
    const int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        tk_adjust_heap(a, first, parent, len, a.get(first + parent));
        if (parent == 0) return;
    }
}
template <class A> __device__ inline void tk_pop_heap(A& a, int first, int last, int result) {
    const TkE value = a.get(result);
    a.set(result, a.get(first));
    tk_adjust_heap(a, first, 0, last - first, value);
}
template <class A> __device__ inline void tk_heap_select(A& a, int first, int middle, int last) {
    tk_make_heap(a, first, middle);
    for (int i = middle; i < last; ++i)
        if (a.get(i).v < a.get(first).v) tk_pop_heap(a, first, middle, i);
}
template <class A> __device__ inline void tk_sort_heap(A& a, int first, int last) {
    while (last - first > 1) { --last; tk_pop_heap(a, first, last, last); }
}
// std::sort of a short range: __introsort_loop (recursion on the upper part turned into a small explicit stack) + final insertion sort
template <class A> __device__ inline void tk_sort(A& a, int first, int last) {
    if (first == last) return;
    int sf[40], sl[40], sd[40];
    int sp = 0;
    sf[0] = first; sl[0] = last; sd[0] = 2 * tkd_lg(last - first); sp = 1;
    while (sp) {
        --sp;
        int f = sf[sp], l = sl[sp], d = sd[sp];
        while (l - f > 16) {
            if (d == 0) { tk_heap_select(a, f, l, l); tk_sort_heap(a, f, l); break; }
            --d;
            const int cut = tk_partition_pivot(a, f, l);
            // the reference recurses into [cut, l) FIRST and then continues with [f, cut): the two ranges are disjoint, so the
            // order in which they are finished does not change the result
            if (sp < 40) { sf[sp] = cut; sl[sp] = l; sd[sp] = d; ++sp; }
            l = cut;
        }
    }
    if (last - first > 16) {
        tk_insertion_sort(a, first, first + 16);
        for (int i = first + 16; i != last; ++i) tk_unguarded_linear_insert(a, i);
    } else tk_insertion_sort(a, first, last);
}
// a range of at most 32 entries needs at most 16 pending sub-ranges; ranges <= 16 skip the loop entirely
template <class A> __device__ inline void tk_sort_short(A& a, int first, int last) {
    if (last - first <= 16) { tk_insertion_sort(a, first, last); return; }
    tk_sort(a, first, last);
}

// ---- short arrays on the lanes, WAVE-PARALLEL forms (modelled lane by lane against the sequential routines in
// tools/sim_tie_pass.py: 3 400 tie-rich rows, all list lengths / row lengths of the stack, zero mismatches) -----------------------
// A single wave running a scalar program retires one instruction every ~5-8 clocks (nothing else hides its dependent issue), so
// the sequential routines above cost ~300 clocks per element step on the lanes and about the same on LDS (measured with
// -DHSP_TIE_PROF: 21 us for Pool_layer's heap_select over a 1028-entry row, 13 us for nth_element, 7 us for the sort of 20).
// What is parallel in them:
//   __adjust_heap + __push_heap: the hole sinks along the path of "larger child" choices, which every node can make at once
//       (two ballots); the walk over those masks is scalar; the value then rises to just below the deepest path node that is not
//       below it (one ballot + find-last-bit); net effect: path nodes above the landing slot take their path child's entry.
//   __unguarded_partition_pivot on <= 64 entries: the t-th entry from the left that is not below the pivot swaps with the t-th
//       from the right that is not above it while the former lies left of the latter -- ranks from two ballots, partners through
//       64-entry LDS tables, the swap a lane shuffle.
//   the final insertion sorts: stable, i.e. a rank by (value, position): count + ds_permute.
__device__ __forceinline__ float rl_f(float x, int p) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), p)); }
__device__ __forceinline__ int rl_i(int x, int p) { return __builtin_amdgcn_readlane(x, p); }

// std::__adjust_heap(first = lane 0, hole0, len, value) on a heap whose node j lives in lane j
__device__ __forceinline__ void lh_adjust(LaneAcc& H, int lane, int hole0, int len, float valv, int vali) {
    const int l = 2 * lane + 1, r = l + 1;
    const float vl = __shfl(H.v, l & 63), vr = __shfl(H.v, r & 63);
    const int il = __shfl(H.i, l & 63), ir = __shfl(H.i, r & 63);
    const bool two = r < len;
    const bool right = two && !(vr < vl);                      // child = 2 (child + 1); if (h[child] < h[child - 1]) --child;
    const bool left = (two && vr < vl) || (!two && l < len);   // (a last node with a left child only: the even-length case)
    const unsigned long long mR = __ballot(right), mL = __ballot(left);
    int cur = hole0;
    unsigned long long path = 1ull << cur;
    for (;;) {
        if ((mR >> cur) & 1ull) cur = 2 * cur + 2;
        else if ((mL >> cur) & 1ull) cur = 2 * cur + 1;
        else break;
        path |= 1ull << cur;
    }
    const bool onp = (path >> lane) & 1ull;
    // __push_heap from the leaf: the hole passes a parent while parent < value; the parents' entries are the path nodes' OWN
    // (the sink moved each up by one) -- so the value lands on the deepest path node below hole0 that is not below it, else on hole0
    const unsigned long long fail = __ballot(onp && lane != hole0 && !(H.v < valv));
    const int s = fail ? 63 - __builtin_clzll(fail) : hole0;
    if (onp && lane < s) { H.v = right ? vr : vl; H.i = right ? ir : il; }
    if (lane == s) { H.v = valv; H.i = vali; }
}
__device__ __forceinline__ void lh_make_heap(LaneAcc& H, int lane, int len) {
    if (len < 2) return;
    for (int parent = (len - 2) / 2; parent >= 0; --parent) lh_adjust(H, lane, parent, len, rl_f(H.v, parent), rl_i(H.i, parent));
}
__device__ __forceinline__ void lh_sort_heap(LaneAcc& H, int lane, int len) {
    for (int last = len - 1; last >= 1; --last) {              // __pop_heap(first, last, last)
        const float valv = rl_f(H.v, last);
        const int vali = rl_i(H.i, last);
        const float tv = rl_f(H.v, 0);
        const int ti = rl_i(H.i, 0);
        if (lane == last) { H.v = tv; H.i = ti; }
        lh_adjust(H, lane, 0, last, valv, vali);
    }
}
// std::__unguarded_partition_pivot(first, last) on lane-resident entries; SA / SB: 64 ints of LDS each
__device__ __forceinline__ int lp_partition(LaneAcc& H, int lane, int first, int last, int* SA, int* SB) {
    const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
    const float va = rl_f(H.v, x), vb = rl_f(H.v, y), vc = rl_f(H.v, z);
    int sel;
    if (va < vb) sel = vb < vc ? y : (va < vc ? z : x);
    else sel = va < vc ? x : (vb < vc ? z : y);
    const float fv = rl_f(H.v, first), pv = rl_f(H.v, sel);
    const int fi = rl_i(H.i, first), si = rl_i(H.i, sel);
    if (lane == first) { H.v = pv; H.i = si; }
    if (lane == sel) { H.v = fv; H.i = fi; }
    const bool in = lane > first && lane < last;
    const bool a = in && !(H.v < pv), b = in && !(pv < H.v);
    const unsigned long long ba = __ballot(a), bb = __ballot(b);
    const int ra = __popcll(ba & ((1ull << lane) - 1ull)), rb = lane == 63 ? 0 : __popcll(bb >> (lane + 1));
    if (a) SA[ra] = lane;
    if (b) SB[rb] = lane;
    __builtin_amdgcn_wave_barrier();
    const int nA = __popcll(ba), nB = __popcll(bb), nmin = nA < nB ? nA : nB;
    const int pb = (a && ra < nmin) ? SB[ra] : -1;
    const int T = __popcll(__ballot(pb > lane));               // (the pairs that swap are a prefix)
    int partner = lane;
    if (a && ra < T) partner = pb;
    if (b && rb < T) partner = SA[rb];
    H.v = __shfl(H.v, partner);
    H.i = __shfl(H.i, partner);
    const int aT = T < nA ? SA[T] : 0x7fffffff, bp = T > 0 ? SB[T - 1] : last;
    __builtin_amdgcn_wave_barrier();
    return __builtin_amdgcn_readfirstlane(aT < bp ? aT : bp);
}
// stable sort of [first, last) == what __insertion_sort / __unguarded_linear_insert leave
__device__ __forceinline__ void lp_ranksort(LaneAcc& H, int lane, int first, int last) {
    const bool in = lane >= first && lane < last;
    int cnt = 0;
    for (int t = first; t < last; ++t) {
        const float vt = rl_f(H.v, t);
        cnt += (vt < H.v || (vt == H.v && t < lane)) ? 1 : 0;
    }
    const int dest = in ? first + cnt : lane;
    H.v = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(H.v)));
    H.i = __builtin_amdgcn_ds_permute(dest << 2, H.i);
}
// std::__introselect on lane-resident entries
__device__ inline void lp_introselect(LaneAcc& H, int lane, int first, int nth, int last, int depth, int* SA, int* SB) {
    while (last - first > 3) {
        if (depth == 0) { tk_heap_select(H, first, nth + 1, last); tk_swap(H, first, nth); return; }   // (sequential: never seen)
        --depth;
        const int cut = lp_partition(H, lane, first, last, SA, SB);
        if (cut <= nth) first = cut; else last = cut;
    }
    lp_ranksort(H, lane, first, last);
}
// std::sort of at most 32 lane-resident entries: of the two parts a partition leaves only one can exceed 16, so
// __introsort_loop's recursion never has work pending; the final insertion sort runs over the whole range
__device__ inline void lp_sort(LaneAcc& H, int lane, int first, int last, int* SA, int* SB) {
    int f = first, l = last, d = 2 * tkd_lg(last - first > 0 ? last - first : 1);
    while (l - f > 16) {
        if (d == 0) { tk_heap_select(H, f, l, l); tk_sort_heap(H, f, l); break; }                      // (sequential: never seen)
        --d;
        const int cut = lp_partition(H, lane, f, l, SA, SB);
        if (l - cut > 16) f = cut; else l = cut;
    }
    lp_ranksort(H, lane, first, last);
}

// ---- the same partition step by a whole wave -------------------------------------------------------------------------------------
// libstdc++'s __unguarded_partition walks two pointers towards each other over elements the other pointer has not touched yet, so
// its swaps are exactly: the t-th element from the LEFT that is not below the pivot <-> the t-th element from the RIGHT that is not
// above it, for as long as the former lies left of the latter (T pairs); it returns min(position of the (T+1)-th left element,
// position of the T-th right element) -- checked against the sequential form on 20 000 tie-rich arrays.  Both lists come out of one
// ballot / popcount sweep, the swaps are independent: N / 64 wave steps per pass instead of ~N dependent LDS round trips (a row of
// 1028 distances: ~0.15 ms sequentially, the pace of the whole kernel).
// (The workgroup is ONE wave: its LDS operations execute in program order, so a store by one lane is seen by a later load of
// another without a barrier; wave_barrier only pins the compiler's order.)
__device__ inline int tkw_partition_pivot(TkE* q, int* LA, int* LB, int first, int last, int lane) {
    // __move_median_to_first(first, first + 1, mid, last - 1): every lane reads the three candidates (broadcast loads, one round
    // trip) and picks; lane 0 swaps.  The scan below substitutes the swapped-in value at the donor slot instead of waiting for it.
    const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
    const TkE ef = q[first];
    const float va = q[x].v, vb = q[y].v, vc = q[z].v;
    int sel;
    if (va < vb) sel = vb < vc ? y : (va < vc ? z : x);
    else sel = va < vc ? x : (vb < vc ? z : y);
    const float pv = sel == x ? va : (sel == y ? vb : vc);
    if (lane == 0) { const TkE es = q[sel]; q[sel] = ef; q[first] = es; }
    const int lo = first + 1, hi = last;
    const unsigned long long below = (1ull << lane) - 1ull;
    int nA = 0, nB = 0;
    for (int base = lo; base < hi; base += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + 64 * u + lane;
            v[u] = q[p < hi ? p : hi - 1].v;
            if (p == sel) v[u] = ef.v;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + 64 * u + lane;
            const bool valid = p < hi;
            const bool a = valid && !(v[u] < pv), b = valid && !(pv < v[u]);
            const unsigned long long ba = __ballot(a), bb = __ballot(b);
            if (a) LA[nA + __popcll(ba & below)] = p;
            if (b) LB[nB + __popcll(bb & below)] = p;
            nA += __popcll(ba); nB += __popcll(bb);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int nmin = nA < nB ? nA : nB;
    int T = 0;
    for (int base = 0; base < nmin; base += 256) {          // four 64-pair groups per round trip
        int la[4], lb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = base + 64 * u + lane;
            la[u] = LA[t < nmin ? t : 0];
            lb[u] = LB[t < nmin ? nB - 1 - t : 0];
        }
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = base + 64 * u + lane;
            const int cu = __popcll(__ballot(t < nmin && la[u] < lb[u]));
            c += (c == 64 * u) ? cu : 0;                    // (the pairs that swap are a prefix: stop counting at the first gap)
        }
        T += c;
        if (c < 256) break;
    }
    for (int t0 = lane; t0 < T; t0 += 256) {
        int pa[4], pb[4];
        TkE ex[4], ey[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 64 * u < T ? t0 + 64 * u : T - 1;
            pa[u] = LA[t]; pb[u] = LB[nB - 1 - t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { ex[u] = q[pa[u]]; ey[u] = q[pb[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + 64 * u < T) { q[pa[u]] = ey[u]; q[pb[u]] = ex[u]; }
    }
    const int aT = T < nA ? LA[T] : 0x7fffffff, bp = T > 0 ? LB[nB - T] : hi;
    __builtin_amdgcn_wave_barrier();
    return aT < bp ? aT : bp;
}
template <class A> __device__ inline void tk_introselect(A& a, int first, int nth, int last, int depth) {
    while (last - first > 3) {
        if (depth == 0) { tk_heap_select(a, first, nth + 1, last); tk_swap(a, first, nth); return; }
        --depth;
        const int cut = tk_partition_pivot(a, first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    tk_insertion_sort(a, first, last);
}
// std::nth_element(q, q + nth, q + n): __introselect.  Ranges above 64 entries: partition steps by the wave on LDS (~1 us each:
// a handful of dependent LDS round trips); once the range fits the wave it moves onto the lanes and the sequential algorithm
// finishes there (~20 clocks per step), then goes back.
__device__ inline void tkw_nth_element(TkE* q, int* LA, int* LB, int nth, int n, int lane) {
    int first = 0, last = n, depth = 2 * tkd_lg(n);
    while (last - first > 64) {
        if (depth == 0) {
            LdsAcc a{q};
            if (lane == 0) { tk_heap_select(a, first, nth + 1, last); tk_swap(a, first, nth); }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        --depth;
        const int cut = tkw_partition_pivot(q, LA, LB, first, last, lane);
        if (cut <= nth) first = cut; else last = cut;
    }
    const int len = last - first;
    LaneAcc R;
    const TkE e = q[first + (lane < len ? lane : 0)];
    R.v = e.v; R.i = e.i;
    lp_introselect(R, lane, 0, nth - first, len, depth, LA, LB);
    if (lane < len) { TkE o; o.v = R.v; o.i = R.i; q[first + lane] = o; }
    __builtin_amdgcn_wave_barrier();
}

// torch.topk(d, m, largest=False, sorted=True) of the N (value, index) entries of q (LDS, index order): rank r ends up in LANE r of
// the result (r < m <= 64).  ATen's TopKImpl.h: std::partial_sort for m * 64 <= N -- here the heap lives on the lanes and the scan
// over the other N - m entries is one ballot per 64 of them (an entry enters the heap only if it is below the CURRENT top, so the
// entries of a chunk are taken in order, the ballot refreshed after each pop) and q is left UNTOUCHED; otherwise std::nth_element
// (in place, wave-parallel partitions) + std::sort of the first m - 1 (on the lanes).  ``destroys`` tells the caller whether q
// was permuted.
__device__ __forceinline__ bool tkw_topk_destroys(int m, int N) { return !((long long)m * 64 <= N); }

__device__ inline LaneAcc tkw_topk(TkE* q, int* LA, int* LB, int m, int N, int lane) {
    LaneAcc H;
    if ((long long)m * 64 <= N) {                              // std::partial_sort(q, q + m, q + N)
        const TkE e0 = q[lane < m ? lane : 0];
        H.v = e0.v; H.i = e0.i;
        TIE_STAMP(1);
        lh_make_heap(H, lane, m);
        float top = rl_f(H.v, 0);
        TIE_STAMP(2);
        for (int base = m; base < N; base += 64) {
            const int p = base + lane;
            const bool valid = p < N;
            const TkE e = q[valid ? p : 0];
            unsigned long long mask = __ballot(valid && e.v < top);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                TkE val;
                val.v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.v), l));
                val.i = __builtin_amdgcn_readlane(e.i, l);
                lh_adjust(H, lane, 0, m, val.v, val.i);        // __pop_heap(first, middle, i) minus the store to *i (never read again)
                top = rl_f(H.v, 0);
                const unsigned long long later = l == 63 ? 0ull : ~((2ull << l) - 1ull);
                mask = __ballot(valid && e.v < top) & later;
            }
        }
        TIE_STAMP(3);
        lh_sort_heap(H, lane, m);
        TIE_STAMP(4);
    } else {
        TIE_STAMP(5);
        if (m - 1 != N) tkw_nth_element(q, LA, LB, m - 1, N, lane);
        TIE_STAMP(6);
        const TkE e0 = q[lane < m ? lane : 0];
        H.v = e0.v; H.i = e0.i;
        lp_sort(H, lane, 0, m - 1, LA, LB);
        TIE_STAMP(7);
    }
    return H;
}

// one WAVE's share (rows w, w + G, ...) of the tie pass of a coordinate search; scratch: 16 N bytes of LDS owned by this wave
__device__ inline void tie_rows_xyz(char* scratch, const float* __restrict__ x, const uint8_t* __restrict__ tie, int B, int N,
                                    int k, int k2, int drop, int32_t* __restrict__ idx, int32_t* __restrict__ idx2,
                                    int* __restrict__ nties, int w, int G, int lane) {
    TkE* q = reinterpret_cast<TkE*>(scratch);
    int* LA = reinterpret_cast<int*>(q + N);
    int* LB = LA + N;
    const int rows = B * N;
    // wave w owns rows w, w + G, ...; their flags are read 64 at a time (one round trip per 64 rows: a per-row read made the pass
    // cost ~1 us per unflagged row and wave -- 0.4 ms at B = 64, N = 4096).  The selection already wrote every row's short list as
    // the prefix of its long one: final wherever bit 1 is clear (the k2 + drop + 1 nearest are pairwise different, and a tie
    // further down the long list cannot reach the short one)
    for (int base = 0; base < rows; base += 64 * G) {
      const int row_l = base + lane * G + w;
      const int fl = row_l < rows ? (int)tie[row_l] : 0;
      unsigned long long todo = __ballot(fl != 0);
      while (todo) {
        const int tl = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int row = base + tl * G + w;
        const int flags = __builtin_amdgcn_readlane(fl, tl);
        int32_t* out = idx + (size_t)row * k;
        int32_t* out2 = idx2 ? idx2 + (size_t)row * k2 : nullptr;
        const int b = row / N, i = row - b * N;
        const float* xb = x + (size_t)b * N * 3;
        const float qx = xb[i * 3], qy = xb[i * 3 + 1], qz = xb[i * 3 + 2];
        const float qq = quad3(qx, qy, qz);
        if (lane == 0 && nties) atomicAdd(nties, 1);
        TIE_STAMP(0);
        // the list whose search leaves q untouched (partial_sort) goes first: one fill serves both
        const int m1 = k + drop, m2 = k2 + drop;
        const bool want2 = out2 && (flags & 2);
        const bool second_first = want2 && !tkw_topk_destroys(m2, N) && tkw_topk_destroys(m1, N);
        bool filled = false;
        for (int pass = 0; pass < (want2 ? 2 : 1); ++pass) {
            const bool short_list = (pass == 0) == second_first && want2;
            if (!filled) {
                for (int j0 = lane; j0 < N; j0 += 8 * 64) {                 // eight rows' loads in flight per lane
                    float px[8], py[8], pz[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + 64 * u < N ? j0 + 64 * u : N - 1;
                        px[u] = xb[j * 3]; py[u] = xb[j * 3 + 1]; pz[u] = xb[j * 3 + 2];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + 64 * u;
                        const float inner = dot3_chain(qx, qy, qz, px[u], py[u], pz[u]);
                        // (NaN / +inf -> FLT_MAX as in the selection kernels: the partition loops need a total order)
                        TkE e;
                        e.v = fminf(add_rn(add_rn(mul_rn(inner, -2.0f), quad3(px[u], py[u], pz[u])), qq), 3.402823466e+38f);
                        e.i = j;
                        if (j < N) q[j] = e;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                TIE_STAMP(8);
            }
            const int m = short_list ? m2 : m1;
            const LaneAcc H = tkw_topk(q, LA, LB, m, N, lane);
            filled = !tkw_topk_destroys(m, N);
            int32_t* o = short_list ? out2 : out;
            if (lane >= drop && lane < m) o[lane - drop] = H.i;
            __builtin_amdgcn_wave_barrier();
        }
      }
    }
}


}  // namespace hsp
