// csr.hip -- reverse-edge (CSR) index of a KNN graph, for atomic-free gather-form backward passes.
//
// A neighbour index idx (B,Nq,k) says "query i lists source row m = idx[i][n] at slot n".  Every
// backward of the hot path scatters gradient from (i,n) to m (reference: the _index_put_impl_
// accumulate of the gather in network/fs_net_repo/gcn3d.py:39-47).  Instead of fp32 atomics
// (measured ~20 G atomics/s on MI355X: 0.6 ms for one N=1028 layer), the backward kernels walk, for
// each source row m, the list of edges e = i*k + n that point at it, and write each gradient row once.
//
//   rev_off  (B, Nsrc+1) int32 : list of row m = rev_edge[b][rev_off[m] .. rev_off[m+1])
//   rev_edge (B, Nq*k)   int32 : edge ids e = i*k + n, ASCENDING within each list -> the summation
//                                order of every backward is fixed: bit-reproducible gradients.
//
// One 1024-thread workgroup per cloud: LDS histogram -> workgroup exclusive scan -> chunk-ordered fill
// (edges are taken 1024 at a time, so a list is sorted up to swaps inside one chunk) -> per-row
// insertion sort, O(length + inversions).
#include "common.h"

namespace hsp {

#define REV_THREADS 1024

// EDGES_IN_LDS: the edge array is assembled and sorted in LDS and written out once, coalesced (the per-row
// insertion sort is a chain of dependent accesses: ~100 cycles a step in LDS, 1-2 us a step in global memory)
template <bool EDGES_IN_LDS>
__global__ __launch_bounds__(REV_THREADS) void rev_build_kernel(const int32_t* __restrict__ idx, int Nq, int Nsrc,
                                                                int k, int kstride, int32_t* __restrict__ rev_off,
                                                                int32_t* __restrict__ rev_edge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cnt = reinterpret_cast<int*>(smem);            // Nsrc + 1 (histogram, then cursor)
    int* led = cnt + Nsrc + 1;                          // E edges (EDGES_IN_LDS)
    __shared__ int wsum[REV_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* ib = idx + (size_t)b * Nq * kstride;
    int32_t* off = rev_off + (size_t)b * (Nsrc + 1);
    int32_t* edge = rev_edge + (size_t)b * Nq * k;
    int* ed = EDGES_IN_LDS ? led : edge;
    const int E = Nq * k;

    for (int m = tid; m <= Nsrc; m += REV_THREADS) cnt[m] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += REV_THREADS) {
        const int i = e / k, n = e - i * k;
        atomicAdd(&cnt[ib[(size_t)i * kstride + n]], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..Nsrc): thread t owns a contiguous slice; wave scan by shuffles, 16 wave totals by wave 0
    const int per = (Nsrc + REV_THREADS - 1) / REV_THREADS;
    const int lo = min(tid * per, Nsrc), hi = min(lo + per, Nsrc);
    int s = 0;
    for (int m = lo; m < hi; ++m) s += cnt[m];
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d);
        if (lane >= d) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        int t = lane < REV_THREADS / 64 ? wsum[lane] : 0;
#pragma unroll
        for (int d = 1; d < REV_THREADS / 64; d <<= 1) {
            const int v = __shfl_up(t, d);
            if (lane >= d) t += v;
        }
        if (lane < REV_THREADS / 64) wsum[lane] = t;    // inclusive wave totals
    }
    __syncthreads();
    int run = inc - s + (wv ? wsum[wv - 1] : 0);
    for (int m = lo; m < hi; ++m) {
        const int c = cnt[m];
        cnt[m] = run;                                    // becomes the fill cursor
        off[m] = run;
        run += c;
    }
    if (tid == 0) off[Nsrc] = E;
    __syncthreads();
    // chunk-ordered fill
    for (int e0 = 0; e0 < E; e0 += REV_THREADS) {
        const int e = e0 + tid;
        if (e < E) {
            const int i = e / k, n = e - i * k;
            const int pos = atomicAdd(&cnt[ib[(size_t)i * kstride + n]], 1);
            ed[pos] = e;
        }
        __syncthreads();
    }
    if (!EDGES_IN_LDS) __threadfence_block();
    __syncthreads();
    // after the fill cnt[m] is the END of row m, its start the end of row m-1
    if (EDGES_IN_LDS && k == 1) {
        // row maps (k = 1: few, long rows).  Ascending order by RANK: one thread per edge counts the smaller edge ids of its row (O(row length) LDS reads,
        // parallel over edges) and writes the edge to its final global slot -- a thread-per-row insertion sort is a
        // serial O(length^2) chain (18 us for 64 rows of ~16 edges)
        for (int p = tid; p < E; p += REV_THREADS) {
            const int e = led[p];
            const int i = e / k, n = e - i * k;
            const int m = ib[(size_t)i * kstride + n];
            const int a = m ? cnt[m - 1] : 0, z = cnt[m];
            int c = 0;
            for (int q = a; q < z; ++q) c += led[q] < e ? 1 : 0;
            edge[a + c] = e;
        }
    } else {
        for (int m = tid; m < Nsrc; m += REV_THREADS) {    // nearly sorted lists (in LDS when they fit): insertion sort per row
            const int a = m ? cnt[m - 1] : 0, z = cnt[m];
            for (int p = a + 1; p < z; ++p) {
                const int v = ed[p];
                int q = p - 1;
                while (q >= a && ed[q] > v) { ed[q + 1] = ed[q]; --q; }
                ed[q + 1] = v;
            }
        }
        if (EDGES_IN_LDS) {
            __syncthreads();
            for (int e = tid; e < E; e += REV_THREADS) edge[e] = led[e];
        }
    }
}

}  // namespace hsp

using namespace hsp;

extern "C" int hsp_rev_build(const int32_t* idx, int B, int Nq, int Nsrc, int k, int kstride, int32_t* rev_off,
                             int32_t* rev_edge, hspStream_t stream) {
    if (!idx || !rev_off || !rev_edge || B <= 0 || Nq <= 0 || Nsrc <= 0 || k <= 0 || kstride < k) return HSP_ERR_BAD_ARG;
    const size_t lds0 = (size_t)(Nsrc + 1) * sizeof(int);
    if (lds0 > 140 * 1024) return HSP_ERR_UNSUPPORTED;
    const size_t lds1 = lds0 + (size_t)Nq * k * sizeof(int);
    const bool in_lds = lds1 <= 140 * 1024;
    const size_t lds = in_lds ? lds1 : lds0;
    auto kern = in_lds ? rev_build_kernel<true> : rev_build_kernel<false>;
    if (lds > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(REV_THREADS), lds, as_stream(stream), idx, Nq, Nsrc, k, kstride, rev_off,
                       rev_edge);
    return check_launch();
}
