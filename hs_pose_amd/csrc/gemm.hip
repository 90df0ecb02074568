// gemm.hip -- split-K weight-gradient GEMM on the fp32 matrix cores of gfx950.
//
//   C[m][n] = sum_k A[k][m] * B[k][n]          A (K x M, row stride lda), B (K x N, row stride ldb)
//   colsum[n] = sum_k B[k][n]                  (optional: the bias gradient, fused)
//
// This is the parameter-gradient shape of every dense op of the HS stack (reference: autograd of
// `feature_map @ self.weights + self.bias`, network/fs_net_repo/gcn3d.py:171, and of the 1x1 Conv1d
// layers :85,:149,:186): K = B*N point rows (16 448 at B=16, N=1028) against a small M x N weight
// (128 x 1024 ...).  A library GEMM sees 32 output tiles and no parallelism over K (measured 36 TF/s);
// here K is split over up to ~2000 independent waves.
//
// Each wave owns a 64x64 output tile for one K slice: 4 accumulators of v_mfma_f32_32x32x2_f32
// (fp32 in, fp32 accumulate, an exact k-ordered fma chain).  Both operands are k-major rows, which
// is exactly the MFMA operand layout (lane l supplies A[k = 2s + (l>>5)][m], B[k][n]): a lane reads one
// float2 of A and one of B per k-pair straight from global memory -- the two components feed the
// even-row / odd-row (even-col / odd-col) MFMA tiles, so every half-wave reads 256 contiguous bytes;
// no LDS, no barriers, register double-buffered prefetch 4 k-pairs deep.  Partials go to a workspace
// and are folded in a fixed order (deterministic) by wgrad_reduce_kernel, which writes C with an
// arbitrary leading dimension (so a gradient can land in a column block of a larger tensor).
#include "common.h"
#include "folds.h"

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#define WG_UNROLL 4   // k-pairs per prefetch group

// KB = 1: the 4 waves of a workgroup take 4 neighbouring tiles of one K slice (they share B rows in L1).
// KB = 4: they take 4 consecutive K slices of ONE tile and fold their accumulators through LDS (fixed order), so
//         a small output with a long K (128x128 <- 16448 rows) can be cut into hundreds of slices -- enough waves to
//         fill the chip -- without multiplying the partial sums the reduce kernel has to read.
// FT: storage type of A and B (float, or bf16_t: the pair of columns a lane owns is one 4-byte load, widened to fp32 --
// the products still run on the fp32 MFMA, exact and in the same order as for fp32 operands)
template <bool COLSUM, int KB, typename FT>
__device__ __forceinline__ void wgrad_body(const int bid, const FT* __restrict__ A, int lda,
                                           const FT* __restrict__ B, int ldb, int M, int N, int K,
                                           int SK, int kslice, float* __restrict__ part,
                                           float* __restrict__ cs_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and provably so
    const int tiles = (M >> 6) * (N >> 6);
    int slice, t;
    if (KB == 1) {
        const int w = bid * 4 + wv;
        if (w >= tiles * SK) return;
        slice = w / tiles; t = w - slice * tiles;
    } else {
        const int grp = bid / tiles;
        t = bid - grp * tiles;
        slice = grp * 4 + wv;                               // slices past K read zeros
    }
    const int tiles_m = M >> 6;
    const int tn = t / tiles_m, tm = t - tn * tiles_m;   // tm fastest: the waves of a block share B rows (L1/L2 reuse)
    const int m0 = tm << 6, n0 = tn << 6;
    const int k0 = slice * kslice;
    const int k1 = min(K, k0 + kslice);
    const int h = lane >> 5, i = lane & 31;
    // operands by raw buffer loads: descriptor over the whole matrix, per-lane byte offset fixed for the kernel
    // (row parity h, columns 2i, 2i+1 of the tile), row pair selected by a scalar offset -> no vector address
    // arithmetic in the loop, and rows >= K (ragged end of the last slice; slices are multiples of 8 rows) read 0
    constexpr int ES = sizeof(FT);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<FT*>(A), 0, (int)((((size_t)K - 1) * lda + M) * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<FT*>(B), 0, (int)((((size_t)K - 1) * ldb + N) * ES), 0x00020000);
    const int va = (h * lda + m0 + 2 * i) * ES, vb = (h * ldb + n0 + 2 * i) * ES;
    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
    float2 cs = make_float2(0.f, 0.f);

    // two register buffers, used alternately (no copies): the loads of group g+1 are in flight under the 16 MFMAs
    // of group g and are first touched when group g+1 starts
    float2 a0[WG_UNROLL], b0[WG_UNROLL], a1[WG_UNROLL], b1[WG_UNROLL];
    auto load_group = [&](int kk, float2 (&aa)[WG_UNROLL], float2 (&bb)[WG_UNROLL]) {
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            if constexpr (ES == 4) {
                const auto x = __builtin_amdgcn_raw_buffer_load_b64(ra, va, (kk + 2 * u) * lda * 4, 0);
                const auto y = __builtin_amdgcn_raw_buffer_load_b64(rb, vb, (kk + 2 * u) * ldb * 4, 0);
                aa[u] = make_float2(__int_as_float(x[0]), __int_as_float(x[1]));
                bb[u] = make_float2(__int_as_float(y[0]), __int_as_float(y[1]));
            } else {
                const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(ra, va, (kk + 2 * u) * lda * 2, 0);
                const unsigned y = __builtin_amdgcn_raw_buffer_load_b32(rb, vb, (kk + 2 * u) * ldb * 2, 0);
                aa[u] = make_float2(__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u));
                bb[u] = make_float2(__uint_as_float(y << 16), __uint_as_float(y & 0xffff0000u));
            }
        }
    };
    auto mma_group = [&](const float2 (&aa)[WG_UNROLL], const float2 (&bb)[WG_UNROLL]) {
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].x, bb[u].x, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].x, bb[u].y, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].y, bb[u].x, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].y, bb[u].y, c11, 0, 0, 0);
            if (COLSUM) { cs.x += bb[u].x; cs.y += bb[u].y; }
        }
    };
    constexpr int GR = 2 * WG_UNROLL;                       // k rows per group
    // the next group is requested unconditionally (past the slice it is simply not used; past K it reads 0):
    // a branch around the loads would make the compiler wait for them at the join
    // Branch-free body (slices are multiples of 2 groups; the ragged end of the last slice multiplies zeros): with
    // an early exit between a load group and its MFMAs hipcc sinks the loads behind the branch, next to their
    // use, and nothing is prefetched.  sched_barrier keeps the machine scheduler from interleaving them back.
    load_group(k0, a0, b0);
    for (int kk = k0; kk < k1; kk += 2 * GR) {
        load_group(kk + GR, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_group(kk + 2 * GR, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator r <-> tile row (r&3) + 8*(r>>2) + 4*h, tile col i; tile (ta,tb) holds C[m0+2*row+ta][n0+2*i+tb]
    if (KB == 1) {
        float* pc = part + (size_t)slice * M * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<float2*>(pc + (size_t)(m0 + 2 * row) * N + n0 + 2 * i) = make_float2(c00[r], c01[r]);
            *reinterpret_cast<float2*>(pc + (size_t)(m0 + 2 * row + 1) * N + n0 + 2 * i) = make_float2(c10[r], c11[r]);
        }
        if (COLSUM && tm == 0) {
            cs.x += __shfl_xor(cs.x, 32);
            cs.y += __shfl_xor(cs.y, 32);
            if (h == 0) *reinterpret_cast<float2*>(cs_part + (size_t)slice * N + n0 + 2 * i) = cs;
        }
        return;
    }
    // KB == 4: buf[wave][acc][r][lane]; wave a then folds accumulator a of the four waves (order 0..3) and stores its
    // interleaved quarter of the tile; the column sums go through csb[wave][lane]
    float* buf = reinterpret_cast<float*>(smem);
    float2* csb = reinterpret_cast<float2*>(buf + 4 * 4 * 16 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        buf[((wv * 4 + 0) * 16 + r) * 64 + lane] = c00[r];
        buf[((wv * 4 + 1) * 16 + r) * 64 + lane] = c01[r];
        buf[((wv * 4 + 2) * 16 + r) * 64 + lane] = c10[r];
        buf[((wv * 4 + 3) * 16 + r) * 64 + lane] = c11[r];
    }
    if (COLSUM) csb[wv * 64 + lane] = cs;
    __syncthreads();
    const int grp = bid / tiles;
    float* pc = part + (size_t)grp * M * N;
    const int ta = wv >> 1, tb = wv & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = buf[((0 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((1 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((2 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((3 * 4 + wv) * 16 + r) * 64 + lane];
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        pc[(size_t)(m0 + 2 * row + ta) * N + n0 + 2 * i + tb] = v;
    }
    if (COLSUM && tm == 0 && wv == 0) {
        float2 c = csb[lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) { c.x += csb[w2 * 64 + lane].x; c.y += csb[w2 * 64 + lane].y; }
        c.x += __shfl_xor(c.x, 32);
        c.y += __shfl_xor(c.y, 32);
        if (h == 0) *reinterpret_cast<float2*>(cs_part + (size_t)grp * N + n0 + 2 * i) = c;
    }
}

template <bool COLSUM, int KB, typename FT>
__global__ __launch_bounds__(256) void wgrad_kernel(const FT* __restrict__ A, int lda,
                                                    const FT* __restrict__ B, int ldb, int M, int N, int K,
                                                    int SK, int kslice, float* __restrict__ part,
                                                    float* __restrict__ cs_part) {
    wgrad_body<COLSUM, KB, FT>((int)blockIdx.x, A, lda, B, ldb, M, N, K, SK, kslice, part, cs_part);
}

// two problems of the K-sliced form (KB = 4, no column sum) in ONE launch: workgroups [0, blocks0) work on problem 0, the
// rest on problem 1.  An HS layer's backward has two such products that depend only on the incoming gradient -- g^T F (conv2's
// first half) and g^T X (the STE weight) --, each too small to fill the chip on its own.
struct WgradProb { const float* A; const float* B; float* part; int lda, ldb, M, N, K, SK, kslice; };
__global__ __launch_bounds__(256) void wgrad_pair_kernel(const WgradProb p0, const WgradProb p1, int blocks0) {
    const bool second = (int)blockIdx.x >= blocks0;
    const WgradProb& p = second ? p1 : p0;
    wgrad_body<false, 4, float>((int)blockIdx.x - (second ? blocks0 : 0), p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K, p.SK, p.kslice,
                                p.part, nullptr);
}

// C[m][n] = sum_s part[s][m][n]; likewise colsum.  Workgroup = 64 float4 elements x 4 slice groups:
// group g sums slices g, g+4, ... with loads in flight, the 4 groups are folded through LDS in a
// fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int SK, int M, int N,
                                                           float* __restrict__ C, int ldc,
                                                           const float* __restrict__ cs_part,
                                                           float* __restrict__ colsum) {
    __shared__ float4 red[4][64];
    const int le = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int nq = N >> 2;
    const long long total = (long long)M * nq;                 // float4 elements of C
    const long long ncs = colsum ? (N >> 2) : 0;               // float4 elements of colsum
    const long long e = (long long)blockIdx.x * 64 + le;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = nullptr;
    size_t stride = 0;
    if (e < total) { src = part + (size_t)e * 4; stride = (size_t)M * N; }
    else if (e < total + ncs) { src = cs_part + (size_t)(e - total) * 4; stride = (size_t)N; }
    if (src) {
        float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
        int sl = sg;
        for (; sl + 4 < SK; sl += 8) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * stride);
            const float4 u = *reinterpret_cast<const float4*>(src + (size_t)(sl + 4) * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
        }
        if (sl < SK) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
    }
    red[sg][le] = s;
    __syncthreads();
    if (sg == 0 && src) {
        float4 r = red[0][le];
#pragma unroll
        for (int g = 1; g < 4; ++g) { const float4 v = red[g][le]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        if (e < total) {
            const int m = (int)(e / nq), q = (int)(e - (long long)m * nq);
            float* c = C + (size_t)m * ldc + (q << 2);
            c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w;    // ldc need not be a multiple of 4
        } else {
            *reinterpret_cast<float4*>(colsum + (size_t)(e - total) * 4) = r;
        }
    }
}

// the same fold for up to HSP_FOLD_MAX_WGRAD weight-gradient problems in ONE launch (the three parameter gradients of an HS
// layer's backward are folded together, or -- hsp_step_fold -- every pending fold of a whole backward pass).  Block ranges:
// problem p owns blocks [first[p], first[p+1]).
struct WgradFoldTab {
    int n;
    int first[HSP_FOLD_MAX_WGRAD + 1];
    HspWgradPending p[HSP_FOLD_MAX_WGRAD];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradFoldTab tab) {
    __shared__ float4 red[4][64];
    int q = 0;
    while (q + 1 < tab.n && (int)blockIdx.x >= tab.first[q + 1]) ++q;
    wgrad_fold_body(tab.p[q], (int)blockIdx.x - tab.first[q], red);
}

// every fold a backward pass left pending, in one launch: blocks [0, wfirst[nw]) fold parameter-gradient partials, the blocks
// after them fold the support-direction partials of the receptive-field layers (rfconv.hip) -- nothing before the optimizer
// reads either, so the step's ten fold launches become one
struct StepFoldTab {
    int nw, nd;
    int wfirst[HSP_FOLD_MAX_WGRAD + 1];
    int dfirst[HSP_FOLD_MAX_DIRS + 1];                      // (relative to wfirst[nw])
    HspWgradPending w[HSP_FOLD_MAX_WGRAD];
    HspDirsPending d[HSP_FOLD_MAX_DIRS];
};
__global__ __launch_bounds__(256) void step_fold_kernel(const StepFoldTab tab) {
    __shared__ __attribute__((aligned(16))) float smem[16 * 3 * 64];
    const int nwb = tab.wfirst[tab.nw];
    if ((int)blockIdx.x < nwb) {
        int q = 0;
        while (q + 1 < tab.nw && (int)blockIdx.x >= tab.wfirst[q + 1]) ++q;
        wgrad_fold_body(tab.w[q], (int)blockIdx.x - tab.wfirst[q], reinterpret_cast<float4(*)[64]>(smem));
    } else {
        const int blk = (int)blockIdx.x - nwb;
        int q = 0;
        while (q + 1 < tab.nd && blk >= tab.dfirst[q + 1]) ++q;
        const HspDirsPending pd = tab.d[q];
        dirs_fold_body<256>(reinterpret_cast<const float*>(pd.part), pd.nparts, pd.SC, reinterpret_cast<const float*>(pd.dirs),
                            reinterpret_cast<float*>(pd.grad_dirs), blk - tab.dfirst[q], reinterpret_cast<float(*)[3][64]>(smem));
    }
}

// ---- bf16 point rows on the bf16 matrix cores ---------------------------------------------------------------------------
// Same product, partial-sum layout and reduce kernel as above, for A (K x M) / B (K x N) stored in bf16 with M, N multiples
// of 128.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane, the rows are k-major: each staging thread loads an
// 8 (k) x 8 (m) block (eight 16-byte row pieces), transposes it in registers (16-bit interleaves: v_perm_b32; the 32-bit
// steps of the transpose are register renaming) and writes eight 16-byte (m, 8 k) chunks into an LDS image [m][64 k] with the
// chunk position XOR-swizzled by the row -- the image gemm_rows.hip's MFMA loop reads conflict-free with ds_read_b128.
// Workgroup: 128 x 128 output tile x one K slice, waves 2 x 2 (64 x 64 each), 64 k per block, two stages (64 KB LDS).
// The kernel is HBM-bound (one pass over A and B): 8 loads of 16 bytes per thread in flight, 2 workgroups per CU.
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;

template <bool COLSUM>
__global__ __launch_bounds__(256) void wgrad_bf16_mfma_kernel(const unsigned short* __restrict__ A, int lda,
                                                              const unsigned short* __restrict__ B, int ldb, int M, int N,
                                                              int K, int kslice, float* __restrict__ part,
                                                              float* __restrict__ cs_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OP_BYTES = 128 * 128;          // one operand image: 128 rows (m or n) x 64 k x 2 B
    constexpr int STAGE = 2 * OP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = M >> 7, tiles = tiles_m * (N >> 7);
    const int slice = blockIdx.x / tiles, t = blockIdx.x - slice * tiles;
    const int tn = t / tiles_m, tm = t - tn * tiles_m;
    const int m0 = tm << 7, n0 = tn << 7;
    const int k0 = slice * kslice, k1 = min(K, k0 + kslice);
    const int nblk = (k1 - k0 + 63) >> 6;

    // staging role: waves 0,1 transpose A blocks, waves 2,3 B blocks; block = (k octet kb8, 8-column chunk mc)
    const int half = tid >> 7, id = tid & 127;
    const int kb8 = id >> 4, mc = id & 15;
    const unsigned short* src = half ? B + n0 + mc * 8 : A + m0 + mc * 8;
    const int ld = half ? ldb : lda;
    char* const img = smem + half * OP_BYTES;

    uint4 r[8];
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    auto fetch = [&](int b) {
        const int kr = k0 + (b << 6) + (kb8 << 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = kr + j < k1 ? kr + j : k1 - 1;                  // clamped (branch-free); zeroed in stash
            r[j] = *reinterpret_cast<const uint4*>(src + (size_t)row * ld);
        }
    };
    auto stash = [&](int b, int buf) {
        const int kr = k0 + (b << 6) + (kb8 << 3);
        if (kr + 8 > k1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kr + j >= k1) r[j] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (COLSUM && half == 1 && tm == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    cs[2 * d] += __uint_as_float(w[d] << 16);
                    cs[2 * d + 1] += __uint_as_float(w[d] & 0xffff0000u);
                }
            }
        }
        // out[e][i]: row e of the block (m = 8 mc + e), k pair i (k = 2i, 2i + 1)
        unsigned o[8][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned x[4] = {r[2 * i].x, r[2 * i].y, r[2 * i].z, r[2 * i].w};
            const unsigned y[4] = {r[2 * i + 1].x, r[2 * i + 1].y, r[2 * i + 1].z, r[2 * i + 1].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                o[2 * d][i] = __builtin_amdgcn_perm(y[d], x[d], 0x05040100u);       // (x.lo16, y.lo16)
                o[2 * d + 1][i] = __builtin_amdgcn_perm(y[d], x[d], 0x07060302u);   // (x.hi16, y.hi16)
            }
        }
        char* base = img + buf * STAGE;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = (mc << 3) + e;
            *reinterpret_cast<uint4*>(base + row * 128 + ((kb8 ^ ((row >> 1) & 7)) << 4)) = make_uint4(o[e][0], o[e][1], o[e][2], o[e][3]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int key = (li >> 1) & 7;
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE + (wm0 + li) * 128;
        const char* sb = smem + buf * STAGE + OP_BYTES + (wn0 + li) * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int coff = ((2 * s + lh) ^ key) << 4;
            uint4 fa[2], fb[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) fa[x] = *reinterpret_cast<const uint4*>(sa + x * 32 * 128 + coff);
#pragma unroll
            for (int y = 0; y < 2; ++y) fb[y] = *reinterpret_cast<const uint4*>(sb + y * 32 * 128 + coff);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[x]),
                                                                        __builtin_bit_cast(bf16x8_t, fb[y]), acc[x][y], 0, 0, 0);
        }
    };

    if (nblk > 0) {
        fetch(0);
        stash(0, 0);
    }
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) fetch(b + 1);
        compute(b & 1);
        if (b + 1 < nblk) stash(b + 1, (b + 1) & 1);
        __syncthreads();
    }
    // partial tile: accumulator q of (x,y) <-> row wm0 + 32x + (q&3) + 8(q>>2) + 4 lh, column wn0 + 32y + li
    float* pc = part + (size_t)slice * M * N;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + wm0 + 32 * x + (q & 3) + 8 * (q >> 2) + 4 * lh;
                pc[(size_t)row * N + n0 + wn0 + 32 * y + li] = acc[x][y][q];
            }
    if (COLSUM && tm == 0) {
        float* red = reinterpret_cast<float*>(smem);          // [8 k octets][128 columns]  (all stages are dead here)
        if (half == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[kb8 * 128 + mc * 8 + e] = cs[e];
        }
        __syncthreads();
        if (tid < 128) {
            float v = red[tid];
#pragma unroll
            for (int o8 = 1; o8 < 8; ++o8) v += red[o8 * 128 + tid];
            cs_part[(size_t)slice * N + n0 + tid] = v;
        }
    }
}

// slices for the bf16-MFMA form: ~512 workgroups, >= 4 blocks of 64 rows per slice, <= 128 partial copies
// ---- fp32 point rows on the BF16 matrix cores, fp32-accurate: C = A^T B from exact three-way bf16 splits ------------------------
// (the weight-gradient twin of csrc/gemm_x3.hip: x = hi + mid + lo by truncation, six exact slice products per k accumulated
// in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are < 2^-23 |a||b| per product.)  wgrad_kernel holds 135 TFLOP/s = 87 %
// of the fp32-MFMA peak on the two large shapes of the step -- it IS matrix-core-bound -- and six bf16 MFMAs per 16 k take 192
// clocks where the fp32 form takes 512.  128 x 128 tile x K slice per workgroup; a staging thread loads 8 (k) x 4 (m) fp32
// (eight 16-byte row pieces, 512 contiguous bytes per k row and half-wave), splits them and writes, per column and plane, the
// 16-byte (m, 8 k) chunk the MFMA wants -- the k-major -> m-major transpose costs nothing extra because every fp32 sits in its own
// register (one v_perm_b32 per bf16 pair).  LDS: 6 planes of 128 rows x 80 bytes (64 of k + 16 of pad: the 16 lanes of a
// ds_read_b128 group hit 16 disjoint bank quads without a swizzle), one stage, two workgroups per CU.  Measured (B=16 N=1028):
// 128x1024 <- 16448 rows 34.8 us (the fp32-MFMA kernel: 31.4 -- both read 134 MB, i.e. ~4 TB/s: the shape is bound by its
// operand traffic, not by the matrix cores), 256x2048 <- 4112 22 us (31.4); the step -17 us.  Two staging sets (loads two blocks
// ahead, 226 VGPRs) change nothing.
template <bool COLSUM>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                          int M, int N, int K, int kslice, float* __restrict__ part,
                                                          float* __restrict__ cs_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCH = 80, PLANE = 128 * PITCH;             // bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // M need not be a multiple of 128 (nor of 4: the K = 1286 / 1289 / 771 inputs of the heads' first layers, PoseR.py:27,
    // PoseTs.py:32, FaceRecon.py:38,116): the last row tile reads only the column quads below ceil4(M) <= lda -- the operand's rows
    // sit on a 16-byte pitch (ops.assemble_feat / cat_rows_pitched) -- the quads above it enter as zeros, and the partial tile keeps
    // its rows below M.  N (the other operand's width) stays a multiple of 128.
    const int tiles_m = (M + 127) >> 7, tiles = tiles_m * (N >> 7);
    const int slice = blockIdx.x / tiles, t = blockIdx.x - slice * tiles;
    const int tn = t / tiles_m, tm = t - tn * tiles_m;
    const int m0 = tm << 7, n0 = tn << 7;
    const int k0 = slice * kslice, k1 = min(K, k0 + kslice);
    const int nblk = (k1 - k0 + 31) >> 5;

    // staging role: waves 0,1 take A, waves 2,3 take B; thread = (k octet kb8 of the 32-row block, column quad mc)
    const int half = tid >> 7, id = tid & 127;
    const int kb8 = id >> 5, mc = id & 31;
    const float* src = half ? B + n0 + mc * 4 : A + m0 + mc * 4;
    const int ld = half ? ldb : lda;
    const bool live = half || m0 + mc * 4 < M;                 // (a column quad of A past the ragged edge)
    char* const img = smem + half * 3 * PLANE;

    float4 r[8];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int b) {
        const int kr = k0 + (b << 5) + (kb8 << 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = kr + j < k1 ? kr + j : k1 - 1;                  // clamped (branch-free); zeroed in stash
            r[j] = live ? *reinterpret_cast<const float4*>(src + (size_t)row * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int b) {
        const int kr = k0 + (b << 5) + (kb8 << 3);
        if (kr + 8 > k1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kr + j >= k1) r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (COLSUM && half == 1 && tm == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs[0] += r[j].x; cs[1] += r[j].y; cs[2] += r[j].z; cs[3] += r[j].w; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[8], r1[8], r2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = e == 0 ? r[j].x : e == 1 ? r[j].y : e == 2 ? r[j].z : r[j].w;
                const float h = __uint_as_float(__float_as_uint(v[j]) & 0xffff0000u);
                r1[j] = v[j] - h;
                const float mm = __uint_as_float(__float_as_uint(r1[j]) & 0xffff0000u);
                r2[j] = r1[j] - mm;
            }
            uint4 ch, cm, cl;
            unsigned* ph = reinterpret_cast<unsigned*>(&ch);
            unsigned* pm = reinterpret_cast<unsigned*>(&cm);
            unsigned* pl = reinterpret_cast<unsigned*>(&cl);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ph[i] = __builtin_amdgcn_perm(__float_as_uint(v[2 * i + 1]), __float_as_uint(v[2 * i]), 0x07060302u);
                pm[i] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * i + 1]), __float_as_uint(r1[2 * i]), 0x07060302u);
                pl[i] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * i + 1]), __float_as_uint(r2[2 * i]), 0x07060302u);
            }
            char* dst = img + ((mc << 2) + e) * PITCH + (kb8 << 4);
            *reinterpret_cast<uint4*>(dst) = ch;
            *reinterpret_cast<uint4*>(dst + PLANE) = cm;
            *reinterpret_cast<uint4*>(dst + 2 * PLANE) = cl;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    auto compute = [&]() {
        const char* sa = smem + (wm0 + li) * PITCH;
        const char* sb = smem + 3 * PLANE + (wn0 + li) * PITCH;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int coff = (2 * s2 + lh) << 4;
            uint4 fa[2][3], fb[2][3];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[x][p] = *reinterpret_cast<const uint4*>(sa + p * PLANE + x * 32 * PITCH + coff);
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[y][p] = *reinterpret_cast<const uint4*>(sb + p * PLANE + y * 32 * PITCH + coff);
            constexpr int SA[6] = {0, 2, 1, 0, 1, 0}, SB[6] = {2, 0, 1, 1, 0, 0};     // small terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[x][SA[q]]),
                                                                            __builtin_bit_cast(bf16x8_t, fb[y][SB[q]]), acc[x][y], 0, 0, 0);
        }
    };

    if (nblk > 0) fetch(0);
    for (int b = 0; b < nblk; ++b) {
        __syncthreads();                       // the previous block's fragment reads are done
        stash(b);
        __syncthreads();
        if (b + 1 < nblk) fetch(b + 1);
        compute();
    }
    float* pc = part + (size_t)slice * M * N;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + wm0 + 32 * x + (q & 3) + 8 * (q >> 2) + 4 * lh;
                if (row < M) pc[(size_t)row * N + n0 + wn0 + 32 * y + li] = acc[x][y][q];
            }
    if (COLSUM && tm == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [4 k octets][128 columns]  (the stage is dead here)
        if (half == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[kb8 * 128 + mc * 4 + e] = cs[e];
        }
        __syncthreads();
        if (tid < 128) cs_part[(size_t)slice * N + n0 + tid] = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
    }
}

static bool wgrad_x3_ok(const void* A, int lda, const void* B, int ldb, int M, int N) {
    // (>= 4 output tiles: a single 128 x 128 tile leaves the K slices as the only parallelism -- 65 workgroups at K = 16448 --
    // and the fp32 kernel's 4-slices-per-workgroup form is faster there: 9.8 vs 18.3 us)
    // (ragged M: the last column quad of A is read whole, so its rows must reach ceil4(M))
    return (N & 127) == 0 && ((M + 127) >> 7) * (N >> 7) >= 4 && (lda & 3) == 0 && (ldb & 3) == 0 && lda >= ((M + 3) & ~3) &&
           ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(B)) & 15) == 0;
}
// shapes only the x3 form takes: M not a multiple of 64 (the heads' K = 1286 / 1289 / 771 first layers)
static bool wgrad_ragged_m(int M, int N) { return (M & 63) != 0 && (N & 127) == 0 && M >= 128; }

static int wgrad_bf16_pick(int M, int N, int K, int* kslice) {
    const int tiles = ((M + 127) >> 7) * (N >> 7);
    int sk = tiles > 0 ? (512 + tiles - 1) / tiles : 1;
    const int max_sk = (K + 255) / 256;
    if (sk > max_sk) sk = max_sk;
    if (sk > 128) sk = 128;
    if (sk < 1) sk = 1;
    // two workgroups per CU: 512 run at once.  A tile count that does not divide 512 (the ragged shapes of the heads: 88, 44, 28
    // tiles) would put 528 / 532 workgroups in TWO rounds with the rule above -- one slice fewer keeps it to one, longer, round
    // (1286 x 1024 <- 16448: 385 -> 2xx us).  Cost = rounds x rows per slice; the rule above wins ties (every shape whose
    // tile count divides 512 keeps its slices, so its bits).
    auto ksof = [&](int s_) { int k_ = (K + s_ - 1) / s_; return (k_ + 63) / 64 * 64; };
    auto cost = [&](int s_) { return (long long)((tiles * s_ + 511) / 512) * ksof(s_); };
    int best = sk;
    for (int c = sk - 1; c >= 1 && c >= sk - 2; --c)
        if (cost(c) < cost(best)) best = c;
    sk = best;
    const int ks = ksof(sk);
    *kslice = ks;
    return (K + ks - 1) / ks;
}
static bool wgrad_bf16_mfma_ok(const void* A, int lda, const void* B, int ldb, int M, int N) {
    return (M & 127) == 0 && (N & 127) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 &&
           ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(B)) & 15) == 0;
}

// (slices, rows per slice, waves of a workgroup along K)
static int wgrad_pick_sk(int M, int N, int K, int* kslice, int* kblock) {
    const int tiles = (M >> 6) * (N >> 6);
    constexpr int G = 4 * WG_UNROLL;                           // whole pairs of prefetch groups
    if (tiles <= 64) {
        // few tiles: 4 K slices per workgroup, folded in LDS; >= 32 rows per slice, <= 64 partial sums
        int sk = 2048 / tiles;
        if (sk > K / 32) sk = K / 32;
        if (sk > 256) sk = 256;
        if (sk >= 8) {
            int ks = (K + sk - 1) / sk;
            ks = (ks + G - 1) / G * G;
            sk = (K + ks - 1) / ks;
            sk = (sk + 3) & ~3;
            *kslice = ks;
            *kblock = 4;
            return sk;
        }
    }
    int sk = (2048 + tiles - 1) / tiles;                       // ~2 waves per SIMD
    int max_sk = (K + 127) / 128;                              // at least 128 rows per slice
    if (max_sk > 64) max_sk = 64;                              // bound the partial-sum traffic
    if (sk > max_sk) sk = max_sk;
    if (sk < 1) sk = 1;
    int ks = (K + sk - 1) / sk;
    ks = (ks + G - 1) / G * G;
    sk = (K + ks - 1) / ks;
    *kslice = ks;
    *kblock = 1;
    return sk;
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_wgrad_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int ks, kb;
    if (wgrad_ragged_m(M, N)) {                                // (x3 form only)
        const size_t p2 = (size_t)wgrad_bf16_pick(M, N, K, &ks);
        return p2 * ((size_t)M * N + N) * sizeof(float);
    }
    const int sk = wgrad_pick_sk(M, N, K, &ks, &kb);
    size_t parts = (size_t)(sk / kb);
    if ((M & 127) == 0 && (N & 127) == 0) {                    // the bf16-MFMA form may cut K finer
        int ks2;
        const size_t p2 = (size_t)wgrad_bf16_pick(M, N, K, &ks2);
        if (p2 > parts) parts = p2;
    }
    return parts * ((size_t)M * N + N) * sizeof(float);
}

template <bool COLSUM, typename FT>
static int wgrad_launch(const FT* A, int lda, const FT* B, int ldb, int M, int N, int K, int sk, int ks, int kb,
                        float* part, float* cs_part, hipStream_t st) {
    const int tiles = (M >> 6) * (N >> 6);
    if (kb == 1) {
        hipLaunchKernelGGL((wgrad_kernel<COLSUM, 1, FT>), dim3((tiles * sk + 3) / 4), dim3(256), 0, st, A, lda, B, ldb, M, N, K,
                           sk, ks, part, cs_part);
        return check_launch();
    }
    const size_t lds = (size_t)4 * 4 * 16 * 64 * sizeof(float) + (size_t)4 * 64 * sizeof(float2);
    auto kern = wgrad_kernel<COLSUM, 4, FT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    hipLaunchKernelGGL(kern, dim3(tiles * (sk / 4)), dim3(256), lds, st, A, lda, B, ldb, M, N, K, sk, ks, part, cs_part);
    return check_launch();
}

template <typename FT>
static int wgrad_impl(const FT* A, int lda, const FT* B, int ldb, int M, int N, int K, float* C, int ldc,
                      float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream, HspWgradPending* pending = nullptr) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < M || ldb < N || ldc < N) return HSP_ERR_BAD_ARG;
    const bool ragged = sizeof(FT) == 4 && wgrad_ragged_m(M, N);                       // fp32 rows, x3 form only
    if (ragged && !wgrad_x3_ok(A, lda, B, ldb, M, N)) return HSP_ERR_UNSUPPORTED;
    if (!ragged && ((M & 63) || (N & 63) || (lda & 1) || (ldb & 1))) return HSP_ERR_UNSUPPORTED;   // 64x64 wave tiles, float2 loads
    if (!ws || ws_bytes < hsp_wgrad_workspace_bytes(M, N, K)) return HSP_ERR_WORKSPACE;
    int ks, kb;
    hipStream_t st = as_stream(stream);
    if constexpr (sizeof(FT) == 2) {
        if (wgrad_bf16_mfma_ok(A, lda, B, ldb, M, N)) {
            const int sk2 = wgrad_bf16_pick(M, N, K, &ks);
            float* part = reinterpret_cast<float*>(ws);
            float* cs_part = part + (size_t)sk2 * M * N;
            const int grid = (M >> 7) * (N >> 7) * sk2;
            const int lds = 4 * 128 * 128;
            const unsigned short* a = reinterpret_cast<const unsigned short*>(A);
            const unsigned short* b = reinterpret_cast<const unsigned short*>(B);
#define WG_BF16_LAUNCH(CS)                                                                                                      \
    do {                                                                                                                       \
        auto kern = wgrad_bf16_mfma_kernel<CS>;                                                                                \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                             \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, lda, b, ldb, M, N, K, ks, part, cs_part);                   \
    } while (0)
            if (colsum_B) WG_BF16_LAUNCH(true); else WG_BF16_LAUNCH(false);
#undef WG_BF16_LAUNCH
            int rc = check_launch();
            if (rc) return rc;
            if (pending) { *pending = HspWgradPending{part, cs_part, C, colsum_B, sk2, M, N, ldc}; return HSP_OK; }
            const long long total = (long long)M * (N >> 2) + (colsum_B ? (N >> 2) : 0);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, sk2, M, N, C, ldc,
                               cs_part, colsum_B);
            return check_launch();
        }
    }
    if constexpr (sizeof(FT) == 4) {
        if (wgrad_x3_ok(A, lda, B, ldb, M, N)) {
            const int sk2 = wgrad_bf16_pick(M, N, K, &ks);
            float* part = reinterpret_cast<float*>(ws);
            float* cs_part = part + (size_t)sk2 * M * N;
            const int grid = ((M + 127) >> 7) * (N >> 7) * sk2;
            const int lds = 6 * 128 * 80;
            if (colsum_B) hipLaunchKernelGGL(wgrad_x3_kernel<true>, dim3(grid), dim3(256), lds, st, A, lda, B, ldb, M, N, K, ks, part, cs_part);
            else hipLaunchKernelGGL(wgrad_x3_kernel<false>, dim3(grid), dim3(256), lds, st, A, lda, B, ldb, M, N, K, ks, part, cs_part);
            int rc = check_launch();
            if (rc) return rc;
            if (pending) { *pending = HspWgradPending{part, cs_part, C, colsum_B, sk2, M, N, ldc}; return HSP_OK; }
            const long long total = (long long)M * (N >> 2) + (colsum_B ? (N >> 2) : 0);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, sk2, M, N, C, ldc,
                               cs_part, colsum_B);
            return check_launch();
        }
    }
    const int sk = wgrad_pick_sk(M, N, K, &ks, &kb);
    const int nparts = sk / kb;
    float* part = reinterpret_cast<float*>(ws);
    float* cs_part = part + (size_t)nparts * M * N;
    int rc = colsum_B ? wgrad_launch<true, FT>(A, lda, B, ldb, M, N, K, sk, ks, kb, part, cs_part, st)
                      : wgrad_launch<false, FT>(A, lda, B, ldb, M, N, K, sk, ks, kb, part, cs_part, st);
    if (rc) return rc;
    if (pending) { *pending = HspWgradPending{part, cs_part, C, colsum_B, nparts, M, N, ldc}; return HSP_OK; }
    const long long total = (long long)M * (N >> 2) + (colsum_B ? (N >> 2) : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, nparts, M, N, C,
                       ldc, cs_part, colsum_B);
    return check_launch();
}

extern "C" int hsp_wgrad_f32(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                             float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream) {
    return wgrad_impl<float>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream);
}
/* bf16 point rows in, fp32 parameter gradient out */
extern "C" int hsp_wgrad_bf16(const hsp_bf16_t* A, int lda, const hsp_bf16_t* B, int ldb, int M, int N, int K, float* C,
                              int ldc, float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream) {
    return wgrad_impl<bf16_t>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream);
}

/* split forms: the partial-sum launch only (the fold is left pending), and one fold launch for up to 4 pending problems */
extern "C" int hsp_wgrad_partial_f32(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                                     float* colsum_B, void* ws, size_t ws_bytes, HspWgradPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    return wgrad_impl<float>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream, pending);
}
extern "C" int hsp_wgrad_partial_bf16(const hsp_bf16_t* A, int lda, const hsp_bf16_t* B, int ldb, int M, int N, int K, float* C,
                                      int ldc, float* colsum_B, void* ws, size_t ws_bytes, HspWgradPending* pending,
                                      hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    return wgrad_impl<bf16_t>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream, pending);
}
/* two weight gradients that share nothing but the launch: both K-sliced (few output tiles), no column sums; any other pair is
 * issued as two hsp_wgrad_partial_f32 launches */
extern "C" int hsp_wgrad_partial_pair_f32(const float* A0, int lda0, const float* B0, int ldb0, int M0, int N0, int K0, float* C0,
                                          int ldc0, void* ws0, size_t ws_bytes0, const float* A1, int lda1, const float* B1,
                                          int ldb1, int M1, int N1, int K1, float* C1, int ldc1, void* ws1, size_t ws_bytes1,
                                          HspWgradPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    int ks0, kb0, ks1, kb1;
    const bool shapes_ok = A0 && B0 && C0 && A1 && B1 && C1 && M0 > 0 && N0 > 0 && K0 > 0 && M1 > 0 && N1 > 0 && K1 > 0 &&
                           !((M0 | N0 | M1 | N1) & 63) && !((lda0 | ldb0 | lda1 | ldb1) & 1) && lda0 >= M0 && ldb0 >= N0 &&
                           lda1 >= M1 && ldb1 >= N1 && ldc0 >= N0 && ldc1 >= N1;
    if (shapes_ok) {
        int sk0 = wgrad_pick_sk(M0, N0, K0, &ks0, &kb0), sk1 = wgrad_pick_sk(M1, N1, K1, &ks1, &kb1);
        if (kb0 == 4 && kb1 == 4) {
            // the two problems share ONE grid and the kernel's 66 KB of LDS let two workgroups live on a CU: 512 at once.  Each
            // problem alone is sized for ~512 workgroups, so the pair came to 528 / 704 / 768 -- a second, mostly empty round that
            // costs a whole slice time.  Fewer, longer K slices keep the pair inside one round (same kernel, same fold; the
            // workspace rule is unchanged: fewer partials than it allows for).
            constexpr int G = 4 * WG_UNROLL;
            const int t0 = (M0 >> 6) * (N0 >> 6), t1 = (M1 >> 6) * (N1 >> 6);
            int total = t0 * (sk0 / 4) + t1 * (sk1 / 4);
            for (int it = 0; it < 8 && total > 2 * HSP_NUM_CU && sk0 > 8 && sk1 > 8; ++it) {
                const double f = (double)(2 * HSP_NUM_CU) / total;
                auto shrink = [&](int K, int& sk, int& ks) {
                    int want = (int)(sk * f) & ~3;
                    if (want >= sk) want = sk - 4;
                    if (want < 8) want = 8;
                    ks = ((K + want - 1) / want + G - 1) / G * G;
                    sk = (((K + ks - 1) / ks) + 3) & ~3;
                };
                shrink(K0, sk0, ks0);
                shrink(K1, sk1, ks1);
                total = t0 * (sk0 / 4) + t1 * (sk1 / 4);
            }
        }
        if (kb0 == 4 && kb1 == 4 && ws0 && ws1 && ws_bytes0 >= hsp_wgrad_workspace_bytes(M0, N0, K0) &&
            ws_bytes1 >= hsp_wgrad_workspace_bytes(M1, N1, K1)) {
            const WgradProb p0{A0, B0, reinterpret_cast<float*>(ws0), lda0, ldb0, M0, N0, K0, sk0, ks0};
            const WgradProb p1{A1, B1, reinterpret_cast<float*>(ws1), lda1, ldb1, M1, N1, K1, sk1, ks1};
            const int blocks0 = (M0 >> 6) * (N0 >> 6) * (sk0 / 4), blocks1 = (M1 >> 6) * (N1 >> 6) * (sk1 / 4);
            const size_t lds = (size_t)4 * 4 * 16 * 64 * sizeof(float) + (size_t)4 * 64 * sizeof(float2);
            static bool attr_set = false;
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pair_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
                attr_set = true;
            }
            hipLaunchKernelGGL(wgrad_pair_kernel, dim3(blocks0 + blocks1), dim3(256), lds, as_stream(stream), p0, p1, blocks0);
            int rc = check_launch();
            if (rc) return rc;
            pending[0] = HspWgradPending{p0.part, nullptr, C0, nullptr, sk0 / 4, M0, N0, ldc0};
            pending[1] = HspWgradPending{p1.part, nullptr, C1, nullptr, sk1 / 4, M1, N1, ldc1};
            return HSP_OK;
        }
    }
    int rc = wgrad_impl<float>(A0, lda0, B0, ldb0, M0, N0, K0, C0, ldc0, nullptr, ws0, ws_bytes0, stream, pending);
    if (rc) return rc;
    return wgrad_impl<float>(A1, lda1, B1, ldb1, M1, N1, K1, C1, ldc1, nullptr, ws1, ws_bytes1, stream, pending + 1);
}

extern "C" int hsp_wgrad_fold(const HspWgradPending* pending, int n, hspStream_t stream) {
    if (!pending || n <= 0 || n > HSP_FOLD_MAX_WGRAD) return HSP_ERR_BAD_ARG;
    WgradFoldTab tab;
    tab.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const HspWgradPending& p = pending[i];
        if (!p.part || !p.C || p.nparts <= 0 || p.M <= 0 || p.N <= 0 || (p.N & 3) || p.ldc < p.N || (p.colsum && !p.cs_part))
            return HSP_ERR_BAD_ARG;
        tab.first[i] = blocks;
        const long long total = (long long)p.M * (p.N >> 2) + (p.colsum ? (p.N >> 2) : 0);
        blocks += (int)((total + 63) / 64);
        tab.p[i] = p;
    }
    tab.first[n] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tab);
    return check_launch();
}

static bool wgrad_pending_ok(const HspWgradPending& p) {
    return p.part && p.C && p.nparts > 0 && p.M > 0 && p.N > 0 && !(p.N & 3) && p.ldc >= p.N && !(p.colsum && !p.cs_part);
}

extern "C" int hsp_step_fold(const HspWgradPending* wgrads, int nw, const HspDirsPending* dirs, int nd, hspStream_t stream) {
    if (nw < 0 || nd < 0 || nw > HSP_FOLD_MAX_WGRAD || nd > HSP_FOLD_MAX_DIRS || (nw && !wgrads) || (nd && !dirs)) return HSP_ERR_BAD_ARG;
    if (nw + nd == 0) return HSP_OK;
    StepFoldTab tab;
    tab.nw = nw; tab.nd = nd;
    int blocks = 0;
    for (int i = 0; i < nw; ++i) {
        const HspWgradPending& p = wgrads[i];
        if (!wgrad_pending_ok(p)) return HSP_ERR_BAD_ARG;
        tab.wfirst[i] = blocks;
        const long long total = (long long)p.M * (p.N >> 2) + (p.colsum ? (p.N >> 2) : 0);
        blocks += (int)((total + 63) / 64);
        tab.w[i] = p;
    }
    tab.wfirst[nw] = blocks;
    int dblocks = 0;
    for (int i = 0; i < nd; ++i) {
        const HspDirsPending& p = dirs[i];
        if (!p.part || !p.dirs || !p.grad_dirs || p.nparts <= 0 || p.SC <= 0) return HSP_ERR_BAD_ARG;
        tab.dfirst[i] = dblocks;
        dblocks += (p.SC + 63) / 64;
        tab.d[i] = p;
    }
    tab.dfirst[nd] = dblocks;
    hipLaunchKernelGGL(step_fold_kernel, dim3((unsigned)(blocks + dblocks)), dim3(256), 0, as_stream(stream), tab);
    return check_launch();
}
