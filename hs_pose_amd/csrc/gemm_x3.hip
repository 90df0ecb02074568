// gemm_x3.hip -- the fp32 dense per-point products of the HS stack and of the heads on the BF16 matrix cores of gfx950,
// without giving up fp32 accuracy: every fp32 operand is split EXACTLY into three bf16 slices
//       x = hi + mid + lo,   hi = x & 0xffff0000,  mid = (x - hi) & 0xffff0000,  lo = (x - hi) - mid
// (truncations, so each difference is exact and lo has at most 8 significant bits: a bf16), and a product row is
//       a.b ~= sum_k  ah.bh + ah.bm + am.bh + am.bm + ah.bl + al.bh
// six slice products, each EXACT in fp32 (8 x 8 significant bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three
// dropped terms (am.bl, al.bm, al.bl) are below 2^-23 |a||b| per product -- the size of the rounding an fp32 fma chain commits
// on every step.  Measured against fp64 (tests/test_gpu_gemm_x3.py) the result is as close as the fp32-MFMA kernels'.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s), a sixteenth of the bf16 rate; six bf16 MFMAs per 16 k
// take 192 clocks where the fp32 form takes 512.  The dense products then stop being MFMA-bound (the K = 128 products become
// bound by writing their output, the deep-K input-gradient products by streaming gfm once).
//
//   C[r][n] = alpha * ( sum_k A1[r][k] W1[n][k]  (+ sum_k A2[r][k] W2[n][k]) ) (+ bias[n]) (+ resid[r][n]) (+ cloud_bias[r / rpc][n])
//
// the contract of gemm_rows.hip (reference network/fs_net_repo/gcn3d.py:149,171,186 and their input gradients; the Conv1d(k=1)
// layers of the heads, PoseR.py:16-39, PoseTs.py:18-45, FaceRecon.py:37-68) with the weight-side operand handed over ALREADY
// SPLIT, in (N, K) form: three bf16 planes written once per step by split_params_x3_kernel (one launch for every weight of the
// network, transposing where the product wants W^T).  The activation-side operand is split in the staging path of the kernel.
//
// Structure: 256-thread workgroup = 2 x 2 waves on a (64 WM) x 128 tile, K in blocks of 32; per block a thread fetches 8 WM
// fp32 of A (split into 3 x 16 bytes) and six 16-byte pieces of the weight planes, all held in registers across the previous
// block's MFMAs (global loads one block ahead), then written to a single LDS stage: rows of 64 bytes (32 bf16), the four
// 16-byte chunks of row r stored at chunk ^ ((r >> 2) & 3) -- conflict-free for the 16-lane groups of ds_read_b128.  Two
// workgroups per CU alternate staging and MFMA phases.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct X3Args {
    const float* A[2];                     // activation rows (M, K) fp32, row pitch lda (a multiple of 4)
    const unsigned short* P[2];            // weight planes: 3 x (N, ldp) bf16, plane p at P + p * ps; zero beyond K
    int lda[2], ldp[2], K[2];
    long long ps[2];
    float* C; int ldc;
    int M, N;
    const float* bias;
    const float* resid; int ldr;
    const float* cbias; int rpc;
    float alpha;
    int tiles_m, tiles_n, nsplit;
    float* ws;                             // split-K partial tiles ws[split][M][N]
    float* bn_shift;                       // EPI bit 3: per-column shift (N), WRITTEN here: resid[0][n] + cloud_bias[0][n] ...
    float* bn_part;                        // ... and the BatchNorm partial sums [tile_m][2][N] of the RESULT rows
};

#define X3_BN 128
#define X3_BK 32

// 8 consecutive fp32 of a row -> the 16-byte bf16 chunks of the three planes
__device__ __forceinline__ void x3_split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float h = __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
        r1[i] = x[i] - h;                                                  // exact
        const float m = __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
        r2[i] = r1[i] - m;                                                 // exact, <= 8 significant bits
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                          // v_perm_b32: high halves of two dwords
        hi[i] = __builtin_amdgcn_perm(__float_as_uint(x[2 * i + 1]), __float_as_uint(x[2 * i]), 0x07060302u);
        mid[i] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * i + 1]), __float_as_uint(r1[2 * i]), 0x07060302u);
        lo[i] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * i + 1]), __float_as_uint(r2[2 * i]), 0x07060302u);
    }
}

// WM: 32-row blocks per wave along M (tile = 64 WM x 128).  TWO: second source.  EPI: bit 0 bias, bit 1 residual, bit 2
// per-cloud bias (template parameters: a load inside a run-time branch costs a drained queue at the join)
// (measured and removed in round 3: a short-K form that fetched every activation block of a tile up front -- 1.93 vs 1.92 ms per
// step: the tile's time is the weight planes' round trips as much as the activations')
template <int WM, bool TWO, int EPI, int PA = 0>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const X3Args g) {
    constexpr int BM = 64 * WM;
    constexpr int A_PLANE = BM * 64, B_PLANE = X3_BN * 64;                 // bytes
    constexpr int NAU = WM;                                                // 8-float A units per thread and block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                                       // 3 planes
    char* sB = smem + 3 * A_PLANE;                                         // 3 planes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 64;

    // ---- item: (tile, k split); tiles ordered tn fastest (the workgroups in flight share A rows in L2)
    const int item = blockIdx.x;
    const int o = item / g.nsplit, ks = item - o * g.nsplit;
    const int tm = o / g.tiles_n, tn = o - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * X3_BN;
    const int T0 = (g.K[0] + X3_BK - 1) / X3_BK;
    const int T1 = TWO ? (g.K[1] + X3_BK - 1) / X3_BK : 0;
    const int TT = T0 + T1;
    const int per = (TT + g.nsplit - 1) / g.nsplit;
    const int t_begin = ks * per, t_end = min(TT, t_begin + per);

    f32x16 acc[WM][2];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    if constexpr (EPI == 2) {
        // residual ONLY (alpha == 1; the input-gradient chain of layers that share their input rows): the accumulators START from
        // the residual -- its loads land in the registers the tile owns anyway and fly under the first blocks' staging, where an
        // epilogue read costs its own registers (256 VGPRs = one workgroup per CU fewer) and an exposed round trip per row group
        // (measured 324 vs 231 us on 16448 x 1286 <- 1024)
        // (buffer loads: rows past M fall outside the descriptor and read 0, columns past N are sent there; one vector offset per
        // column block, the row steps are scalar offsets -- no per-element address or select registers)
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.resid), 0, (int)((size_t)g.M * g.ldr * 4), 0x00020000);
        const unsigned ldr4 = (unsigned)g.ldr * 4u;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = n0 + wn0 + 32 * y + li;
            const unsigned voff = col < g.N ? (unsigned)(m0 + wm0 + 4 * lh) * ldr4 + (unsigned)col * 4u : 0xfffffff0u;
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[x][y][r] = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(rsR, voff, (unsigned)(32 * x + (r & 3) + 8 * (r >> 2)) * ldr4, 0));
        }
    }

    // ---- staging registers of one block (ONE set, loads one block ahead.  Two sets / two blocks ahead were measured: the
    // 64-row kernels went from 112-128 to 146-158 VGPRs = one wave per SIMD fewer, and the step from 2.03 to 2.06 ms -- these
    // products are bound by memory latency against the workgroups resident per CU, not by the loads in flight per workgroup)
    u32x4 bv[6];
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A[0]), 0, (int)((((size_t)g.M - 1) * g.lda[0] + g.K[0]) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(TWO ? g.A[1] : g.A[0]), 0, (int)((((size_t)g.M - 1) * g.lda[TWO ? 1 : 0] + g.K[TWO ? 1 : 0]) * 4), 0x00020000);
    // A through a buffer descriptor: no branch around the loads (a load inside a run-time branch makes hipcc drain the queue at
    // the join); rows past M and bytes past the last row read 0, a ragged K is masked when the block is split (stashA)
    auto fetchA = [&](float (&dst)[NAU][8], int t) {
        const int src = (TWO && t >= T0) ? 1 : 0;
        const int kb = (src ? t - T0 : t) * X3_BK;
        const __amdgpu_buffer_rsrc_t ra = src ? rsA1 : rsA0;
        const unsigned lda = (unsigned)g.lda[src];
#pragma unroll
        for (int i = 0; i < NAU; ++i) {
            const int u = tid + 256 * i, row = u >> 2, c8 = u & 3;
            const int gr = m0 + row, k0 = kb + 8 * c8;
            const unsigned off = gr < g.M ? ((unsigned)gr * lda + (unsigned)k0) * 4u : 0xfffffff0u;
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(ra, off, 16, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dst[i][e] = __uint_as_float(v0[e]); dst[i][4 + e] = __uint_as_float(v1[e]); }
        }
    };
    auto fetchB = [&](int t) {
        const int src = (TWO && t >= T0) ? 1 : 0;
        const int kb = (src ? t - T0 : t) * X3_BK;
        const unsigned short* P = g.P[src];
        const int ldp = g.ldp[src];
        const long long ps = g.ps[src];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = tid + 256 * i, plane = q >> 9, w = q & 511, row = w >> 2, c = w & 3;
            bv[i] = *reinterpret_cast<const u32x4*>(P + plane * ps + (size_t)min(n0 + row, g.N - 1) * ldp + kb + 8 * c);
        }
    };
    auto stashA = [&](float (&srcv)[NAU][8], int t) {
        {   // ragged K: the last block of a source may reach past K (into the next row's bytes): zero those elements
            const int src = (TWO && t >= T0) ? 1 : 0;
            const int kb = (src ? t - T0 : t) * X3_BK, K = g.K[src];
            if (kb + X3_BK > K) {
#pragma unroll
                for (int i = 0; i < NAU; ++i) {
                    const int k0 = kb + 8 * ((tid + 256 * i) & 3);
#pragma unroll
                    for (int e = 0; e < 8; ++e) srcv[i][e] = k0 + e < K ? srcv[i][e] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NAU; ++i) {
            const int u = tid + 256 * i, row = u >> 2, c8 = u & 3;
            u32x4 h, m, l;
            x3_split8(srcv[i], h, m, l);
            const int off = row * 64 + ((c8 ^ ((row >> 2) & 3)) << 4);
            *reinterpret_cast<u32x4*>(sA + off) = h;
            *reinterpret_cast<u32x4*>(sA + A_PLANE + off) = m;
            *reinterpret_cast<u32x4*>(sA + 2 * A_PLANE + off) = l;
        }
    };
    auto stashB = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = tid + 256 * i, plane = q >> 9, w = q & 511, row = w >> 2, c = w & 3;
            *reinterpret_cast<u32x4*>(sB + plane * B_PLANE + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = bv[i];
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[WM][3], fb[2][3];
            const int c = 2 * s + lh;
#pragma unroll
            for (int x = 0; x < WM; ++x) {
                const int row = wm0 + 32 * x + li;
                const int off = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[x][p] = *reinterpret_cast<const u32x4*>(sA + p * A_PLANE + off);
            }
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int row = wn0 + 32 * y + li;
                const int off = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[y][p] = *reinterpret_cast<const u32x4*>(sB + p * B_PLANE + off);
            }
            // six slice products, small terms first: (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
            constexpr int SA[6] = {0, 2, 1, 0, 1, 0}, SB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[x][SA[q]]),
                                                                            __builtin_bit_cast(bf16x8, fb[y][SB[q]]), acc[x][y], 0, 0, 0);
        }
    };
    {
        float av[NAU][8];
        if (t_begin < t_end) { fetchA(av, t_begin); fetchB(t_begin); }
        for (int t = t_begin; t < t_end; ++t) {
            __syncthreads();                   // the previous block's fragment reads are done
            stashA(av, t);
            stashB();
            __syncthreads();
            if (t + 1 < t_end) { fetchA(av, t + 1); fetchB(t + 1); }   // in flight under this block's MFMAs
            mma();
        }
    }

    // ---- epilogue: accumulator r of (x, y) <-> row m0 + wm0 + 32 x + (r & 3) + 8 (r >> 2) + 4 lh, column n0 + wn0 + 32 y + li
    if (g.nsplit > 1) {
        float* wsp = g.ws + (size_t)ks * g.M * g.N;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = n0 + wn0 + 32 * y + li;
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.M && col < g.N) wsp[(size_t)row * g.N + col] = acc[x][y][r];
                }
        }
        return;
    }
    constexpr bool HAS_BIAS = EPI & 1, HAS_RES = (EPI & 2) && EPI != 2, HAS_CB = EPI & 4, HAS_BN = EPI & 8;
    float bn1[2] = {0.f, 0.f}, bn2[2] = {0.f, 0.f};           // per column of this lane: sum (v - shift), sum (v - shift)^2
    // per-cloud bias: a tile of BM rows spans at most two clouds when rows_per_cloud >= BM (boundary compare, no division)
    const int c0 = m0 / g.rpc, nb = (c0 + 1) * g.rpc;
    const bool two_clouds = g.rpc >= BM;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int col = n0 + wn0 + 32 * y + li;
        const bool cok = col < g.N;
        const int colc = min(col, g.N - 1);
        float bvv = 0.f, cb0 = 0.f, cb1 = 0.f, bsh = 0.f;
        if constexpr (HAS_BIAS) bvv = g.bias[colc];
        if constexpr (HAS_BN) {
            // the shift of the shifted sums: any per-column value every tile agrees on and that sits near the column's mean --
            // the residual + per-cloud-bias part of ROW 0 of the result (data of this step only: a replayed graph and an eager
            // step compute identical bits; a running statistic would not give that); the heads' Linear + BatchNorm form
            // (EPI 9) shifts by the layer's bias: what is left in the sums is x W^T of post-ReLU rows, whose column mean can be a
            // few standard deviations -- the variance then carries (mean / std)^2 * 2^-24 of relative error (1.5e-6 at 5 sigma),
            // two orders below the tolerance the BatchNorm outputs are held to; a per-tile data shift would need a third row
            // in bn_part and a Chan merge in bn_finalize
            if constexpr (HAS_RES) bsh = g.resid[colc] + g.cbias[colc];
            else bsh = bvv;
            if (tm == 0 && wm0 == 0 && lh == 0 && cok) g.bn_shift[col] = bsh;
        }
        if constexpr (HAS_CB) {
            cb0 = g.cbias[(size_t)c0 * g.N + colc];
            cb1 = g.cbias[(size_t)min(c0 + 1, (g.M - 1) / g.rpc) * g.N + colc];
        }
#pragma unroll
        for (int x = 0; x < WM; ++x) {
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                float rv[4];
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = min(m0 + wm0 + 32 * x + q + 8 * (r4 >> 2) + 4 * lh, g.M - 1);
                        rv[q] = g.resid[(size_t)row * g.ldr + colc];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r4 + q;
                    const int row = m0 + wm0 + 32 * x + q + 8 * (r4 >> 2) + 4 * lh;
                    float v = g.alpha * acc[x][y][r] + bvv;
                    if constexpr (HAS_RES) v += rv[q];
                    if constexpr (HAS_CB) {
                        if (two_clouds) v += row >= nb ? cb1 : cb0;
                        else v += g.cbias[(size_t)(min(row, g.M - 1) / g.rpc) * g.N + colc];
                    }
                    if (row < g.M && cok) g.C[(size_t)row * g.ldc + col] = v;
                    if constexpr (HAS_BN) {
                        const float d = row < g.M ? v - bsh : 0.f;
                        bn1[y] += d;
                        bn2[y] += d * d;
                    }
                }
            }
        }
    }
    if constexpr (HAS_BN) {
        // the train-mode BatchNorm that follows the layer (FaceRecon.py:90-95) gets its first pass here: per-tile shifted column
        // sums of the rows just written, in the layout bn_finalize_kernel folds ([row tile][2][C]; fixed order: lane halves,
        // then the two row waves)
        __syncthreads();                                       // every wave is past its last fragment read: smem is free
        float* red = reinterpret_cast<float*>(smem);           // [row wave][2][128]
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float a = bn1[y] + __shfl_xor(bn1[y], 32, 64), b = bn2[y] + __shfl_xor(bn2[y], 32, 64);
            if (lh == 0) {
                red[((wave >> 1) * 2 + 0) * 128 + wn0 + 32 * y + li] = a;
                red[((wave >> 1) * 2 + 1) * 128 + wn0 + 32 * y + li] = b;
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < g.N) {
            g.bn_part[((size_t)tm * 2 + 0) * g.N + n0 + tid] = red[0 * 128 + tid] + red[2 * 128 + tid];
            g.bn_part[((size_t)tm * 2 + 1) * g.N + n0 + tid] = red[1 * 128 + tid] + red[3 * 128 + tid];
        }
    }
}

// ================================================================================================================================
// The PANEL form for the K = 128, wide-N products (fm = X W + b of conv_1 / conv_2, gcn3d.py:171: N = 1024 / 2048 columns).
// The tile kernel above spends a K = 128 tile as a four-block pipeline with an exposed memory round trip per block (matrix pipe 17 %
// busy); here the weight operand never moves again after the prologue:
//   * a 256-thread workgroup owns a 128-column PANEL for its whole life; each of its 4 waves holds the three bf16 planes of ITS 32
//     columns as MFMA B operands IN REGISTERS (8 ksteps x 3 planes x 4 VGPRs = 96) -- staged once through LDS with whole-line loads
//     (fragment-shaped loads straight from the (N, K) planes touch 64 lines for 1 KB: the prologue then cost 4-8 us per workgroup);
//   * the workgroup walks the 32-row tiles of the activations: a tile is fetched as fp32 two tiles ahead (registers), split into
//     its three planes on the way into a double-buffered LDS tile, ONE barrier per tile;
//   * the walk is a per-wave software pipeline: while the 48 MFMAs of tile i run (one accumulator chain: 8 x (3 ds_read_b128 +
//     6 MFMA 32x32x16)), the same wave stages tile i + 1 (ksteps 0-1), refills its prefetch registers (kstep 2) and stores tile
//     i - 1 from the other accumulator (ksteps 4-7) -- staging, multiplying and storing in separate phases left the matrix pipe
//     idle through two of the three (measured: every phase's time simply added up);
//   * two workgroups per CU, each with its own barrier;
//   * LDS rows are 256 bytes with the 16-byte chunks XOR-swizzled by (row & 15): the fragment reads (ds_read_b128 lane groups
//     = 16 distinct rows mod 16) and the staging writes (one row x 16 chunks) both touch all 64 banks (SQ_LDS_BANK_CONFLICT = 0);
//   * every memory operation of the walk is UNCONDITIONAL -- a tile past the end fetches rows past M, which the buffer descriptor
//     answers with zeros, and stores beyond the descriptor, which drops them: with branches around them hipcc can no longer count
//     the operations in flight and waits for ALL of them, prefetches included, inside every tile;
//   * workgroup -> (panel, row-tile slot) so that the workgroups of one XCD (blockIdx % 8) take ALL panels of their row slots: a
//     row tile is fetched from HBM once and found in that XCD's L2 by the other panels.
// ================================================================================================================================
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3_panel_kernel(const X3Args g) {
    constexpr int KS = 8, K = 128, D = 2;
    constexpr int ROWB = K * 2, PLANE = 32 * ROWB, BUF = 3 * PLANE;        // bytes: a row of one plane, a plane, a tile (24 KB)
    constexpr int UPR = K / 8, NAU = (32 * UPR) / 256;                     // 8-float units per row (16) / per thread (2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                                       // [2][3][32][K] bf16 (prologue: [3][64][K], a half panel)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int P = g.tiles_n, R = gridDim.x / P;
    const int xq = blockIdx.x / HSP_NUM_XCD, xcd = blockIdx.x % HSP_NUM_XCD;
    const int panel = xq % P, slot = (xq / P) * HSP_NUM_XCD + xcd;
    const int col = panel * 128 + 32 * wave + li;                          // (N is a multiple of 128: every column exists)
    const int ntiles = g.tiles_m;

    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A[0]), 0, (int)((((size_t)g.M - 1) * g.lda[0] + g.K[0]) * 4), 0x00020000);
    auto fetchA = [&](float (&dst)[NAU][8], int tile) {
#pragma unroll
        for (int i = 0; i < NAU; ++i) {
            const int u = tid + 256 * i, row = u / UPR, k0 = 8 * (u % UPR);
            const int gr = tile * 32 + row;
            const unsigned off = gr < g.M ? ((unsigned)gr * (unsigned)g.lda[0] + (unsigned)k0) * 4u : 0xfffffff0u;
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsA0, off, 0, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rsA0, off, 16, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dst[i][e] = __uint_as_float(v0[e]); dst[i][4 + e] = __uint_as_float(v1[e]); }
        }
    };
    auto stashA = [&](float (&srcv)[8], int unit, int buf) {
        const int u = tid + 256 * unit, row = u / UPR, c8 = u % UPR;
        u32x4 h, m, l;
        x3_split8(srcv, h, m, l);
        char* d = sA + buf * BUF + row * ROWB + ((c8 ^ (row & 15)) << 4);
        *reinterpret_cast<u32x4*>(d) = h;
        *reinterpret_cast<u32x4*>(d + PLANE) = m;
        *reinterpret_cast<u32x4*>(d + 2 * PLANE) = l;
    };

    // ---- activation tiles two ahead in registers (set j holds tile j mod 2); in flight under the weight staging
    float av[D][NAU][8];
    int t = slot;
#pragma unroll
    for (int j = 0; j < D; ++j) fetchA(av[j], t + j * R);

    // ---- this wave's weight fragments (column `col`, k = 16 s + 8 lh ... + 8, three planes) through LDS, half a panel (64 columns)
    // at a time: 16 consecutive threads read one 256-byte row of a plane, the waves that own those columns pick their fragments up
    u32x4 br[KS][3];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                                         // the first half's fragment reads are done
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int q = tid + 256 * i, p = q >> 10, n = (q >> 4) & 63, c = q & 15;
            const u32x4 v = *reinterpret_cast<const u32x4*>(g.P[0] + p * g.ps[0] + (size_t)(panel * 128 + 64 * half + n) * g.ldp[0] + 8 * c);
            *reinterpret_cast<u32x4*>(sA + (p * 64 + n) * ROWB + ((c ^ (n & 15)) << 4)) = v;
        }
        __syncthreads();
        if ((wave >> 1) == half) {
            const int n = 32 * (wave & 1) + li;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    br[s][p] = *reinterpret_cast<const u32x4*>(sA + (p * 64 + n) * ROWB + (((2 * s + lh) ^ (n & 15)) << 4));
        }
    }
    float bvv = 0.f;
    if constexpr (EPI & 1) bvv = g.bias[col];
    __syncthreads();                                                       // the staging area becomes the activation tiles
    // nothing of the prologue is pending inside the walk (a pending prologue load would put a conservative vmcnt wait in front of
    // every MFMA that reads a fragment register)
    __builtin_amdgcn_s_waitcnt(0x0070);                                    // vmcnt(0) lgkmcnt(0)
    stashA(av[0][0], 0, 0);
    stashA(av[0][1], 1, 0);
    fetchA(av[0], t + D * R);
    __syncthreads();

    // (stores: the row part of the address sits in the range-checked vector offset, so rows past M -- and a whole tile sent to offset
    // 2^31: the descriptor ends below that, nothing wraps -- are dropped; two VALU per store)
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)((((size_t)g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
    const unsigned ldc4 = (unsigned)g.ldc * 4u;
    auto store4 = [&](const f32x16& acc, int tile, bool valid, int r0) {                   // rows r0 .. r0 + 3 of the tile
        const unsigned base = valid ? (unsigned)(tile * 32 + 4 * lh) * ldc4 + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g.alpha * acc[r] + bvv), rsC, base + (unsigned)dr * ldc4, 0, 0);
        }
    };
    f32x16 acc2[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[a][r] = 0.f;
    int it = 0;
    auto step = [&](auto SJ, auto SP) {                                    // SJ: the register set that holds the NEXT tile; SP: accumulator
        constexpr int J = decltype(SJ)::value, PAR = decltype(SP)::value;
        const int buf = it & 1;
        const char* ab = sA + buf * BUF + li * ROWB;
        constexpr int SA[6] = {0, 2, 1, 0, 1, 0}, SB[6] = {2, 0, 1, 1, 0, 0};             // small terms first (as the tile kernel)
        u32x4 fa[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[0][p] = *reinterpret_cast<const u32x4*>(ab + p * PLANE + ((lh ^ (li & 15)) << 4));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                const int off = ((2 * (s + 1) + lh) ^ (li & 15)) << 4;
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[(s + 1) & 1][p] = *reinterpret_cast<const u32x4*>(ab + p * PLANE + off);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc2[PAR] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][SA[q]]),
                                                                    __builtin_bit_cast(bf16x8, br[s][SB[q]]),
                                                                    (s == 0 && q == 0) ? zero : acc2[PAR], 0, 0, 0);   // (C = 0: an inline constant)
            }
            // side work of this kstep, under its MFMAs
            if (s < NAU) stashA(av[J][s], s, buf ^ 1);                     // next tile -> the other LDS tile
            else if (s == NAU) fetchA(av[J], t + (D + 1) * R);             // refill two tiles ahead
            else if (s >= 4) store4(acc2[PAR ^ 1], t - R, it > 0, 4 * (s - 4));            // the previous tile
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        t += R;
        ++it;
    };
    while (t < ntiles) {                                                   // iteration `it` stages set (it + 1) mod 2 into tile (it + 1) & 1
        step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        if (t >= ntiles) break;
        step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {                                   // the last tile (tile t - R)
        if (it & 1) store4(acc2[0], t - R, it > 0, r0);
        else store4(acc2[1], t - R, it > 0, r0);
    }
}

// sum of the split-K partial tiles (fixed order), scaled; the products that split carry no other epilogue
__global__ __launch_bounds__(256) void gemm_x3_reduce_kernel(const float* __restrict__ ws, int nsplit, long long total, int N,
                                                             float alpha, float* __restrict__ C, int ldc) {
    if ((N & 3) == 0 && (ldc & 3) == 0) {
        const long long total4 = total >> 2;
        const int N4 = N >> 2;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
            float4 s = reinterpret_cast<const float4*>(ws)[e];
            for (int sp = 1; sp < nsplit; ++sp) {
                const float4 v = reinterpret_cast<const float4*>(ws)[(size_t)sp * total4 + e];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const long long row = e / N4;
            const int c4 = (int)(e - row * N4);
            *reinterpret_cast<float4*>(C + (size_t)row * ldc + 4 * c4) = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            float s = ws[e];
            for (int sp = 1; sp < nsplit; ++sp) s += ws[(size_t)sp * total + e];
            const long long row = e / N;
            C[(size_t)row * ldc + (int)(e - row * N)] = alpha * s;
        }
    }
}

// ---- fp32 parameters -> three bf16 planes in (N, K) form, every tensor of a step in ONE launch ------------------------------
// entry e: src (rows, cols) fp32 with row pitch ld.  transpose == 0: src is (N, K) -> plane[p][n][k] = slice_p(src[n][k]);
// transpose == 1: src is (K, N) -> plane[p][n][k] = slice_p(src[k][n]).  Plane row pitch kp (>= K rounded up to 32; the
// columns k >= K are zero: written once when the buffer is allocated), plane stride ps elements.  tile0 counts pieces of
// 32 (n) x 64 (k) OUTPUT elements: ceil(N / 32) * ceil(K / 64) per entry.
__global__ __launch_bounds__(256) void split_params_x3_kernel(const HspSplitDesc* __restrict__ tab, int n) {
    // a workgroup writes a 32 (n) x 64 (k) piece of the three planes; a thread owns k PAIRS (one dword store per plane)
    __shared__ float tile[64][33];
    int e = 0;
    while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].tile0) ++e;
    const HspSplitDesc d = tab[e];
    const int t = (int)blockIdx.x - d.tile0;
    const int Nn = d.transpose ? d.cols : d.rows, Kk = d.transpose ? d.rows : d.cols;
    const int tk = (Kk + 63) >> 6;
    const int n0 = (t / tk) * 32, k0 = (t % tk) * 64;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    unsigned* dst = reinterpret_cast<unsigned*>(d.dst);               // (kp and ps are even: dword-aligned pairs)
    float x0[4], x1[4];
    if (d.transpose) {                                                 // src rows = k, cols = n
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ly + 8 * j, nn = n0 + lx;
            tile[ly + 8 * j][lx] = (k < Kk && nn < Nn) ? d.src[(size_t)k * d.ld + nn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = tile[2 * lx][ly + 8 * j]; x1[j] = tile[2 * lx + 1][ly + 8 * j]; }
    } else {                                                           // src rows = n, cols = k
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + ly + 8 * j, k = k0 + 2 * lx;
            const float* p = d.src + (size_t)min(nn, Nn - 1) * d.ld;
            x0[j] = (nn < Nn && k < Kk) ? p[k] : 0.f;
            x1[j] = (nn < Nn && k + 1 < Kk) ? p[k + 1] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + ly + 8 * j, k = k0 + 2 * lx;
        if (nn >= Nn || k >= Kk) continue;
        float r1[2], r2[2];
        const float xs[2] = {x0[j], x1[j]};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float h = __uint_as_float(__float_as_uint(xs[q]) & 0xffff0000u);
            r1[q] = xs[q] - h;
            const float m = __uint_as_float(__float_as_uint(r1[q]) & 0xffff0000u);
            r2[q] = r1[q] - m;
        }
        const size_t o = ((size_t)nn * d.kp + k) >> 1;                 // dword index inside a plane
        const size_t psd = (size_t)d.ps >> 1;
        dst[o] = __builtin_amdgcn_perm(__float_as_uint(xs[1]), __float_as_uint(xs[0]), 0x07060302u);
        dst[psd + o] = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
        dst[2 * psd + o] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
    }
}

static int x3_pick_split(long long tiles, int TT) {
    // workgroups aimed at: 2 per CU (measured 3 and 4 per CU -- more, shorter K slices and a deeper fold: no faster)
    constexpr int per_cu = 2;
    if (tiles >= 2 * HSP_NUM_CU || TT < 16) return 1;
    int ns = (int)((per_cu * HSP_NUM_CU + tiles - 1) / tiles);
    if (ns > TT / 8) ns = TT / 8;
    if (ns > 16) ns = 16;
    if (ns < 2) return 1;
    const int per = (TT + ns - 1) / ns;
    return (TT + per - 1) / per;
}

// tile height: 128 rows when that still gives every CU a tile (or K is short and the tile count is what fills the chip)
static int x3_pick_wm(int M, int N) {
    const long long t128 = (long long)((M + 127) / 128) * ((N + X3_BN - 1) / X3_BN);
    return t128 >= 2 * HSP_NUM_CU ? 2 : 1;
}

}  // namespace hsp

using namespace hsp;

extern "C" int hsp_split_params_x3(const HspSplitDesc* table_dev, int n, int total_tiles, hspStream_t stream) {
    if (!table_dev || n <= 0 || total_tiles <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(split_params_x3_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), table_dev, n);
    return check_launch();
}

/* 1 when hsp_gemm_x3_f32 takes the shape (N >= 64; the activation rows must also be 16-byte aligned) */
extern "C" int hsp_gemm_x3_supported(int M, int N, int K1, int K2) {
    if (!(M > 0 && N >= 64 && K1 > 0 && K2 >= 0)) return 0;
    // fewer than 128 (64-row) tiles and no K deep enough to split: the wave / tile kernels keep more of the chip busy
    const long long t64 = (long long)((M + 63) / 64) * ((N + X3_BN - 1) / X3_BN);
    const int TT = (K1 + X3_BK - 1) / X3_BK + (K2 > 0 ? (K2 + X3_BK - 1) / X3_BK : 0);
    return (t64 >= 128 || TT >= 32) ? 1 : 0;
}

extern "C" size_t hsp_gemm_x3_workspace_bytes(int M, int N, int K1, int K2) {
    if (!hsp_gemm_x3_supported(M, N, K1, K2)) return 0;
    const int wm = x3_pick_wm(M, N), bm = 64 * wm;
    const int TT = (K1 + X3_BK - 1) / X3_BK + (K2 > 0 ? (K2 + X3_BK - 1) / X3_BK : 0);
    const int ns = x3_pick_split((long long)((M + bm - 1) / bm) * ((N + X3_BN - 1) / X3_BN), TT);
    return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

static int gemm_x3_impl(const float* A1, int lda1, const hsp_bf16_t* P1, int ldp1, long long ps1, int K1,
                        const float* A2, int lda2, const hsp_bf16_t* P2, int ldp2, long long ps2, int K2, int M, int N,
                        const float* bias, const float* resid, int ldr, const float* cloud_bias, int rows_per_cloud,
                        float alpha, float* C, int ldc, void* ws, size_t ws_bytes, float* bn_shift, float* bn_part,
                        hspStream_t stream) {
    if (!A1 || !P1 || !C || M <= 0 || N <= 0 || K1 <= 0 || lda1 < K1 || ldc < N) return HSP_ERR_BAD_ARG;
    const bool two = A2 != nullptr;
    if (two && (!P2 || K2 <= 0 || lda2 < K2)) return HSP_ERR_BAD_ARG;
    if (!two) K2 = 0;
    if (resid && ldr < N) return HSP_ERR_BAD_ARG;
    if (cloud_bias && rows_per_cloud <= 0) return HSP_ERR_BAD_ARG;
    if (!hsp_gemm_x3_supported(M, N, K1, K2)) return HSP_ERR_UNSUPPORTED;
    if ((long long)M * lda1 * 4 >= (1ll << 31) || (two && (long long)M * lda2 * 4 >= (1ll << 31))) return HSP_ERR_UNSUPPORTED;
    auto al16 = [](const void* q, long long ld, int es) { return ((reinterpret_cast<size_t>(q) | ((size_t)ld * es)) & 15) == 0; };
    const int kp1 = (K1 + X3_BK - 1) / X3_BK * X3_BK, kp2 = (K2 + X3_BK - 1) / X3_BK * X3_BK;
    if (!al16(A1, lda1, 4) || !al16(P1, ldp1, 2) || (ps1 & 7) || ldp1 < kp1) return HSP_ERR_UNSUPPORTED;
    if (two && (!al16(A2, lda2, 4) || !al16(P2, ldp2, 2) || (ps2 & 7) || ldp2 < kp2)) return HSP_ERR_UNSUPPORTED;
    X3Args g{};
    g.A[0] = A1; g.P[0] = P1; g.lda[0] = lda1; g.ldp[0] = ldp1; g.K[0] = K1; g.ps[0] = ps1;
    g.A[1] = A2; g.P[1] = P2; g.lda[1] = lda2; g.ldp[1] = ldp2; g.K[1] = K2; g.ps[1] = ps2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.cbias = cloud_bias; g.rpc = rows_per_cloud > 0 ? rows_per_cloud : 1; g.alpha = alpha;
    g.bn_shift = bn_shift; g.bn_part = bn_part;
    const bool bn = bn_shift != nullptr;
    // BatchNorm partials: the layer's out product (residual + per-cloud bias; 64-row tiles) or a Linear with bias (one source)
    const bool bn_out = bn && resid && cloud_bias && !bias, bn_lin = bn && bias && !resid && !cloud_bias && !two;
    if (bn && (!bn_part || !(bn_out || bn_lin))) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    {   // the panel form: K = 128, columns in whole 128-wide panels, enough row tiles per panel slot to amortise the weight fragments
        const int epi0 = (bias ? 1 : 0) | (resid ? 2 : 0) | (cloud_bias ? 4 : 0);
        if (!bn && !two && epi0 <= 1 && K1 == 128 && N % 128 == 0 && (long long)(M + 32) * ldc * 4 < (1ll << 31)) {
            const int P = N / 128;
            const int R = (2 * HSP_NUM_CU / P) / HSP_NUM_XCD * HSP_NUM_XCD;  // two workgroups per CU; a multiple of the XCD count
            const int nt = (M + 31) / 32;
            const int rounds = R ? (nt + R - 1) / R : 0;
            if (R && rounds >= 3 && (double)nt / ((double)rounds * R) >= 0.75) {
                g.tiles_m = nt; g.tiles_n = P; g.nsplit = 1; g.ws = nullptr;
                const dim3 pgrid((unsigned)(P * R)), pblock(256);
                const size_t lds = (size_t)2 * 3 * 32 * 128 * 2;             // 48 KB
#define X3_PANEL(EPI_)                                                                                                    \
    do {                                                                                                                  \
        auto kern = gemm_x3_panel_kernel<EPI_>;                                                                           \
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                        \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, pgrid, pblock, lds, st, g);                                                              \
    } while (0)
                if (epi0) X3_PANEL(1);
                else X3_PANEL(0);
#undef X3_PANEL
                return check_launch();
            }
        }
    }
    const int wm = bn_out ? 1 : x3_pick_wm(M, N), bm = 64 * wm;
    g.tiles_m = (M + bm - 1) / bm; g.tiles_n = (N + X3_BN - 1) / X3_BN;
    const int TT = (K1 + X3_BK - 1) / X3_BK + (two ? (K2 + X3_BK - 1) / X3_BK : 0);
    const int epi = (bias ? 1 : 0) | (resid ? 2 : 0) | (cloud_bias ? 4 : 0);
    int ns = epi ? 1 : x3_pick_split((long long)g.tiles_m * g.tiles_n, TT);
    if (ns > 1 && (!ws || (size_t)ns * M * N * sizeof(float) > ws_bytes)) ns = 1;
    g.nsplit = ns; g.ws = ns > 1 ? reinterpret_cast<float*>(ws) : nullptr;
    const long long items = (long long)g.tiles_m * g.tiles_n * ns;
    if (items > (1ll << 30)) return HSP_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)items), block(256);
    // instantiated epilogues: 0 none, 1 bias, 2 residual (one source: the input-gradient chain of layers that share their input
    // rows -- C may BE resid: an element is read and written by the same thread), 6 residual + per-cloud bias (the layer's out
    // product), 14 = 6 + BatchNorm partials
    // 4: per-cloud bias alone (one source: a Linear whose input is cat[per-cloud vector, rows] -- the vector's columns become a bias)
    if (epi != 0 && epi != 1 && epi != 6 && !(epi == 2 && !two && alpha == 1.0f) && !(epi == 4 && !two)) return HSP_ERR_UNSUPPORTED;
    if (bn_out) {
        if (two) hipLaunchKernelGGL((gemm_x3_kernel<1, true, 14, 0>), grid, block, 3 * (size_t)(64 + X3_BN) * 64, st, g);
        else hipLaunchKernelGGL((gemm_x3_kernel<1, false, 14, 0>), grid, block, 3 * (size_t)(64 + X3_BN) * 64, st, g);
        return check_launch();
    }
    if (bn_lin) {
        if (g.tiles_m > 512) return HSP_ERR_UNSUPPORTED;                       // (BN_MAX_PARTIALS of norm.hip)
        if (wm == 2) hipLaunchKernelGGL((gemm_x3_kernel<2, false, 9, 0>), grid, block, 3 * (size_t)(128 + X3_BN) * 64, st, g);
        else hipLaunchKernelGGL((gemm_x3_kernel<1, false, 9, 0>), grid, block, 3 * (size_t)(64 + X3_BN) * 64, st, g);
        return check_launch();
    }
#define X3_K(WM_, TWO_, EPI_, PA_)                                                                                 \
    do {                                                                                                           \
        auto kern = gemm_x3_kernel<WM_, TWO_, EPI_, PA_>;                                                          \
        const size_t lds = 3 * (size_t)(64 * WM_ + X3_BN) * 64;                                                    \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g);                                                         \
    } while (0)
#define X3_LAUNCH(WM_, PA_)                                        \
    do {                                                           \
        if (two) {                                                 \
            if (epi == 0) X3_K(WM_, true, 0, PA_);                 \
            else if (epi == 1) X3_K(WM_, true, 1, PA_);            \
            else X3_K(WM_, true, 6, PA_);                          \
        } else {                                                   \
            if (epi == 0) X3_K(WM_, false, 0, PA_);                \
            else if (epi == 1) X3_K(WM_, false, 1, PA_);           \
            else if (epi == 2) X3_K(WM_, false, 2, 0);             \
            else if (epi == 4) X3_K(WM_, false, 4, 0);             \
            else X3_K(WM_, false, 6, PA_);                         \
        }                                                          \
    } while (0)
    if (wm == 2) X3_LAUNCH(2, 0);
    else X3_LAUNCH(1, 0);
#undef X3_LAUNCH
#undef X3_K
    int rc = check_launch();
    if (rc || ns == 1) return rc;
    const long long total = (long long)M * N;
    long long rg = (total / 4 + 255) / 256;
    if (rg > HSP_NUM_CU * 8) rg = HSP_NUM_CU * 8;
    hipLaunchKernelGGL(gemm_x3_reduce_kernel, dim3((unsigned)rg), dim3(256), 0, st, g.ws, ns, total, N, alpha, C, ldc);
    return check_launch();
}

/* a Linear / Conv1d(k=1) with bias whose result feeds a train-mode BatchNorm (the heads: PoseR.py:27-30, FaceRecon.py:37-47):
 * C = A W^T + bias AND, per row tile, the shifted column sums of C (shift = bias, written to bn_shift): bn_part[tiles][2][N] with
 * tiles = hsp_gemm_x3_bn_tiles(M, N); fold with hsp_bn_relu_fwd_partials(..., bn_part, tiles, bn_shift) */
extern "C" int hsp_gemm_x3_bn_tiles(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (M + 64 * x3_pick_wm(M, N) - 1) / (64 * x3_pick_wm(M, N));
}
extern "C" int hsp_gemm_x3_bias_bn_f32(const float* A1, int lda1, const hsp_bf16_t* P1, int ldp1, long long ps1, int K1, int M, int N,
                                       const float* bias, float* C, int ldc, float* bn_shift, float* bn_part, hspStream_t stream) {
    if (!bn_shift || !bn_part || !bias) return HSP_ERR_BAD_ARG;
    return gemm_x3_impl(A1, lda1, P1, ldp1, ps1, K1, nullptr, 0, nullptr, 0, 0, 0, M, N, bias, nullptr, 0, nullptr, 0, 1.0f, C, ldc,
                        nullptr, 0, bn_shift, bn_part, stream);
}

// ================================================================================================================================
// the per-CLOUD products of the ORL branch: 16 rows (one per cloud of the batch) against a (C, C) weight block
//   t   = fg Wb^T            (gcn3d.py:186, the f_global half of conv2)          "nt": out[m][n] = sum_k A[m][k] W[n][k]
//   gfg = (gt Wb) / N        (its input gradient)                                "nn": out[m][n] = alpha sum_k A[m][k] W[k][n]
//   gWb = gt^T fg            (its weight gradient)                               outer: out[i][j] = sum_b a[b][i] c[b][j]
// Three tiny kernels, one launch each (the tile kernels need split-K + a fold for a 16-row product: two launches of ~7 us).
// fp32 fma chains in a fixed order (bit-reproducible).
// ================================================================================================================================
namespace hsp {

#define SR_MAXM 16

// "nt": lanes along k (coalesced rows of W), a wave per output column pair, wave-level tree reduction (fixed order)
__global__ __launch_bounds__(256) void small_rows_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            int M, int N, int K, float alpha, float* __restrict__ out, int ldo) {
    extern __shared__ float sA[];                              // M x K
    for (int e = threadIdx.x; e < M * K; e += 256) sA[e] = A[(size_t)(e / K) * lda + (e % K)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = blockIdx.x * 8 + wave; n < min(N, (int)blockIdx.x * 8 + 8); n += 4) {
        float p[SR_MAXM];
#pragma unroll
        for (int m = 0; m < SR_MAXM; ++m) p[m] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float w = W[(size_t)n * ldw + k];
#pragma unroll
            for (int m = 0; m < SR_MAXM; ++m)
                if (m < M) p[m] = __fmaf_rn(sA[m * K + k], w, p[m]);
        }
#pragma unroll
        for (int m = 0; m < SR_MAXM; ++m) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) p[m] += __shfl_xor(p[m], o, 64);
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < SR_MAXM; ++m)
                if (m < M) out[(size_t)m * ldo + n] = alpha * p[m];
        }
    }
}

// "nn": a thread per output column (coalesced rows of W), four k slices per workgroup folded through LDS in slice order
__global__ __launch_bounds__(256) void small_rows_nn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            int M, int N, int K, float alpha, float* __restrict__ out, int ldo) {
    extern __shared__ float smem_f[];
    float* sA = smem_f;                                        // M x K
    float* red = smem_f + M * K;                               // 3 x 64 x SR_MAXM
    for (int e = threadIdx.x; e < M * K; e += 256) sA[e] = A[(size_t)(e / K) * lda + (e % K)];
    __syncthreads();
    const int cl = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    const int kper = (K + 3) / 4, k0 = ks * kper, k1 = min(K, k0 + kper);
    float p[SR_MAXM];
#pragma unroll
    for (int m = 0; m < SR_MAXM; ++m) p[m] = 0.f;
    if (n < N) {
        for (int k = k0; k < k1; ++k) {
            const float w = W[(size_t)k * ldw + n];
#pragma unroll
            for (int m = 0; m < SR_MAXM; ++m)
                if (m < M) p[m] = __fmaf_rn(sA[m * K + k], w, p[m]);
        }
    }
    if (ks > 0) {
#pragma unroll
        for (int m = 0; m < SR_MAXM; ++m) red[((ks - 1) * 64 + cl) * SR_MAXM + m] = p[m];
    }
    __syncthreads();
    if (ks == 0 && n < N) {
#pragma unroll
        for (int m = 0; m < SR_MAXM; ++m) {
            if (m < M) {
                float s = p[m];
                for (int q = 0; q < 3; ++q) s += red[(q * 64 + cl) * SR_MAXM + m];
                out[(size_t)m * ldo + n] = alpha * s;
            }
        }
    }
}

// the fast form of both layouts (K a multiple of 128, 16-byte aligned rows): v_mfma_f32_16x16x4_f32, a workgroup per 16 output
// columns, its four waves take a quarter of K each (every operand load of a wave's quarter is issued before the first MFMA: one
// memory round trip) and fold through LDS in wave order.  "nt" lanes fetch 16 bytes = 4 consecutive k of their row: MFMA e of a
// group of four multiplies the k set {e, 4+e, 8+e, 12+e} -- the same permutation on both operands, the same products.
using f32x4v = __attribute__((ext_vector_type(4))) float;
template <bool NN>
__global__ __launch_bounds__(256) void small_rows_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                              int M, int N, int K, float alpha, float* __restrict__ out, int ldo) {
    __shared__ float red[3][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int mbase = blockIdx.y * 16;                         // 16-row blocks along grid.y (M up to 64: one row per cloud)
    const int n = blockIdx.x * 16 + j, nc = min(n, N - 1), mc = min(mbase + j, M - 1);
    const float amask = mbase + j < M ? 1.f : 0.f;
    const int kper = K >> 2, k0 = wave * kper;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NN) {
        // kper <= 128 (K <= 512): every load of the quarter in flight at once; deeper K in rounds of 128
        for (int kb = 0; kb < kper; kb += 128) {
            float a[32], b[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int k = k0 + min(kb + 4 * u, kper - 4) + kk;
                b[u] = W[(size_t)k * ldw + nc];
                a[u] = A[(size_t)mc * lda + k];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (kb + 4 * u < kper) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u] * amask, b[u], acc, 0, 0, 0);
        }
    } else {
        for (int kb = 0; kb < kper; kb += 128) {
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + min(kb + 16 * u, kper - 16) + 4 * kk;
                b[u] = *reinterpret_cast<const float4*>(W + (size_t)nc * ldw + k);
                a[u] = *reinterpret_cast<const float4*>(A + (size_t)mc * lda + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kb + 16 * u < kper) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x * amask, b[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y * amask, b[u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z * amask, b[u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w * amask, b[u].w, acc, 0, 0, 0);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && n < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mbase + 4 * kk + r;
            if (row < M) out[(size_t)row * ldo + n] = alpha * (((acc[r] + red[0][lane][r]) + red[1][lane][r]) + red[2][lane][r]);
        }
    }
}

// out (Ma, Nb) = a^T c over the B (<= 16) per-cloud rows: a (B, Ma), c (B, Nb)
__global__ __launch_bounds__(256) void small_outer_kernel(const float* __restrict__ a, int lda, const float* __restrict__ c, int ldc_,
                                                          int B, int Ma, int Nb, float* __restrict__ out, int ldo,
                                                          const float* __restrict__ mom, int ldm, int Cm, float* __restrict__ gste) {
    const long long total = (long long)Ma * Nb;
    // rider job (HSlayer_surface): gste[c][j] = sum_b mom[b][j * Cm + c], j = 0..2 -- the per-cloud coordinate moments of g
    // (hsp_colsum_rows_xyz) summed over the batch, in cloud order
    if (mom) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < 3 * Cm; e += gridDim.x * 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += mom[(size_t)b * ldm + e];
            const int j = e / Cm, cc = e - j * Cm;
            gste[cc * 3 + j] = s;
        }
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / Nb), j = (int)(e - (long long)i * Nb);
        float s = 0.f;
        for (int b0 = 0; b0 < B; b0 += 16) {                   // 32 loads in flight, then the chain in row order
            float av[16], cv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = min(b0 + u, B - 1);
                av[u] = a[(size_t)b * lda + i]; cv[u] = c[(size_t)b * ldc_ + j];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b0 + u < B) s = __fmaf_rn(av[u], cv[u], s);
        }
        out[(size_t)i * ldo + j] = s;
    }
}

}  // namespace hsp

/* out (M, N) = alpha * A (M, K) op(W): w_layout 0 = W is (N, K) ("nt"), 1 = W is (K, N) ("nn"); M <= 16 rows (one per cloud),
 * K <= 2048.  gcn3d.py:186 (the f_global half of conv2) and its input gradient. */
extern "C" int hsp_small_rows_f32(const float* A, int lda, const float* W, int ldw, int w_layout, int M, int N, int K, float alpha,
                                  float* out, int ldo, hspStream_t stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || lda < K || ldo < N) return HSP_ERR_BAD_ARG;
    if (K > 2048 || M > 64) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (w_layout != 0 && w_layout != 1) return HSP_ERR_BAD_ARG;
    const bool al = ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(W) | ((size_t)lda * 4) | ((size_t)ldw * 4)) & 15) == 0;
    if (K % 128 == 0 && (w_layout == 1 || al)) {
        const dim3 grid((N + 15) / 16, (M + 15) / 16);
        if (w_layout == 1)
            hipLaunchKernelGGL(small_rows_mfma_kernel<true>, grid, dim3(256), 0, st, A, lda, W, ldw, M, N, K, alpha, out, ldo);
        else
            hipLaunchKernelGGL(small_rows_mfma_kernel<false>, grid, dim3(256), 0, st, A, lda, W, ldw, M, N, K, alpha, out, ldo);
        return check_launch();
    }
    if (M > SR_MAXM) return HSP_ERR_UNSUPPORTED;
    if (w_layout == 0)
        hipLaunchKernelGGL(small_rows_nt_kernel, dim3((N + 7) / 8), dim3(256), (size_t)M * K * 4, st, A, lda, W, ldw, M, N, K, alpha, out, ldo);
    else if (w_layout == 1) {
        const size_t lds = ((size_t)M * K + 3 * 64 * SR_MAXM) * 4;
        if (lds > 64 * 1024) return HSP_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(small_rows_nn_kernel, dim3((N + 63) / 64), dim3(256), lds, st, A, lda, W, ldw, M, N, K, alpha, out, ldo);
    } else return HSP_ERR_BAD_ARG;
    return check_launch();
}

/* out (Ma, Nb) = a^T c for per-cloud rows a (B, Ma), c (B, Nb), B <= 16: the weight gradient of the f_global half of conv2 */
extern "C" int hsp_small_outer_f32(const float* a, int lda, const float* c, int ldc, int B, int Ma, int Nb, float* out, int ldo,
                                   const float* mom, int ldm, int Cm, float* gste, hspStream_t stream) {
    if (!a || !c || !out || B <= 0 || Ma <= 0 || Nb <= 0 || lda < Ma || ldc < Nb || ldo < Nb) return HSP_ERR_BAD_ARG;
    if (mom && (!gste || Cm <= 0 || ldm < 3 * Cm)) return HSP_ERR_BAD_ARG;
    if (B > 64) return HSP_ERR_UNSUPPORTED;
    const long long total = (long long)Ma * Nb;
    long long g = (total + 255) / 256;
    if (g > HSP_NUM_CU * 8) g = HSP_NUM_CU * 8;
    hipLaunchKernelGGL(small_outer_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), a, lda, c, ldc, B, Ma, Nb, out, ldo,
                       mom, ldm, Cm, gste);
    return check_launch();
}

extern "C" int hsp_gemm_x3_f32(const float* A1, int lda1, const hsp_bf16_t* P1, int ldp1, long long ps1, int K1,
                               const float* A2, int lda2, const hsp_bf16_t* P2, int ldp2, long long ps2, int K2, int M, int N,
                               const float* bias, const float* resid, int ldr, const float* cloud_bias, int rows_per_cloud,
                               float alpha, float* C, int ldc, void* ws, size_t ws_bytes, hspStream_t stream) {
    return gemm_x3_impl(A1, lda1, P1, ldp1, ps1, K1, A2, lda2, P2, ldp2, ps2, K2, M, N, bias, resid, ldr, cloud_bias, rows_per_cloud,
                        alpha, C, ldc, ws, ws_bytes, nullptr, nullptr, stream);
}

/* the layer's out product (residual + per-cloud bias) that ALSO leaves the first pass of the train-mode BatchNorm that follows
 * it: bn_part[(M + 63) / 64][2][N] = per 64-row tile, sum (c - s) and sum (c - s)^2 over the tile's rows of the result c, with
 * the shift s[n] = resid[0][n] + cloud_bias[0][n] written to bn_shift (N floats); fold with hsp_bn_relu_fwd_partials
 * (nblk = (M + 63) / 64, shift = bn_shift) */
extern "C" int hsp_gemm_x3_bn_f32(const float* A1, int lda1, const hsp_bf16_t* P1, int ldp1, long long ps1, int K1,
                                  const float* A2, int lda2, const hsp_bf16_t* P2, int ldp2, long long ps2, int K2, int M, int N,
                                  const float* resid, int ldr, const float* cloud_bias, int rows_per_cloud, float* C, int ldc,
                                  float* bn_shift, float* bn_part, hspStream_t stream) {
    if (!bn_shift || !bn_part || (M + 63) / 64 > 512) return HSP_ERR_BAD_ARG;
    return gemm_x3_impl(A1, lda1, P1, ldp1, ps1, K1, A2, lda2, P2, ldp2, ps2, K2, M, N, nullptr, resid, ldr, cloud_bias,
                        rows_per_cloud, 1.0f, C, ldc, nullptr, 0, bn_shift, bn_part, stream);
}
