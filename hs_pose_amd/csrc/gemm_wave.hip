// gemm_wave.hip -- the dense per-point products of the HS stack on the fp32 matrix cores of gfx950 WITHOUT LDS: every wave
// is an independent worker that streams both operands from L2 / L1 straight into MFMA operand layout with 16-byte loads.
//
//   C[r][n] = alpha * ( sum_k A1[r][k] * op(B1)[k][n]  (+ sum_k A2[r][k] * op(B2)[k][n]) )
//             (+ bias[n]) (+ resid[r][n]) (+ cloud_bias[r / rows_per_cloud][n]) (+ xyz3[r] . w3[n])
//
// the same contract as gemm_rows.hip (reference network/fs_net_repo/gcn3d.py:149,171,186 and their input gradients) for the
// shapes of the layer path: K1, K2 multiples of 32, N a multiple of the wave tile's width, 16-byte aligned rows.
//
// Why no LDS.  v_mfma_f32_32x32x2_f32 retires 4096 flop in 64 cycles, so a wave tile of 32x128 ... 64x128 needs 12-20 bytes of
// operands per clock and CU -- about what a CU can pull from L2 (measured here: ~15 B/clk/CU with every CU streaming) -- and
// nothing else: no barrier per k-block, no staging writes, no workgroup-wide tile whose size quantises 257 * 2^j rows badly
// on 256 CUs.  The products this pays for are the ones with MANY output tiles and a SHORT K (fm = X W + b: K = 128 / 256,
// N = 1024 ... 4096), where an LDS-staged tile lives for four k-blocks and spends as long in its prologue and epilogue.
//   * a wave owns a (32 RB) x (32 NCB) tile of C; lane (li, lh) of the wave holds, per step of 8 k,
//       A:  RB  x 16 bytes  A[row0 + 32 rb + li][k8 + 4 lh .. +3]                    (one row per lane: the MFMA A layout)
//       B:  "nn" (K,N): 4 x (4 NCB) bytes  B[k8 + 4 lh + j][col0 + NCB li .. + NCB-1]   j = 0..3
//           "nt" (N,K): NCB x 16 bytes     B[col0 + NCB li + cb][k8 + 4 lh .. +3]
//     The two half-waves then trade registers (v_permlane32_swap: (t0,t1) -> (k8+0|k8+1), (k8+4|k8+5); (t2,t3) -> (k8+2|k8+3),
//     (k8+6|k8+7)) so that the four MFMAs of a step multiply the k pairs (0,1), (2,3), (4,5), (6,7) IN ASCENDING ORDER: a
//     product is then bit for bit the k-ordered fp32 fma chain from 0 that the reference's CPU GEMM runs for K <= 256
//     (oracle/gen_golden_exact.py pins that) -- 10 VALU swaps per 16 MFMAs.  The column a lane stands for in column block cb is
//     col0 + NCB li + cb -- the NCB accumulators of a row are NCB CONSECUTIVE columns, so "nn" operands load with one wide
//     access per k and the tile is written with 16-byte stores (NCB = 4);
//   * the loads run a ring of 4 steps ahead of the MFMAs (buffer loads: scalar step offsets, rows past M read 0) and the ring
//     does not drain between the two sources of a product nor between the tiles of a wave: the last three load groups of a
//     segment already fetch the first three steps of the next one, so a tile's epilogue and the next tile's first-load latency
//     overlap (measured before that: two waves of a SIMD in lockstep both sat in their prologue at the same time);
//   * work is cut per WAVE: tiles in panel order, wave w takes a contiguous run; the tiles that do not divide by the wave count
//     are cut into their 32 x 32 blocks (one accumulator, full K: no partial sums, no fold) and dealt one per wave.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

#ifndef GW_ABLATE
#define GW_ABLATE 0                        // profiling builds: 1 = no MFMA, 2 = every step re-reads step 0 (L1 hits)
#endif

struct GwArgs {
    const float* A[2]; const float* B[2];
    int lda[2], ldb[2], K[2], lb[2];       // lb: 0 = (N,K) rows ("nt"), 1 = (K,N) rows ("nn")
    float* C; int ldc; int M, N;
    const float* bias;
    const float* resid; int ldr;
    const float* resid2; int ldr2;         // exact forms: a second residual, added LAST
    const float* cbias; int rpc;
    const float* xyz3; const float* w3;
    float alpha;
    int TM, TN;                            // tiles along M / N
    int T0, T1;                            // steps (of 8 k) of source 0 / 1
    int base;                              // whole tiles per wave
    int u_rem, npieces;                    // first leftover tile; 32 x 32 blocks of the leftover tiles (one per wave)
    int nwaves;
    int order;                             // 1: row panels fastest in the tile order
};

__device__ __forceinline__ float u2f(unsigned v) { return __uint_as_float(v); }

// one source of one tile: where its operands are
template <int RB>
struct GwSeg {
    __amdgpu_buffer_rsrc_t ra, rb;
    unsigned voffA[RB], voffB, ldb_bytes;
    int nsteps, lb;
};

// operand loads of step s of segment g into ring slot Q
template <int RB, int NCB, int LB, int Q>
__device__ __forceinline__ void gw_issue(u32x4 (&a)[4][RB], u32x4 (&b)[4][4], const GwSeg<RB>& g, int s) {
#if GW_ABLATE == 2
    s = 0;
#endif
#pragma unroll
    for (int x = 0; x < RB; ++x) a[Q][x] = __builtin_amdgcn_raw_buffer_load_b128(g.ra, g.voffA[x], (unsigned)s * 32u, 0);
    if constexpr (LB == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned so = ((unsigned)s * 8u + j) * g.ldb_bytes;
            if constexpr (NCB == 4) b[Q][j] = __builtin_amdgcn_raw_buffer_load_b128(g.rb, g.voffB, so, 0);
            else if constexpr (NCB == 2) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(g.rb, g.voffB, so, 0);
                b[Q][j][0] = t[0]; b[Q][j][1] = t[1];
            } else b[Q][j][0] = __builtin_amdgcn_raw_buffer_load_b32(g.rb, g.voffB, so, 0);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCB; ++c)
            b[Q][c] = __builtin_amdgcn_raw_buffer_load_b128(g.rb, g.voffB, (unsigned)c * g.ldb_bytes + (unsigned)s * 32u, 0);
    }
}

// lanes (li, 0) / (li, 1) hold k8 + 0..3 / k8 + 4..7 in (t0,t1,t2,t3): after the two swaps t0 = (k8+0 | k8+1), t2 = (k8+2 | k8+3),
// t1 = (k8+4 | k8+5), t3 = (k8+6 | k8+7) in (lower | upper) half-wave: ascending k pairs in the order 0, 2, 1, 3
#define GW_PAIR_UP(T0, T1, T2, T3)                                                       \
    {                                                                                    \
        const u32x2 p_ = __builtin_amdgcn_permlane32_swap((T0), (T1), false, false);     \
        const u32x2 q_ = __builtin_amdgcn_permlane32_swap((T2), (T3), false, false);     \
        (T0) = p_[0]; (T1) = p_[1]; (T2) = q_[0]; (T3) = q_[1];                           \
    }

template <int RB, int NCB, int LB, int Q>
__device__ __forceinline__ void gw_compute(f32x16 (&acc)[RB][NCB], u32x4 (&a)[4][RB], u32x4 (&b)[4][4]) {
#pragma unroll
    for (int x = 0; x < RB; ++x) GW_PAIR_UP(a[Q][x][0], a[Q][x][1], a[Q][x][2], a[Q][x][3])
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        if constexpr (LB == 1) GW_PAIR_UP(b[Q][0][c], b[Q][1][c], b[Q][2][c], b[Q][3][c])
        else GW_PAIR_UP(b[Q][c][0], b[Q][c][1], b[Q][c][2], b[Q][c][3])
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int j = (jj == 1) ? 2 : (jj == 2) ? 1 : jj;      // registers in ascending-k order: 0, 2, 1, 3
                const float bv = LB == 1 ? u2f(b[Q][j][c]) : u2f(b[Q][c][j]);
                const float av = u2f(a[Q][x][j]);
#if GW_ABLATE == 1
                asm volatile("" ::"v"(bv), "v"(av));
#else
                acc[x][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x][c], 0, 0, 0);
#endif
            }
}

// all steps of segment `cur` (its steps 0..2 are already in ring slots 0..2); the last three load groups fetch steps 0..2 of `nxt`
template <int RB, int NCB, int LBC, int LBN>
__device__ __forceinline__ void gw_segment(f32x16 (&acc)[RB][NCB], u32x4 (&a)[4][RB], u32x4 (&b)[4][4], const GwSeg<RB>& cur,
                                           const GwSeg<RB>& nxt) {
    int s = 0;
    for (; s < cur.nsteps - 4; s += 4) {
        gw_issue<RB, NCB, LBC, 3>(a, b, cur, s + 3);
        __builtin_amdgcn_sched_barrier(0);
        gw_compute<RB, NCB, LBC, 0>(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
        gw_issue<RB, NCB, LBC, 0>(a, b, cur, s + 4);
        __builtin_amdgcn_sched_barrier(0);
        gw_compute<RB, NCB, LBC, 1>(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
        gw_issue<RB, NCB, LBC, 1>(a, b, cur, s + 5);
        __builtin_amdgcn_sched_barrier(0);
        gw_compute<RB, NCB, LBC, 2>(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
        gw_issue<RB, NCB, LBC, 2>(a, b, cur, s + 6);
        __builtin_amdgcn_sched_barrier(0);
        gw_compute<RB, NCB, LBC, 3>(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
    }
    gw_issue<RB, NCB, LBC, 3>(a, b, cur, s + 3);
    __builtin_amdgcn_sched_barrier(0);
    gw_compute<RB, NCB, LBC, 0>(acc, a, b);
    __builtin_amdgcn_sched_barrier(0);
    gw_issue<RB, NCB, LBN, 0>(a, b, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    gw_compute<RB, NCB, LBC, 1>(acc, a, b);
    __builtin_amdgcn_sched_barrier(0);
    gw_issue<RB, NCB, LBN, 1>(a, b, nxt, 1);
    __builtin_amdgcn_sched_barrier(0);
    gw_compute<RB, NCB, LBC, 2>(acc, a, b);
    __builtin_amdgcn_sched_barrier(0);
    gw_issue<RB, NCB, LBN, 2>(a, b, nxt, 2);
    __builtin_amdgcn_sched_barrier(0);
    gw_compute<RB, NCB, LBC, 3>(acc, a, b);
    __builtin_amdgcn_sched_barrier(0);
}

struct GwRsrc { __amdgpu_buffer_rsrc_t A0, B0, A1, B1, C, R, R2; };

// `count` tiles of (32 RB) x (32 NCB): tile i has its first row at row0(i) and lane li its first column at col0(i) + cs li
// (bulk tiles: cs = NCB; a 32 x 32 block cut out of a wider tile keeps that tile's column stride)
// EPI: what the epilogue adds -- bit 0 bias, bit 1 residual + per-cloud bias, bit 2 the K = 3 coordinate product, bit 3 relu;
// the EXACT forms (the reference's own order of additions, gcn3d.py:112,156,186): bit 5 residual alone, bit 6 per-cloud bias THEN
// residual, bit 4 a second residual added last.  A2C: the second source's rows are per-CLOUD rows (row r reads A2[r / rpc]):
// conv2 over cat[F, f_global] (gcn3d.py:111,185) as ONE chain that runs on from F's channels into f_global's.  (Template
// parameters, not run-time tests: a load inside a run-time branch makes hipcc wait vmcnt(0) at the join -- measured: a drained
// prefetch ring and a full memory round trip per tile.)
template <int RB, int NCB, int LB0, int LB1, bool TWO, int EPI, bool A2C, typename TileFn>
__device__ __forceinline__ void gw_tiles(const GwArgs& g, const GwRsrc& rs, int count, int cs, TileFn tile_of) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    constexpr bool two = TWO;
    constexpr bool HAS_BIAS = EPI & 1, HAS_XYZ = EPI & 4, HAS_RELU = EPI & 8;
    constexpr bool EXACT = (EPI & (16 | 32 | 64)) != 0;               // the reference's order of additions
    constexpr bool HAS_RES = (EPI & (2 | 32 | 64)) != 0, HAS_CB = (EPI & (2 | 64)) != 0, HAS_RES2 = (EPI & 16) != 0;
    auto seg_of = [&](int i, int src) {
        GwSeg<RB> sg;
        int row0, col0;
        tile_of(i, row0, col0);
        sg.ra = src ? rs.A1 : rs.A0;
        sg.rb = src ? rs.B1 : rs.B0;
        const unsigned lda = src ? g.lda[1] : g.lda[0], ldb = src ? g.ldb[1] : g.ldb[0];
        sg.lb = src ? g.lb[1] : g.lb[0];
        sg.nsteps = src ? g.T1 : g.T0;
        sg.ldb_bytes = ldb * 4u;
#pragma unroll
        for (int x = 0; x < RB; ++x) {
            unsigned row = (unsigned)(row0 + 32 * x + li);
            if (A2C && src) row = min(row, (unsigned)g.M - 1u) / (unsigned)g.rpc;      // a per-cloud row
            sg.voffA[x] = (row * lda + 4u * lh) * 4u;
        }
        sg.voffB = sg.lb == 1 ? (4u * lh * ldb + col0 + cs * li) * 4u : ((unsigned)(col0 + cs * li) * ldb + 4u * lh) * 4u;
        return sg;
    };
    if (count <= 0) return;
    u32x4 a[4][RB], b[4][4];
    GwSeg<RB> cur = seg_of(0, 0);
    gw_issue<RB, NCB, LB0, 0>(a, b, cur, 0); gw_issue<RB, NCB, LB0, 1>(a, b, cur, 1); gw_issue<RB, NCB, LB0, 2>(a, b, cur, 2);
    for (int i = 0; i < count; ++i) {
        f32x16 acc[RB][NCB];
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][c][r] = 0.f;
        // what the epilogue adds per column is requested NOW, a whole tile of MFMAs before it is used (fetched in the epilogue it
        // cost a memory round trip per tile, which the two waves of a SIMD -- in lockstep -- spent waiting together)
        int row0, col0;
        tile_of(i, row0, col0);
        const int colb = col0 + cs * li;
        const int c0 = row0 / g.rpc, nb = (c0 + 1) * g.rpc;
        float bv[NCB], w3v[NCB][3], cbv[2][NCB];
        const int c1 = min(c0 + 1, (g.M - 1) / g.rpc);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            bv[c] = 0.f;
            if constexpr (HAS_BIAS) bv[c] = g.bias[colb + c];
            if constexpr (HAS_XYZ) { w3v[c][0] = g.w3[(colb + c) * 3]; w3v[c][1] = g.w3[(colb + c) * 3 + 1]; w3v[c][2] = g.w3[(colb + c) * 3 + 2]; }
            if constexpr (HAS_CB) {
                cbv[0][c] = g.cbias[(size_t)c0 * g.N + colb + c];
                cbv[1][c] = g.cbias[(size_t)c1 * g.N + colb + c];
            }
        }
        // (past the last tile the ring re-reads the start of the first segment: harmless, never consumed)
        const GwSeg<RB> nfirst = seg_of(i + 1 < count ? i + 1 : i, 0);
        if (two) {
            const GwSeg<RB> second = seg_of(i, 1);
            gw_segment<RB, NCB, LB0, LB1>(acc, a, b, cur, second);
            gw_segment<RB, NCB, LB1, LB0>(acc, a, b, second, nfirst);
        } else {
            gw_segment<RB, NCB, LB0, LB0>(acc, a, b, cur, nfirst);
        }
        cur = nfirst;

        // ---- epilogue: accumulator r of (x, c) <-> row row0 + 32 x + (r&3) + 8 (r>>2) + 4 lh, column colb + c
#pragma unroll
        for (int x = 0; x < RB; ++x) {
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
            u32x4 rq[4], rq2[4];                              // four residual rows requested together
            if constexpr (HAS_RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r4 + q;
                    const int row = row0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const unsigned off = ((unsigned)row * g.ldr + colb) * 4u;
                    if constexpr (NCB == 4) rq[q] = __builtin_amdgcn_raw_buffer_load_b128(rs.R, off, 0, 0);
                    else if constexpr (NCB == 2) {
                        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs.R, off, 0, 0);
                        rq[q][0] = t[0]; rq[q][1] = t[1];
                    } else rq[q][0] = __builtin_amdgcn_raw_buffer_load_b32(rs.R, off, 0, 0);
                    if constexpr (HAS_RES2) {
                        const unsigned off2 = ((unsigned)row * g.ldr2 + colb) * 4u;
                        if constexpr (NCB == 4) rq2[q] = __builtin_amdgcn_raw_buffer_load_b128(rs.R2, off2, 0, 0);
                        else if constexpr (NCB == 2) {
                            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs.R2, off2, 0, 0);
                            rq2[q][0] = t[0]; rq2[q][1] = t[1];
                        } else rq2[q][0] = __builtin_amdgcn_raw_buffer_load_b32(rs.R2, off2, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r4 + q;
                const int row = row0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int rowc = min(row, g.M - 1);
                float v[NCB];
                if constexpr (EXACT) {
                    // ((conv2 chain [+ its f_global block]) + F) + STE: gcn3d.py:112 / :186 then :90 / :156, each sum rounded once
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] = acc[x][c][r];
                    if constexpr (HAS_CB) {
#pragma unroll
                        for (int c = 0; c < NCB; ++c) v[c] += rowc >= nb ? cbv[1][c] : cbv[0][c];
                    }
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] += u2f(rq[q][c]);
                    if constexpr (HAS_RES2) {
#pragma unroll
                        for (int c = 0; c < NCB; ++c) v[c] += u2f(rq2[q][c]);
                    }
                    if constexpr (HAS_XYZ) {
                        const float* p3 = g.xyz3 + (size_t)rowc * 3;
                        const float px = p3[0], py = p3[1], pz = p3[2];
#pragma unroll
                        for (int c = 0; c < NCB; ++c) v[c] += __fmaf_rn(pz, w3v[c][2], __fmaf_rn(py, w3v[c][1], px * w3v[c][0]));
                    }
                } else {
#pragma unroll
                for (int c = 0; c < NCB; ++c) v[c] = g.alpha * acc[x][c][r] + bv[c];
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] += u2f(rq[q][c]);
                }
                if constexpr (HAS_XYZ) {       // the K = 3 product on raw fp32 coordinates (HSlayer_surface's STE, gcn3d.py:85)
                    const float* p3 = g.xyz3 + (size_t)rowc * 3;
                    const float px = p3[0], py = p3[1], pz = p3[2];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] += __fmaf_rn(pz, w3v[c][2], __fmaf_rn(py, w3v[c][1], px * w3v[c][0]));
                }
                if constexpr (HAS_CB) {          // (the host sends tiles that could span three clouds to gemm_rows)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] += rowc >= nb ? cbv[1][c] : cbv[0][c];
                }
                }
                if constexpr (HAS_RELU) {         // FaceRecon.py:88: relu(conv_0(...)) in the producing kernel
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] = fmaxf(v[c], 0.f);
                }
                const unsigned off = ((unsigned)row * g.ldc + colb) * 4u;       // rows >= M fall outside the descriptor: dropped
#if GW_ABLATE == 3
                if (g.alpha == 12345.f)
#endif
                if constexpr (NCB == 4) {
                    u32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = __float_as_uint(v[c]);
#if GW_ABLATE == 4
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs.C, off, 0, 2);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs.C, off, 0, 0);
#endif
                } else if constexpr (NCB == 2) {
                    u32x2 o;
                    o[0] = __float_as_uint(v[0]); o[1] = __float_as_uint(v[1]);
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs.C, off, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), rs.C, off, 0, 0);
                }
            }
            }
        }
    }
}

// LB0 / LB1: layout of B1 / B2 (0 "nt", 1 "nn"; a single-source product is instantiated with LB1 == LB0)
template <int RB, int NCB, int WPS, int LB0, int LB1, bool TWO, int EPI, bool A2C = false>
__global__ __launch_bounds__(256, WPS) void gemm_wave_kernel(const GwArgs g) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * 4 + wv;
    if (w >= g.nwaves) return;
    GwRsrc rs;
    rs.A0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A[0]), 0, (int)((((size_t)g.M - 1) * g.lda[0] + g.K[0]) * 4), 0x00020000);
    const size_t b0_elems = g.lb[0] == 1 ? ((size_t)g.K[0] - 1) * g.ldb[0] + g.N : ((size_t)g.N - 1) * g.ldb[0] + g.K[0];
    rs.B0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B[0]), 0, (int)(b0_elems * 4), 0x00020000);
    constexpr bool two = TWO;
    const size_t a1_elems = two ? ((size_t)(A2C ? (g.M - 1) / g.rpc : g.M - 1)) * g.lda[1] + g.K[1] : 1;
    const size_t b1_elems = two ? (g.lb[1] == 1 ? ((size_t)g.K[1] - 1) * g.ldb[1] + g.N : ((size_t)g.N - 1) * g.ldb[1] + g.K[1]) : 1;
    rs.A1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(two ? g.A[1] : g.A[0]), 0, (int)(a1_elems * 4), 0x00020000);
    rs.B1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(two ? g.B[1] : g.B[0]), 0, (int)(b1_elems * 4), 0x00020000);
    rs.C = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)((((size_t)g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
    constexpr bool has_res = (EPI & (2 | 32 | 64)) != 0;
    rs.R = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? g.resid : g.C), 0,
                                             (int)((((size_t)g.M - 1) * (has_res ? g.ldr : g.ldc) + g.N) * 4), 0x00020000);
    rs.R2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((EPI & 16) ? g.resid2 : g.C), 0,
                                              (int)((((size_t)g.M - 1) * ((EPI & 16) ? g.ldr2 : g.ldc) + g.N) * 4), 0x00020000);

    // tile order: row panels fastest (the waves of a CU work on ONE column panel at a time and share its B lines in L1) or
    // column panels fastest (they share the A rows)
    auto tile_rc = [&](int u, int& row0, int& col0) {
        const int tm = g.order ? u % g.TM : u / g.TN;
        const int tn = g.order ? u / g.TM : u - tm * g.TN;
        row0 = tm * 32 * RB; col0 = tn * 32 * NCB;
    };
    // whole tiles [w base, (w+1) base)
    const int first = w * g.base;
    gw_tiles<RB, NCB, LB0, LB1, TWO, EPI, A2C>(g, rs, g.base, NCB, [&](int i, int& row0, int& col0) { tile_rc(first + i, row0, col0); });
    // the 32 x 32 blocks of the leftover tiles, dealt round-robin (usually at most one per wave): block (xb, cb) of tile u_rem + w / (RB NCB) -- rows 32 xb .., columns cb + NCB li
    for (int pc = w; pc < g.npieces; pc += g.nwaves) {
        const int u = g.u_rem + pc / (RB * NCB), sub = pc % (RB * NCB);
        int row0, col0;
        tile_rc(u, row0, col0);
        row0 += 32 * (sub / NCB); col0 += sub % NCB;
        gw_tiles<1, 1, LB0, LB1, TWO, EPI, A2C>(g, rs, 1, NCB, [&](int, int& r0, int& c0) { r0 = row0; c0 = col0; });
    }
}

// ---- host side: tile shape and cut ---------------------------------------------------------------------------------
struct GwPlan { int RB, NCB, wps, TM, TN, base, nwaves, u_rem, npieces; };

// cfg: 0 = automatic; otherwise RB | NCB << 4 | (waves per SIMD) << 16 | order << 28  (profiling / tuning)
static bool gw_plan(int M, int N, int K1, int K2, int cfg, GwPlan* p) {
    if (M <= 0 || N <= 0 || K1 <= 0 || (K1 & 31) || (K2 & 31) || K2 < 0 || (N & 31)) return false;
    int RB = cfg & 15, NCB = (cfg >> 4) & 15, wps = (cfg >> 16) & 15;
    auto tiles = [&](int rb, int ncb) { return (long long)((M + 32 * rb - 1) / (32 * rb)) * (N / (32 * ncb)); };
    if (!RB) RB = 1;
    // tile width (measured over the layer shapes at B=16, N=1028, tools/bench_gemm_wave.py): the widest tile that still gives every
    // wave two tiles or more (the second tile's first loads fly under the first one's MFMAs); failing that the widest one with
    // ~500 tiles; else 32 x 32
    if (!NCB) {
        const bool w4 = N % 128 == 0, w2 = N % 64 == 0;
        if (w4 && tiles(RB, 4) >= 4096) NCB = 4;
        else if (w2 && tiles(RB, 2) >= 4096) NCB = 2;
        else if (w4 && tiles(RB, 4) >= 500) NCB = 4;
        else if (w2 && tiles(RB, 2) >= 500) NCB = 2;
        else NCB = 1;
    }
    if (N % (32 * NCB)) return false;
    if (RB != 1 && RB != 2) return false;
    if (NCB != 1 && NCB != 2 && NCB != 4) return false;
    if (RB == 2 && NCB != 4) return false;                 // (not instantiated)
    if (!wps) wps = 2;
    wps = (RB == 2 && NCB == 4) ? 1 : 2;                   // 128 accumulator registers + the ring: one wave per SIMD
    const int nw_max = HSP_NUM_CU * 4 * wps;
    const long long U = tiles(RB, NCB);
    if (U > (1ll << 30)) return false;
    int nwaves = (int)std::min<long long>(U, nw_max);
    const int base = (int)(U / nwaves);
    const int rem = (int)(U - (long long)base * nwaves);
    const int npieces = rem * RB * NCB;
    p->RB = RB; p->NCB = NCB; p->wps = wps;
    p->TM = (M + 32 * RB - 1) / (32 * RB); p->TN = N / (32 * NCB);
    p->base = base; p->nwaves = nwaves; p->u_rem = base * nwaves; p->npieces = npieces;
    return true;
}

}  // namespace hsp

using namespace hsp;

static inline bool has_rc_check(const void* r, const void* c) { return r && c; }

/* host-only: the cut chosen for a shape -> out[10] = RB, NCB, waves per SIMD, tiles_m, tiles_n, tiles per wave, waves,
 * first leftover tile, 32 x 32 blocks of the leftover tiles, 0; returns 0 when the shape is not covered */
extern "C" int hsp_gemm_wave_plan_info(int M, int N, int K1, int K2, int cfg, int* out) {
    GwPlan p;
    if (!out || !gw_plan(M, N, K1, K2, cfg, &p)) return 0;
    const int v[10] = {p.RB, p.NCB, p.wps, p.TM, p.TN, p.base, p.nwaves, p.u_rem, p.npieces, 0};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    return 1;
}

/* 1 when hsp_gemm_wave_f32 covers the shape (K1, K2 multiples of 32, N of 32; every operand row must be 16-byte aligned) */
extern "C" int hsp_gemm_wave_supported(int M, int N, int K1, int K2, int cfg) {
    GwPlan p;
    return gw_plan(M, N, K1, K2, cfg, &p) ? 1 : 0;
}

/* The HS layer's out product in the REFERENCE's order of operations (gcn3d.py:111-112,185-186 then :90,:156), every product a
 * k-ordered fp32 fma chain from 0 -- what the reference's CPU GEMM runs for K <= 256 -- so that the rows handed to the next
 * layer's feature-space neighbour search carry the reference's bits:
 *   two_chain == 0 (2 C <= 256: conv2 over cat[F, f_global] is ONE chain):
 *       out = ((chain_k F[r][k] Wa[n][k]  ->continued->  chain_k fg[r / rows_per_cloud][k] Wb[n][k]) + F[r][n]) + tail
 *   two_chain == 1 (2 C = 512: the CPU GEMM sums two K = 256 block chains):  t = chain(fg Wb^T) comes in as cloud_t (B, N):
 *       out = (((chain_k F[r][k] Wa[n][k]) + cloud_t[r / rows_per_cloud][n]) + F[r][n]) + tail
 *   tail = ste[r][n] (the STE product, its own chain: a plain hsp_gemm_wave_f32 "nt" call) or, for the surface layer (xyz3 != NULL),
 *   the K = 3 chain xyz3[r] . w3[n]; relu != 0 applies FaceRecon.py:88's relu.
 * Shapes: C a multiple of 32, 16-byte aligned rows; rows_per_cloud >= 32. */
extern "C" int hsp_layer_out_exact_f32(const float* F, int ldf, const float* Wa, int ldwa, const float* fg, int ldfg,
                                       const float* Wb, int ldwb, const float* cloud_t, int two_chain, const float* ste, int ldste,
                                       const float* xyz3, const float* w3, int relu, int M, int C, int rows_per_cloud, float* out,
                                       int ldo, hspStream_t stream) {
    if (!F || !Wa || !out || M <= 0 || C <= 0 || rows_per_cloud <= 0) return HSP_ERR_BAD_ARG;
    if (two_chain ? !cloud_t : (!fg || !Wb)) return HSP_ERR_BAD_ARG;
    if ((ste == nullptr) == (xyz3 == nullptr) || (xyz3 && !w3)) return HSP_ERR_BAD_ARG;
    if (relu && !xyz3) return HSP_ERR_UNSUPPORTED;
    GwPlan p;
    if (!gw_plan(M, C, C, two_chain ? 0 : C, 0, &p)) return HSP_ERR_UNSUPPORTED;
    if (rows_per_cloud < 32 * p.RB) return HSP_ERR_UNSUPPORTED;
    auto al16 = [](const void* q, int ld) { return ((reinterpret_cast<size_t>(q) | ((size_t)ld * 4)) & 15) == 0; };
    if (!al16(F, ldf) || !al16(Wa, ldwa) || !al16(out, ldo) || (!two_chain && (!al16(fg, ldfg) || !al16(Wb, ldwb))) ||
        (ste && !al16(ste, ldste)))
        return HSP_ERR_UNSUPPORTED;
    GwArgs g{};
    g.A[0] = F; g.B[0] = Wa; g.lda[0] = ldf; g.ldb[0] = ldwa; g.K[0] = C; g.lb[0] = 0;
    g.A[1] = two_chain ? nullptr : fg; g.B[1] = two_chain ? nullptr : Wb; g.lda[1] = ldfg; g.ldb[1] = ldwb;
    g.K[1] = two_chain ? 0 : C; g.lb[1] = 0;
    g.C = out; g.ldc = ldo; g.M = M; g.N = C; g.resid = F; g.ldr = ldf; g.resid2 = ste; g.ldr2 = ldste;
    g.cbias = cloud_t; g.rpc = rows_per_cloud; g.xyz3 = xyz3; g.w3 = w3; g.alpha = 1.f;
    g.TM = p.TM; g.TN = p.TN; g.T0 = C / 8; g.T1 = two_chain ? 0 : C / 8;
    g.base = p.base; g.nwaves = p.nwaves; g.u_rem = p.u_rem; g.npieces = p.npieces; g.order = 0;
    const dim3 grid((unsigned)((p.nwaves + 3) / 4)), block(256);
    hipStream_t st = as_stream(stream);
    // forms: A one chain + F + ste (32|16, A2C)   B one chain + F + xyz (32|4 [|8], A2C)   C chain + t + F + ste (64|16)
    //        D chain + t + F + xyz (64|4 [|8])
#define GWX_K(R, C_, W_) \
    do { \
        if (!two_chain && ste) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, true, 32 | 16, true>), grid, block, 0, st, g); \
        else if (!two_chain && relu) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, true, 32 | 4 | 8, true>), grid, block, 0, st, g); \
        else if (!two_chain) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, true, 32 | 4, true>), grid, block, 0, st, g); \
        else if (ste) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, false, 64 | 16>), grid, block, 0, st, g); \
        else if (relu) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, false, 64 | 4 | 8>), grid, block, 0, st, g); \
        else hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, 0, 0, false, 64 | 4>), grid, block, 0, st, g); \
    } while (0)
    if (p.RB == 1 && p.NCB == 4) GWX_K(1, 4, 2);
    else if (p.RB == 1 && p.NCB == 2) GWX_K(1, 2, 2);
    else if (p.RB == 1 && p.NCB == 1) GWX_K(1, 1, 2);
    else return HSP_ERR_UNSUPPORTED;
#undef GWX_K
    return check_launch();
}

extern "C" int hsp_gemm_wave_f32(const float* A1, int lda1, const float* B1, int ldb1, int b1_layout, int K1,
                                 const float* A2, int lda2, const float* B2, int ldb2, int b2_layout, int K2, int M, int N,
                                 const float* bias, const float* resid, int ldr, const float* cloud_bias,
                                 int rows_per_cloud, float alpha, const float* xyz3, const float* w3, float* C, int ldc,
                                 int cfg, hspStream_t stream) {
    if (!A1 || !B1 || !C || M <= 0 || N <= 0 || K1 <= 0 || lda1 < K1 || ldc < N) return HSP_ERR_BAD_ARG;
    if ((b1_layout != 0 && b1_layout != 1) || ldb1 < (b1_layout == 0 ? K1 : N)) return HSP_ERR_BAD_ARG;
    const bool two = A2 != nullptr;
    if (two && (!B2 || K2 <= 0 || lda2 < K2 || (b2_layout != 0 && b2_layout != 1) || ldb2 < (b2_layout == 0 ? K2 : N)))
        return HSP_ERR_BAD_ARG;
    if (!two) K2 = 0;
    if (resid && ldr < N) return HSP_ERR_BAD_ARG;
    if (cloud_bias && rows_per_cloud <= 0) return HSP_ERR_BAD_ARG;
    if ((xyz3 == nullptr) != (w3 == nullptr)) return HSP_ERR_BAD_ARG;
    GwPlan p;
    if (!gw_plan(M, N, K1, K2, cfg, &p)) return HSP_ERR_UNSUPPORTED;
    auto al16 = [](const void* q, int ld) { return ((reinterpret_cast<size_t>(q) | ((size_t)ld * 4)) & 15) == 0; };
    if (!al16(A1, lda1) || !al16(B1, ldb1) || !al16(C, ldc) || (two && (!al16(A2, lda2) || !al16(B2, ldb2))) ||
        (resid && !al16(resid, ldr)))
        return HSP_ERR_UNSUPPORTED;
    GwArgs g{};
    g.A[0] = A1; g.B[0] = B1; g.lda[0] = lda1; g.ldb[0] = ldb1; g.K[0] = K1; g.lb[0] = b1_layout;
    g.A[1] = A2; g.B[1] = B2; g.lda[1] = lda2; g.ldb[1] = ldb2; g.K[1] = K2; g.lb[1] = b2_layout;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.cbias = cloud_bias; g.rpc = rows_per_cloud > 0 ? rows_per_cloud : 1;
    g.xyz3 = xyz3; g.w3 = w3; g.alpha = alpha;
    g.TM = p.TM; g.TN = p.TN; g.T0 = K1 / 8; g.T1 = K2 / 8;
    g.base = p.base; g.nwaves = p.nwaves; g.u_rem = p.u_rem; g.npieces = p.npieces;
    g.order = (cfg >> 28) & 1;
    if (((cfg >> 29) & 1) && !(xyz3 && has_rc_check(resid, cloud_bias))) return HSP_ERR_UNSUPPORTED;   // relu: surface form only
    const dim3 grid((unsigned)((p.nwaves + 3) / 4)), block(256);
    hipStream_t st = as_stream(stream);
    // instantiated forms (anything else: HSP_ERR_UNSUPPORTED -> hsp_gemm_rows_f32):
    //   0 fm   "nn"        + bias                      1 g W      "nn"                      2 x W^T  "nt"
    //   3 x W^T "nt" + bias                            4 out      "nt" + "nt", residual + cloud bias
    //   5 out0  "nt", residual + cloud bias + xyz3 (7: + relu, cfg bit 29)     6 gX       "nn" + "nt"
    const bool has_rc = resid && cloud_bias;
    int form = -1;
    if (!two && b1_layout == 1 && bias && !resid && !cloud_bias && !xyz3) form = 0;
    else if (!two && b1_layout == 1 && !bias && !resid && !cloud_bias && !xyz3) form = 1;
    else if (!two && b1_layout == 0 && !bias && !resid && !cloud_bias && !xyz3) form = 2;
    else if (!two && b1_layout == 0 && bias && !resid && !cloud_bias && !xyz3) form = 3;
    else if (two && b1_layout == 0 && b2_layout == 0 && !bias && has_rc && !xyz3) form = 4;
    else if (!two && b1_layout == 0 && !bias && has_rc && xyz3) form = ((cfg >> 29) & 1) ? 7 : 5;       // 7: + relu
    else if (two && b1_layout == 1 && b2_layout == 0 && !bias && !resid && !cloud_bias && !xyz3) form = 6;
    if (form < 0) return HSP_ERR_UNSUPPORTED;
    if (has_rc && g.rpc < 32 * p.RB) return HSP_ERR_UNSUPPORTED;       // a tile may span at most two clouds
#define GW_K(R, C_, W_, L0, L1, T_, E_) hipLaunchKernelGGL((gemm_wave_kernel<R, C_, W_, L0, L1, T_, E_>), grid, block, 0, st, g)
#define GW_LAUNCH(R, C_, W_)                                   \
    do {                                                       \
        switch (form) {                                        \
            case 0: GW_K(R, C_, W_, 1, 1, false, 1); break;    \
            case 1: GW_K(R, C_, W_, 1, 1, false, 0); break;    \
            case 2: GW_K(R, C_, W_, 0, 0, false, 0); break;    \
            case 3: GW_K(R, C_, W_, 0, 0, false, 1); break;    \
            case 4: GW_K(R, C_, W_, 0, 0, true, 2); break;     \
            case 5: GW_K(R, C_, W_, 0, 0, false, 6); break;    \
            case 7: GW_K(R, C_, W_, 0, 0, false, 14); break;   \
            default: GW_K(R, C_, W_, 1, 0, true, 0); break;    \
        }                                                      \
    } while (0)
    if (p.RB == 2 && p.NCB == 4) GW_LAUNCH(2, 4, 1);
    else if (p.RB == 1 && p.NCB == 4) GW_LAUNCH(1, 4, 2);
    else if (p.RB == 1 && p.NCB == 2) GW_LAUNCH(1, 2, 2);
    else GW_LAUNCH(1, 1, 2);
#undef GW_LAUNCH
#undef GW_K
    return check_launch();
}
