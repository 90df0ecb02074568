// gemm_rows.hip -- the dense per-point products of the HS stack and of the heads on the matrix cores of gfx950,
// hand-written (no BLAS library on this path), fp32 and bf16, with the layer's element-wise tail fused in.
//
//   C[r][n] = sum_k A1[r][k] * op(B1)[k][n]  (+ sum_k A2[r][k] * op(B2)[k][n])          r = point row (B*N of them)
//             (+ bias[n]) (+ resid[r][n]) (+ cloud_bias[r / rows_per_cloud][n])
//
// which covers, with ONE launch each (reference network/fs_net_repo/gcn3d.py):
//   fm  = X W + b                                   :171    B1 = weights (Cin,(S+1)Cout), "nn" layout, bias
//   out = X Wste^T + F Wa^T + F + t[cloud]          :149,:186,:156   two sources, residual, per-cloud bias
//   gF  = g Wa                                      backward of :186  ("nn", ldb = 2C)
//   gX  = g Wste + gfm W^T                          backward of :149 and :171, two sources ("nn" + "nt")
//   y   = x W^T + b / gx = g W                      the Conv1d(k=1) layers of the heads (PoseR.py:16-39, PoseTs.py:18-45,
//                                                   FaceRecon.py:37-68)
//
// Structure: persistent 256-thread workgroups (4 waves as 2x2), each walking its share of the BM x BN tiles of C
// (128x128 or 64x64: the row counts of this path are 257 * 2^j, so the shape is chosen per call by how evenly the tiles
// fill 256 CUs).  K is walked in blocks of 128 BYTES per row (32 fp32 / 64 bf16) through a double-buffered LDS stage, ONE
// barrier per block; the (tile, k-block) iteration space of a workgroup is flattened, so the first block of the next tile
// is already in flight while the current tile finishes and is written out (with K = 128 a tile is only four blocks long:
// an un-overlapped prologue per tile cost as much as the MFMAs).
// Staging: when every operand is 16-byte aligned, global -> LDS directly (global_load_lds_dwordx4, no staging registers,
// no ds_write); otherwise global -> registers (4-byte pieces, any alignment) -> ds_write after the block's MFMAs.
// LDS image, identical for both types: tile rows of 128 bytes = 8 chunks of 16 bytes, chunk c of row r stored at
// position c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group (16 different rows, same c) hit 16 different
// 16-byte slots, and a wave's LDS-DMA destination stays lane-linear (the permutation is applied to the SOURCE address).
// Reads at row*128 + ((2*step + (lane>>5)) ^ key)*16 hand every lane
//   fp32: 4 consecutive k  -> 4 x v_mfma_f32_32x32x2_f32  (k pairs (j, j+4): the k order inside a block is permuted
//                              identically for A and B, the sum is the same set of products)
//   bf16: 8 consecutive k  -> 1 x v_mfma_f32_32x32x16_bf16
// "nn" operands (k-major rows, fp32 only) are staged as [k][n] rows and read with ds_read_b32 (lanes = consecutive n).
// Rows past M / columns past N are clamped on load and masked on store; in the ragged last block of K the operands are
// zeroed as they are read from LDS, so any K (3, 771, 1286, 1289 ...) is accepted.
// Tile order: tm fastest inside groups of 16 row panels, then tn -- the ~128 tiles an XCD works on at a time form a
// 16 x 8 rectangle that shares 16 A panels and 8 B panels in that XCD's L2.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct GemmRowsArgs {
    const void* A1; const void* B1; const void* A2; const void* B2;
    int lda1, ldb1, K1;
    int lda2, ldb2, K2;
    void* C; int ldc;
    int M, N;
    const float* bias;                     // (N) fp32 or null
    const void* resid; int ldr;            // (M,N) of the output type or null
    const float* cbias; int rows_per_cloud;  // (ceil(M / rows_per_cloud), N) fp32 or null
    float alpha;                           // scales the accumulated products (before bias / resid / cloud bias)
    const float* xyz3; const float* w3;    // rank-3 fp32 update + xyz3[row] . w3[col] ((M,3), (N,3)) or null
    int c_f32;                             // bf16 operands: write C as fp32 (a tensor that feeds BatchNorm keeps its mantissa)
    int nsplit; float* ws;                 // split-K: nsplit > 1 -> raw fp32 partial tiles to ws[split][M][N], folded (with the
                                           // whole epilogue) by gemm_rows_reduce_kernel
    int tiles_m, tiles_n;
};

#define GR_TM_GROUP 16                     // row panels per tile-order group

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {              // round to nearest even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// T: float or unsigned short (bf16 bits).  WM, WN: 32x32 MFMA tiles per wave along M / N (block tile = 64*WM x 64*WN).
// LB1 / LB2: layout of B1 / B2 -- 1 "nt" (N,K) k contiguous, 2 "nn" (K,N) n contiguous (fp32 only), 0 (LB2) = no 2nd source.
// MODE 1: every operand 16-byte aligned (base and row pitch): 16-byte staging loads; MODE 2: 8-byte aligned (an even fp32
// pitch such as 1286): 8-byte pieces; MODE 0: 4-byte pieces (any alignment).
template <typename T, int WM, int WN, int LB1, int LB2, int MODE>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const GemmRowsArgs g) {
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;                 // elements per 16-byte chunk
    constexpr int BKE = 128 / ES;                // k elements per block
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;      // ("nn": 32 k-rows of BN fp32 -- the same size)
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NN_PITCH = BN * 4;
    constexpr int NA = BM / 32, NB = BN / 32;    // staging pieces per thread (16-byte chunks)
    static_assert(LB1 == 1 || (LB1 == 2 && ES == 4), "nn operands are fp32 only");
    static_assert(LB2 == 0 || LB2 == 1 || (LB2 == 2 && ES == 4), "nn operands are fp32 only");
    static_assert(MODE == 1 || ES == 4, "bf16 operands must be 16-byte aligned");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int li = lane & 31, lh = lane >> 5;
    const int key = (li >> 1) & 7;               // swizzle key of the rows this lane reads (tile row offsets are multiples of 32)

    // ---- work items of this workgroup: (tile, K split) pairs.  Workgroups b = x (mod 8) run on XCD x: it takes a
    // contiguous range of the ordered item list, and its workgroups take every (gridDim/8)-th item of that range, so the
    // items in flight on an XCD are consecutive in the order below.
    const int nitems = g.tiles_m * g.tiles_n * g.nsplit;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int q8 = nitems >> 3, r8 = nitems & 7;
    const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int x_count = q8 + (xcd < r8 ? 1 : 0);
    const int my_count = idx < x_count ? (x_count - idx + per - 1) / per : 0;
    if (my_count == 0) return;

    const int T1 = (g.K1 + BKE - 1) / BKE;
    const int T2 = LB2 ? (g.K2 + BKE - 1) / BKE : 0;
    const int TT = T1 + T2;
    const int Tper = (TT + g.nsplit - 1) / g.nsplit;           // k-blocks per split (the host keeps every split non-empty)

    // position in the flattened (item, k-block) walk
    struct Pos { int ti, t, m0, n0, ks, k1; bool ok; };
    auto item = [&](int i, Pos& p) {
        // ordered tile id -> (tm, tn): groups of GR_TM_GROUP row panels; inside a group tm fastest, then tn
        const int w = x_first + idx + i * per;
        const int o = w / g.nsplit;
        p.ks = w - o * g.nsplit;
        p.t = p.ks * Tper; p.k1 = min(TT, p.t + Tper);
        const int gsz = GR_TM_GROUP * g.tiles_n;
        const int grp = o / gsz, rem = o - grp * gsz;
        const int gh = min(GR_TM_GROUP, g.tiles_m - grp * GR_TM_GROUP);          // panels in this (maybe last, shorter) group
        const int tn = rem / gh, tm = grp * GR_TM_GROUP + (rem - tn * gh);
        p.m0 = tm * BM; p.n0 = tn * BN; p.ti = i; p.ok = true;
    };
    auto next = [&](const Pos& p) {
        Pos q = p;
        if (!p.ok) return q;
        q.t = p.t + 1;
        if (q.t == p.k1) {
            if (p.ti + 1 < my_count) item(p.ti + 1, q);
            else q.ok = false;
        }
        return q;
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- staging: thread -> 16-byte chunk p = tid & 7 (physical position) of tile rows (tid >> 3) + 32 j; the logical chunk
    // is p ^ rowkey (rows s_row + 32 j share the key)
    const int s_row = tid >> 3, s_p = tid & 7;
    const int s_c = s_p ^ ((s_row >> 1) & 7);

    auto layout_of = [&](int t) { return (LB2 == 0 || LB1 == LB2) ? LB1 : ((LB2 && t >= T1) ? LB2 : LB1); };
    // 16 bytes, no branches: a piece at or past `lim` is redirected to element 0 (garbage, zeroed on the LDS read path)
    auto load16 = [&](const char* row, int e0, int lim) {
        if constexpr (MODE == 1) {
            return *reinterpret_cast<const uint4*>(row + (size_t)(e0 < lim ? e0 : 0) * ES);
        } else if constexpr (MODE == 2) {
            const uint2 lo = *reinterpret_cast<const uint2*>(row + (size_t)(e0 < lim ? e0 : 0) * 4);
            const uint2 hi = *reinterpret_cast<const uint2*>(row + (size_t)(e0 + 2 < lim ? e0 + 2 : 0) * 4);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = *reinterpret_cast<const unsigned*>(row + (size_t)(e0 + e < lim ? e0 + e : 0) * 4);
            return make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    auto fetch = [&](const Pos& p, uint4 (&ra)[NA], uint4 (&rb)[NB]) {        // global loads of k-block p.t of tile p
        const bool second = LB2 && p.t >= T1;
        const char* A = reinterpret_cast<const char*>(second ? g.A2 : g.A1);
        const char* B = reinterpret_cast<const char*>(second ? g.B2 : g.B1);
        const int lda = second ? g.lda2 : g.lda1, ldb = second ? g.ldb2 : g.ldb1, K = second ? g.K2 : g.K1;
        const int kb = (second ? p.t - T1 : p.t) * BKE;
        const int k0 = kb + s_c * EPC;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = min(p.m0 + s_row + 32 * j, g.M - 1);
            ra[j] = load16(A + (size_t)row * lda * ES, k0, K);
        }
        if (layout_of(p.t) == 1) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int col = min(p.n0 + s_row + 32 * j, g.N - 1);
                rb[j] = load16(B + (size_t)col * ldb * ES, k0, K);
            }
        } else {
            // "nn": the tile is 32 k-rows of BN fp32, linear; thread -> 16 bytes at byte (tid + 256 j) * 16
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int off = (tid + 256 * j) * 16;
                const int kr = off / NN_PITCH, nc = (off - kr * NN_PITCH) >> 2;
                const int k = kb + kr < K ? kb + kr : 0;
                rb[j] = load16(B + (size_t)k * ldb * 4, p.n0 + nc, g.N);
            }
        }
    };
    // zero the elements of a 16-byte piece at k >= lim (k0 = k of its first element)
    auto keep_k = [&](uint4 v, int k0, int lim) {
        unsigned w[4] = {v.x, v.y, v.z, v.w};
        if constexpr (ES == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = k0 + e < lim ? w[e] : 0u;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] &= (k0 + 2 * e < lim ? 0x0000ffffu : 0u) | (k0 + 2 * e + 1 < lim ? 0xffff0000u : 0u);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    // staged registers -> LDS.  The ragged last block of a source is zero-filled past K here (a uniform branch, rare), so the
    // MFMA loop has one shape
    auto stash = [&](const Pos& p, int buf, uint4 (&ra)[NA], uint4 (&rb)[NB]) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const bool second = LB2 && p.t >= T1;
        const int K = second ? g.K2 : g.K1;
        const int kb = (second ? p.t - T1 : p.t) * BKE;
        const bool nt = layout_of(p.t) == 1;
        if (K - kb < BKE) {
            const int k0 = kb + s_c * EPC;
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = keep_k(ra[j], k0, K);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (nt) rb[j] = keep_k(rb[j], k0, K);
                else if (kb + (tid + 256 * j) * 16 / NN_PITCH >= K) rb[j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4*>(sa + (s_row + 32 * j) * 128 + s_p * 16) = ra[j];
        if (nt) {
#pragma unroll
            for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(sb + (s_row + 32 * j) * 128 + s_p * 16) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4*>(sb + (tid + 256 * j) * 16) = rb[j];
        }
    };

    // ---- operand fragments of one k-block: step s = 32 bytes of k per row (8 fp32 / 16 bf16); this lane holds the half lh
    // (two fragment sets alive at a time: step s lives in set s & 1)
    uint4 fa[2][WM], fb[2][WN];
    auto read_frags = [&](int buf, int lb, int s) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
        const int coff = ((2 * s + lh) ^ key) << 4;
        const char* pa = sa + (wm0 + li) * 128 + coff;
#pragma unroll
        for (int x = 0; x < WM; ++x) fa[s & 1][x] = *reinterpret_cast<const uint4*>(pa + x * 32 * 128);
        if (lb == 1) {
            const char* pb = sb + (wn0 + li) * 128 + coff;
#pragma unroll
            for (int y = 0; y < WN; ++y) fb[s & 1][y] = *reinterpret_cast<const uint4*>(pb + y * 32 * 128);
        } else {
            const char* pb = sb + (8 * s + 4 * lh) * NN_PITCH + (wn0 + li) * 4;
#pragma unroll
            for (int y = 0; y < WN; ++y) {
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const unsigned*>(pb + j * NN_PITCH + y * 128);
                fb[s & 1][y] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    auto mma = [&](int s) {
        if constexpr (ES == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < WM; ++x) {
                    const uint4 a = fa[s & 1][x];
                    const float av = __uint_as_float(j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : a.w);
#pragma unroll
                    for (int y = 0; y < WN; ++y) {
                        const uint4 b = fb[s & 1][y];
                        const float bv = __uint_as_float(j == 0 ? b.x : j == 1 ? b.y : j == 2 ? b.z : b.w);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x][y], 0, 0, 0);
                    }
                }
        } else {
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][x]),
                                                                        __builtin_bit_cast(bf16x8, fb[s & 1][y]), acc[x][y], 0, 0, 0);
        }
    };

    // ---- write a finished item: accumulator r of (x,y) <-> row wm0 + 32x + (r&3) + 8(r>>2) + 4 lh, column wn0 + 32y + li
    // (a row-per-lane form -- MFMA operands swapped, 8-byte stores of 4 consecutive columns -- was measured slower: a store
    // instruction then touches 32 rows instead of 2)
    auto epilogue = [&](int m0, int n0, int ks) {
        if (g.nsplit > 1) {                                   // raw partial tile; the reduce kernel applies the epilogue
            float* wsp = g.ws + (size_t)ks * g.M * g.N;
#pragma unroll
            for (int y = 0; y < WN; ++y) {
                const int col = n0 + wn0 + 32 * y + li;
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (col < g.N && row < g.M) wsp[(size_t)row * g.N + col] = acc[x][y][r];
                        acc[x][y][r] = 0.f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        const int rpc = g.rows_per_cloud;
        // bf16 output in column pairs: needs dword-aligned pairs (even N and pitch, 4-byte aligned base)
        const bool pack2 = ES == 2 && !g.c_f32 && ((g.N | g.ldc) & 1) == 0 && (reinterpret_cast<size_t>(g.C) & 3) == 0;
        int c0 = 0, nb = 0x7fffffff;                      // per-cloud bias: cloud boundaries by comparison, no per-row division
        if (g.cbias) { c0 = m0 / rpc; nb = (c0 + 1) * rpc; }
#pragma unroll
        for (int y = 0; y < WN; ++y) {
            const int col = n0 + wn0 + 32 * y + li;
            const bool cok = col < g.N;
            const float bv = (g.bias && cok) ? g.bias[col] : 0.f;
            float w30 = 0.f, w31 = 0.f, w32 = 0.f;
            if (g.xyz3 && cok) { w30 = g.w3[col * 3]; w31 = g.w3[col * 3 + 1]; w32 = g.w3[col * 3 + 2]; }
#pragma unroll
            for (int x = 0; x < WM; ++x) {
                float keep[16];                               // (bf16 output) finished values, packed in pairs below
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = g.alpha * acc[x][y][r] + bv;
                    acc[x][y][r] = 0.f;
                    keep[r] = 0.f;
                    if (!cok || row >= g.M) continue;
                    if (g.resid) {
                        if (ES == 4) v += reinterpret_cast<const float*>(g.resid)[(size_t)row * g.ldr + col];
                        else v += bf16_to_f32(reinterpret_cast<const unsigned short*>(g.resid)[(size_t)row * g.ldr + col]);
                    }
                    if (g.xyz3) {      // the K = 3 product on raw fp32 coordinates (HSlayer_surface's STE, gcn3d.py:85)
                        const float* p3 = g.xyz3 + (size_t)row * 3;
                        v += __fmaf_rn(p3[2], w32, __fmaf_rn(p3[1], w31, p3[0] * w30));
                    }
                    if (g.cbias) {
                        int c = c0;
                        if (row >= nb) c = c0 + 1 + (row - nb) / rpc;      // rare: the tile spans clouds
                        v += g.cbias[(size_t)c * g.N + col];
                    }
                    if (ES == 4 || g.c_f32) reinterpret_cast<float*>(g.C)[(size_t)row * g.ldc + col] = v;
                    else if (!pack2) reinterpret_cast<unsigned short*>(g.C)[(size_t)row * g.ldc + col] = f32_to_bf16(v);
                    else keep[r] = v;
                }
                if (ES == 2 && !g.c_f32 && pack2) {
                    // 2-byte stores run at half the rate of 4-byte ones (measured: the fp32-output form of the same product is
                    // faster): neighbouring lanes trade values so that the even lane of a pair owns both columns of the
                    // even register rows and the odd lane those of the odd rows -- 8 dword stores instead of 16 short ones
                    const bool odd = li & 1;
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const float mine = odd ? keep[2 * p + 1] : keep[2 * p];
                        const float give = odd ? keep[2 * p] : keep[2 * p + 1];
                        const float got = __shfl_xor(give, 1);
                        const int r = 2 * p + (odd ? 1 : 0);
                        const int row = m0 + wm0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const unsigned lo = f32_to_bf16(odd ? got : mine), hi = f32_to_bf16(odd ? mine : got);
                        if (row < g.M && (col | 1) < g.N)      // (pack2: N is even, so the pair is inside or outside together)
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(g.C) + (size_t)row * g.ldc + (col & ~1)) =
                                lo | (hi << 16);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // one 32 x 32 block at a time: keeps the live set of the unrolled body small
            }
        }
    };

    // ---- the walk over (item, k-block).  The staging pipeline runs ahead of the MFMAs and across item boundaries:
    //   at the top of a step  stage[cur] holds the step's own block (visible) and the staging registers hold the next block
    //   of the walk (`pn`; loads issued in the middle of the previous step, a whole step of MFMAs ago).  The step reads its
    //   first two fragment groups, runs a quarter of the MFMAs, reads the third group into the freed set, runs the second
    //   quarter, reads the fourth, writes the staging registers into stage[cur ^ 1] (last read before the previous barrier) and at once re-issues them for the
    //   block after, then runs the second half of the MFMAs over those ds_writes / loads and meets the other waves at ONE
    //   barrier.
    uint4 ra[NA], rb[NB];
    Pos pn;
    item(0, pn);
    fetch(pn, ra, rb);
    stash(pn, 0, ra, rb);
    pn = next(pn);
    if (pn.ok) fetch(pn, ra, rb);
    __syncthreads();
    int cur = 0;
    for (int i = 0; i < my_count; ++i) {
        Pos it;
        item(i, it);
        for (int t = it.t; t < it.k1; ++t) {
            const int lb = layout_of(t);
            read_frags(cur, lb, 0);
            read_frags(cur, lb, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            read_frags(cur, lb, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            read_frags(cur, lb, 3);
            if (pn.ok) {
                stash(pn, cur ^ 1, ra, rb);
                pn = next(pn);
                if (pn.ok) fetch(pn, ra, rb);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(2);
            mma(3);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            cur ^= 1;
        }
        epilogue(it.m0, it.n0, it.ks);
    }
}

// split-K second stage: C = alpha * sum_s ws[s] (+ the whole epilogue), one thread per output element
template <typename T>
__global__ __launch_bounds__(256) void gemm_rows_reduce_kernel(const GemmRowsArgs g) {
    const long long total = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int row = (int)(e / g.N), col = (int)(e - (long long)row * g.N);
        float s0 = 0.f, s1 = 0.f;
        int sp = 0;
        for (; sp + 1 < g.nsplit; sp += 2) { s0 += g.ws[(size_t)sp * total + e]; s1 += g.ws[(size_t)(sp + 1) * total + e]; }
        if (sp < g.nsplit) s0 += g.ws[(size_t)sp * total + e];
        float v = g.alpha * (s0 + s1);
        if (g.bias) v += g.bias[col];
        if (g.resid) {
            if (sizeof(T) == 4) v += reinterpret_cast<const float*>(g.resid)[(size_t)row * g.ldr + col];
            else v += bf16_to_f32(reinterpret_cast<const unsigned short*>(g.resid)[(size_t)row * g.ldr + col]);
        }
        if (g.xyz3) {
            const float* p3 = g.xyz3 + (size_t)row * 3;
            v += __fmaf_rn(p3[2], g.w3[col * 3 + 2], __fmaf_rn(p3[1], g.w3[col * 3 + 1], p3[0] * g.w3[col * 3]));
        }
        if (g.cbias) v += g.cbias[(size_t)(row / g.rows_per_cloud) * g.N + col];
        if (sizeof(T) == 4 || g.c_f32) reinterpret_cast<float*>(g.C)[(size_t)row * g.ldc + col] = v;
        else reinterpret_cast<unsigned short*>(g.C)[(size_t)row * g.ldc + col] = f32_to_bf16(v);
    }
}

// alignment of an operand (base pointer and row pitch): 16, 8 or 4 bytes
static int align_of(const void* p, int ld_elems, int es) {
    const size_t v = reinterpret_cast<size_t>(p) | ((size_t)ld_elems * es);
    return v % 16 == 0 ? 16 : v % 8 == 0 ? 8 : 4;
}

// splits of K for `tiles` output tiles and TT k-blocks: enough work items for ~2 per CU, >= 4 k-blocks per split, <= 16
static int gemm_rows_pick_split(long long tiles, int TT) {
    if (tiles >= 2 * HSP_NUM_CU || TT < 8) return 1;
    int ns = (int)((2 * HSP_NUM_CU + tiles - 1) / tiles);
    if (ns > TT / 4) ns = TT / 4;
    if (ns > 16) ns = 16;
    if (ns < 2) return 1;
    const int per = (TT + ns - 1) / ns;
    return (TT + per - 1) / per;                   // every split non-empty
}

template <typename T, int WM, int WN, int MODE>
static int launch_cfg(const GemmRowsArgs& a, int lb1, int lb2, float* a_ws, size_t a_ws_bytes, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int ES = sizeof(T);
    GemmRowsArgs g = a;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = (a.N + BN - 1) / BN;
    {   // split-K when the tiles alone cannot fill the chip and K is deep (the input-gradient products: 64 tiles x K = 4608)
        constexpr int BKE = 128 / ES;
        const int TT = (a.K1 + BKE - 1) / BKE + (lb2 ? (a.K2 + BKE - 1) / BKE : 0);
        int ns = gemm_rows_pick_split((long long)g.tiles_m * g.tiles_n, TT);
        if ((size_t)ns * a.M * a.N * sizeof(float) > a_ws_bytes) ns = 1;          // no (or too small a) workspace: unsplit
        g.nsplit = ns;
        g.ws = ns > 1 ? a_ws : nullptr;
    }
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    // persistent grid: as many workgroups as stay resident (LDS: 160 KiB per CU), a multiple of 8 (XCDs), no more than tiles
    const int per_cu = (int)((160 * 1024) / lds) > 4 ? 4 : (int)((160 * 1024) / lds);
    long long nb = (long long)HSP_NUM_CU * per_cu;
    const long long tiles = (long long)g.tiles_m * g.tiles_n * g.nsplit;
    if (nb > tiles) nb = tiles;
    nb = (nb + 7) / 8 * 8;
    const dim3 grid((unsigned)nb), block(256);
#define GR_LAUNCH(L1, L2)                                                                                               \
    do {                                                                                                               \
        auto kern = gemm_rows_kernel<T, WM, WN, L1, L2, MODE>;                                                         \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                     \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g);                                                             \
        int rc_ = check_launch();                                                                                      \
        if (rc_ || g.nsplit == 1) return rc_;                                                                          \
        const long long tot_ = (long long)g.M * g.N;                                                                   \
        long long rg_ = (tot_ + 255) / 256;                                                                            \
        if (rg_ > HSP_NUM_CU * 8) rg_ = HSP_NUM_CU * 8;                                                                \
        hipLaunchKernelGGL(gemm_rows_reduce_kernel<T>, dim3((unsigned)rg_), dim3(256), 0, st, g);                      \
        return check_launch();                                                                                         \
    } while (0)
    // layouts on this path: X W ("nn"), x W^T ("nt"), X Wste^T + F Wa^T ("nt" + "nt"), g Wste + gfm W^T ("nn" + "nt")
    if constexpr (ES == 4) {
        if (lb1 == 1 && lb2 == 0) GR_LAUNCH(1, 0);
        if (lb1 == 2 && lb2 == 0) GR_LAUNCH(2, 0);
        if (lb1 == 1 && lb2 == 1) GR_LAUNCH(1, 1);
        if (lb1 == 2 && lb2 == 1) GR_LAUNCH(2, 1);
    } else {
        if (lb1 == 1 && lb2 == 0) GR_LAUNCH(1, 0);
        if (lb1 == 1 && lb2 == 1) GR_LAUNCH(1, 1);
    }
#undef GR_LAUNCH
    return HSP_ERR_UNSUPPORTED;
}

template <typename T, int WM, int WN>
static int launch_mode(const GemmRowsArgs& a, int lb1, int lb2, int mode, float* ws, size_t wsb, hipStream_t st) {
    if (mode == 1) return launch_cfg<T, WM, WN, 1>(a, lb1, lb2, ws, wsb, st);
    if constexpr (sizeof(T) == 4) {
        if (mode == 2) return launch_cfg<T, WM, WN, 2>(a, lb1, lb2, ws, wsb, st);
        return launch_cfg<T, WM, WN, 0>(a, lb1, lb2, ws, wsb, st);
    }
    return HSP_ERR_UNSUPPORTED;                    // bf16 operands: 16-byte aligned rows
}

// tile shape: the one whose tiles fill the 256 CUs most evenly (cost = tiles per CU, rounded up, x tile area; the small
// tile pays ~6 % for its extra operand traffic)
static bool prefer_small_tile(int M, int N, int TT = 1 << 30) {
    // short K (<= 4 k-blocks): the epilogue is a large share of a tile's life; the small tile keeps several workgroups per CU
    // to overlap it with the others' MFMAs (measured 66 vs 83 us fp32, 420 vs 538 us bf16 on the K = 128 layer products)
    if (TT <= 4 && (long long)((M + 63) / 64) * ((N + 63) / 64) >= HSP_NUM_CU) return true;
    auto cost = [&](int bm, int bn, double pen) {
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        return (double)((tiles + HSP_NUM_CU - 1) / HSP_NUM_CU) * bm * bn * pen;
    };
    return cost(64, 64, 1.06) < cost(128, 128, 1.0);
}

template <typename T>
static int gemm_rows_dispatch(const void* A1, int lda1, const void* B1, int ldb1, int l1, int K1, const void* A2, int lda2,
                              const void* B2, int ldb2, int l2, int K2, int M, int N, const float* bias, const void* resid,
                              int ldr, const float* cbias, int rpc, float alpha, const float* xyz3, const float* w3, void* C, int ldc,
                              int c_f32, void* ws, size_t ws_bytes, hspStream_t stream) {
    constexpr int ES = sizeof(T);
    if (!A1 || !B1 || !C || M <= 0 || N <= 0 || K1 <= 0 || lda1 < K1 || ldc < N) return HSP_ERR_BAD_ARG;
    if (l1 != 0 && l1 != 1) return HSP_ERR_BAD_ARG;
    if (ldb1 < (l1 == 0 ? K1 : N)) return HSP_ERR_BAD_ARG;
    const bool two = A2 != nullptr;
    if (two && (!B2 || K2 <= 0 || lda2 < K2 || (l2 != 0 && l2 != 1) || ldb2 < (l2 == 0 ? K2 : N))) return HSP_ERR_BAD_ARG;
    if (resid && ldr < N) return HSP_ERR_BAD_ARG;
    if (cbias && rpc <= 0) return HSP_ERR_BAD_ARG;
    if (ES == 2 && (l1 == 1 || (two && l2 == 1))) return HSP_ERR_UNSUPPORTED;     // bf16: "nt" operands only
    GemmRowsArgs g{};
    g.A1 = A1; g.B1 = B1; g.lda1 = lda1; g.ldb1 = ldb1; g.K1 = K1;
    int al = std::min(align_of(A1, lda1, ES), align_of(B1, ldb1, ES));
    if (two) {
        g.A2 = A2; g.B2 = B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
        al = std::min(al, std::min(align_of(A2, lda2, ES), align_of(B2, ldb2, ES)));
    }
    if (al < 16 && ES == 4) {                                  // 8- or 4-byte pieces
        const void* ps[4] = {A1, B1, A2, B2};
        for (const void* q : ps)
            if (reinterpret_cast<size_t>(q) % 4) return HSP_ERR_UNSUPPORTED;
    }
    // a dual-source call whose second source is a (K,N)-with-(N,K) pair is issued with the sources swapped
    // (the sum is commutative), which keeps the instantiated layout pairs to ("nt","nt") and ("nn","nt")
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias; g.resid = resid; g.ldr = ldr; g.cbias = cbias;
    g.rows_per_cloud = rpc > 0 ? rpc : 1;
    g.alpha = alpha;
    if ((xyz3 == nullptr) != (w3 == nullptr)) return HSP_ERR_BAD_ARG;
    g.xyz3 = xyz3; g.w3 = w3;
    g.c_f32 = c_f32;
    int lb1 = l1 == 0 ? 1 : 2, lb2 = two ? (l2 == 0 ? 1 : 2) : 0;
    if (two && lb1 == 1 && lb2 == 2) {
        const void* t; int ti;
        t = g.A1; g.A1 = g.A2; g.A2 = t;  t = g.B1; g.B1 = g.B2; g.B2 = t;
        ti = g.lda1; g.lda1 = g.lda2; g.lda2 = ti;  ti = g.ldb1; g.ldb1 = g.ldb2; g.ldb2 = ti;  ti = g.K1; g.K1 = g.K2; g.K2 = ti;
        lb1 = 2; lb2 = 1;
    }
    if (two && lb1 == 2 && lb2 == 2) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int TTall = (K1 + 128 / ES - 1) / (128 / ES) + (two ? (K2 + 128 / ES - 1) / (128 / ES) : 0);
    bool small = prefer_small_tile(M, N, TTall);
    const int mode = al == 16 ? 1 : al == 8 ? 2 : 0;           // staging load width
    g.nsplit = 1;
    if (small) return launch_mode<T, 1, 1>(g, lb1, lb2, mode, reinterpret_cast<float*>(ws), ws ? ws_bytes : 0, st);
    return launch_mode<T, 2, 2>(g, lb1, lb2, mode, reinterpret_cast<float*>(ws), ws ? ws_bytes : 0, st);
}

// ---- fp32 master parameters -> bf16 working copies, all tensors of a step in ONE launch --------------------------------
// entry e: src (rows, cols) fp32 with row pitch ld -> dst (rows, cols) bf16 and / or dstT (cols, rows) bf16 (the (N,K) form
// the bf16 GEMM wants of a (K,N) matrix).  32 x 32 tiles through LDS: both copies are written in 64-byte row segments.
__global__ __launch_bounds__(256) void cast_params_kernel(const HspCastDesc* __restrict__ tab, int n) {
    __shared__ float tile[32][33];
    int e = 0;
    while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].tile0) ++e;
    const HspCastDesc d = tab[e];
    const int t = (int)blockIdx.x - d.tile0;
    const int tcols = (d.cols + 31) >> 5;
    const int r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    bf16_t* dst = reinterpret_cast<bf16_t*>(d.dst);
    bf16_t* dstT = reinterpret_cast<bf16_t*>(d.dstT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ly + 8 * j, c = c0 + lx;
        float v = 0.f;
        if (r < d.rows && c < d.cols) {
            v = d.src[(size_t)r * d.ld + c];
            if (dst) dst[(size_t)r * d.cols + c] = (bf16_t)f32_to_bf16_bits(v);
        }
        tile[ly + 8 * j][lx] = v;
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ly + 8 * j, r = r0 + lx;
        if (r < d.rows && c < d.cols) dstT[(size_t)c * d.rows + r] = (bf16_t)f32_to_bf16_bits(tile[lx][ly + 8 * j]);
    }
}

}  // namespace hsp

using namespace hsp;

extern "C" int hsp_cast_params_bf16(const HspCastDesc* table_dev, int n, int total_tiles, hspStream_t stream) {
    if (!table_dev || n <= 0 || total_tiles <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_params_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), table_dev, n);
    return check_launch();
}

/* split-K workspace: 16 fp32 partial copies of C at most (0 = the shape never splits) */
extern "C" size_t hsp_gemm_rows_workspace_bytes(int M, int N, int K1, int K2, int elem_bytes) {
    if (M <= 0 || N <= 0 || K1 <= 0 || (elem_bytes != 2 && elem_bytes != 4)) return 0;
    const int bke = 128 / elem_bytes;
    const int TT = (K1 + bke - 1) / bke + (K2 > 0 ? (K2 + bke - 1) / bke : 0);
    const bool small = prefer_small_tile(M, N, TT);
    const int bm = small ? 64 : 128;
    const int ns = gemm_rows_pick_split((long long)((M + bm - 1) / bm) * ((N + bm - 1) / bm), TT);
    return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

extern "C" int hsp_gemm_rows_f32(const float* A1, int lda1, const float* B1, int ldb1, int b1_layout, int K1,
                                 const float* A2, int lda2, const float* B2, int ldb2, int b2_layout, int K2, int M, int N,
                                 const float* bias, const float* resid, int ldr, const float* cloud_bias,
                                 int rows_per_cloud, float alpha, const float* xyz3, const float* w3, float* C, int ldc,
                                 void* ws, size_t ws_bytes, hspStream_t stream) {
    return gemm_rows_dispatch<float>(A1, lda1, B1, ldb1, b1_layout, K1, A2, lda2, B2, ldb2, b2_layout, K2, M, N, bias, resid,
                                     ldr, cloud_bias, rows_per_cloud, alpha, xyz3, w3, C, ldc, 1, ws, ws_bytes, stream);
}

extern "C" int hsp_gemm_rows_bf16(const hsp_bf16_t* A1, int lda1, const hsp_bf16_t* B1, int ldb1, int K1,
                                  const hsp_bf16_t* A2, int lda2, const hsp_bf16_t* B2, int ldb2, int K2, int M, int N,
                                  const float* bias, const hsp_bf16_t* resid, int ldr, const float* cloud_bias,
                                  int rows_per_cloud, float alpha, const float* xyz3, const float* w3, void* C, int ldc,
                                  int c_is_f32, void* ws, size_t ws_bytes, hspStream_t stream) {
    return gemm_rows_dispatch<unsigned short>(A1, lda1, B1, ldb1, 0, K1, A2, lda2, B2, ldb2, 0, K2, M, N, bias, resid, ldr,
                                              cloud_bias, rows_per_cloud, alpha, xyz3, w3, C, ldc, c_is_f32 ? 1 : 0, ws, ws_bytes,
                                              stream);
}
