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
// Structure: a 256-thread workgroup (4 waves as 2x2) owns a BM x BN tile of C (128x128 or 64x64: the row counts of this
// path are 257 * 2^j, so tile shape is chosen per call by how evenly the tiles fill 256 CUs); K is walked in blocks of
// 128 BYTES per row (32 fp32 / 64 bf16) staged through LDS: global -> registers (in flight under the MFMAs of the
// previous block) -> LDS, double-buffered, one barrier per block.  The LDS image is the same for both types: rows of
// 128 data bytes + 16 pad bytes (144-byte pitch: the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots),
// read at row*144 + 32*step + 16*(lane>>5), which hands every lane
//   fp32: 4 consecutive k  -> 4 x v_mfma_f32_32x32x2_f32  (k pairs (j, j+4): the k order inside a block is permuted
//                              identically for A and B, the sum is the same set of products)
//   bf16: 8 consecutive k  -> 1 x v_mfma_f32_32x32x16_bf16
// "nn" operands (k-major rows, fp32 only) are staged as [k][n] and read with ds_read_b32 (lanes = consecutive n).
// Rows past M / columns past N are clamped on load and masked on store; k past K is zero-filled in the loader, so any
// K (3, 771, 1286, 1289 ...) and any operand alignment (runtime vector width 16/8/4 bytes) is accepted.
#include "common.h"

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct GemmRowsArgs {
    const void* A1; const void* B1; const void* A2; const void* B2;
    int lda1, ldb1, K1, va1, vb1;          // v*: bytes per staging load the operand's alignment allows (16, 8, 4, 2)
    int lda2, ldb2, K2, va2, vb2;
    void* C; int ldc;
    int M, N;
    const float* bias;                     // (N) fp32 or null
    const void* resid; int ldr;            // (M,N) of the output type or null
    const float* cbias; int rows_per_cloud;  // (ceil(M / rows_per_cloud), N) fp32 or null
    int tiles_m, tiles_n;
};

#define GR_PITCH 144                       // LDS bytes per tile row (128 data + 16 pad)

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {              // round to nearest even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 16 bytes of row `p` starting at element k0 (EPC = 16 / ES elements), elements at or past K zero-filled.
// vb = bytes per load the operand's alignment allows (host: base pointer and row pitch are multiples of it).  A full
// chunk is fetched in 16 / vb pieces; the ragged chunk at the end of K element by element, so nothing past element
// K-1 is ever touched unless vb == 16 (then the pitch is a multiple of 16 bytes and the chunk lies inside the row).
template <int ES>   // element size in bytes (4: fp32, 2: bf16)
__device__ __forceinline__ uint4 load_chunk_guarded(const char* __restrict__ p, int k0, int K, int vb) {
    constexpr int EPC = 16 / ES;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (k0 >= K) return make_uint4(0u, 0u, 0u, 0u);
    const char* q = p + (size_t)k0 * ES;
    const bool full = k0 + EPC <= K;
    if (vb == 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(q);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (!full) {
            const int keep = K - k0;                            // 1 .. EPC-1 elements survive
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ES == 4) w[e] = e < keep ? w[e] : 0u;
                else w[e] &= (2 * e < keep ? 0x0000ffffu : 0u) | (2 * e + 1 < keep ? 0xffff0000u : 0u);
            }
        }
    } else if (full && vb == 8) {
        const uint2 a = *reinterpret_cast<const uint2*>(q);
        const uint2 b = *reinterpret_cast<const uint2*>(q + 8);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
    } else if (full && (vb == 4 || ES == 4)) {
        const unsigned* u = reinterpret_cast<const unsigned*>(q);
        w[0] = u[0]; w[1] = u[1]; w[2] = u[2]; w[3] = u[3];
    } else if (ES == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k0 + e < K) w[e] = *reinterpret_cast<const unsigned*>(q + 4 * e);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < K) w[e >> 1] |= (unsigned)*reinterpret_cast<const unsigned short*>(q + 2 * e) << (16 * (e & 1));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// T: float or unsigned short (bf16 bits).  WM, WN: 32x32 MFMA tiles per wave along M / N (block tile = 64*WM x 64*WN).
// LB1 / LB2: layout of B1 / B2 -- 1 "nt" (N,K) k contiguous, 2 "nn" (K,N) n contiguous (fp32 only), 0 (LB2) = no 2nd source
template <typename T, int WM, int WN, int LB1, int LB2>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const GemmRowsArgs g) {
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;                 // elements per 16-byte chunk
    constexpr int BKE = 128 / ES;                // k elements per block
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_BYTES = BM * GR_PITCH;
    constexpr int NN_PITCH = (BN + 4) * 4;       // "nn" tile: 32 k-rows of BN fp32 + pad
    constexpr int B_BYTES = (BN * GR_PITCH > 32 * NN_PITCH) ? BN * GR_PITCH : 32 * NN_PITCH;
    constexpr int STAGE = A_BYTES + B_BYTES;
    static_assert(LB1 == 1 || (LB1 == 2 && ES == 4), "nn operands are fp32 only");
    static_assert(LB2 == 0 || LB2 == 1 || (LB2 == 2 && ES == 4), "nn operands are fp32 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int li = lane & 31, lh = lane >> 5;

    // tile of this workgroup: XCD-aware (block b runs on XCD b % 8): each XCD walks a contiguous range of tiles, tn
    // fastest, so the workgroups sharing an A row panel share one L2 (bijective for any tile count)
    const int ntiles = g.tiles_m * g.tiles_n;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int T1 = (g.K1 + BKE - 1) / BKE;
    const int T2 = LB2 ? (g.K2 + BKE - 1) / BKE : 0;
    const int TT = T1 + T2;

    uint4 ra[BM / 32], rb[BN / 32];
    // staging roles: A (and "nt" B): thread -> 16-byte chunk (tid & 7) of rows (tid >> 3) + 32 p
    const int s_row = tid >> 3, s_chunk = tid & 7;

    auto fetch = [&](int t) {
        const bool second = LB2 && t >= T1;
        const char* A = reinterpret_cast<const char*>(second ? g.A2 : g.A1);
        const char* B = reinterpret_cast<const char*>(second ? g.B2 : g.B1);
        const int lda = second ? g.lda2 : g.lda1, ldb = second ? g.ldb2 : g.ldb1, K = second ? g.K2 : g.K1;
        const int va = second ? g.va2 : g.va1, vb = second ? g.vb2 : g.vb1;
        const int kb = (second ? t - T1 : t) * BKE;
        const int lb = (LB2 == 0 || LB1 == LB2) ? LB1 : (second ? LB2 : LB1);
#pragma unroll
        for (int p = 0; p < BM / 32; ++p) {
            const int row = min(m0 + s_row + 32 * p, g.M - 1);
            ra[p] = load_chunk_guarded<ES>(A + (size_t)row * lda * ES, kb + s_chunk * EPC, K, va);
        }
        if (lb == 1) {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                const int col = min(n0 + s_row + 32 * p, g.N - 1);
                rb[p] = load_chunk_guarded<ES>(B + (size_t)col * ldb * ES, kb + s_chunk * EPC, K, vb);
            }
        } else {
            // "nn": chunk q = tid + 256 p of the 32 x BN tile: k row q / (BN/4), 4 columns at 4 * (q % (BN/4))
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                const int q = tid + 256 * p;
                const int kr = q / (BN / 4), nc = (q - kr * (BN / 4)) * 4;
                const int k = kb + kr;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (k < K) v = load_chunk_guarded<ES>(B + (size_t)k * ldb * ES, n0 + nc, g.N, vb);
                rb[p] = v;
            }
        }
    };
    auto stash = [&](int t) {
        char* sa = smem + (t & 1) * STAGE;
        char* sb = sa + A_BYTES;
        const bool second = LB2 && t >= T1;
        const int lb = (LB2 == 0 || LB1 == LB2) ? LB1 : (second ? LB2 : LB1);
#pragma unroll
        for (int p = 0; p < BM / 32; ++p)
            *reinterpret_cast<uint4*>(sa + (s_row + 32 * p) * GR_PITCH + s_chunk * 16) = ra[p];
        if (lb == 1) {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p)
                *reinterpret_cast<uint4*>(sb + (s_row + 32 * p) * GR_PITCH + s_chunk * 16) = rb[p];
        } else {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                const int q = tid + 256 * p;
                const int kr = q / (BN / 4), nc = (q - kr * (BN / 4)) * 4;
                *reinterpret_cast<uint4*>(sb + kr * NN_PITCH + nc * 4) = rb[p];
            }
        }
    };
    auto compute = [&](int t) {
        const char* sa = smem + (t & 1) * STAGE;
        const char* sb = sa + A_BYTES;
        const bool second = LB2 && t >= T1;
        const int lb = (LB2 == 0 || LB1 == LB2) ? LB1 : (second ? LB2 : LB1);
        const char* pa = sa + (wm0 + li) * GR_PITCH + 16 * lh;
        if (lb == 1) {
            const char* pb = sb + (wn0 + li) * GR_PITCH + 16 * lh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 a[WM], b[WN];
#pragma unroll
                for (int x = 0; x < WM; ++x) a[x] = *reinterpret_cast<const uint4*>(pa + x * 32 * GR_PITCH + 32 * s);
#pragma unroll
                for (int y = 0; y < WN; ++y) b[y] = *reinterpret_cast<const uint4*>(pb + y * 32 * GR_PITCH + 32 * s);
                if (ES == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int x = 0; x < WM; ++x)
#pragma unroll
                            for (int y = 0; y < WN; ++y) {
                                const unsigned au = j == 0 ? a[x].x : j == 1 ? a[x].y : j == 2 ? a[x].z : a[x].w;
                                const unsigned bu = j == 0 ? b[y].x : j == 1 ? b[y].y : j == 2 ? b[y].z : b[y].w;
                                acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au), __uint_as_float(bu),
                                                                                 acc[x][y], 0, 0, 0);
                            }
                } else {
#pragma unroll
                    for (int x = 0; x < WM; ++x)
#pragma unroll
                        for (int y = 0; y < WN; ++y)
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[x]),
                                                                                __builtin_bit_cast(bf16x8, b[y]), acc[x][y],
                                                                                0, 0, 0);
                }
            }
        } else if (ES == 4) {
            // "nn" B tile [k][n]: lane reads B[k = 8 s + 4 lh + j][wn0 + 32 y + li]
            const char* pb = sb + (4 * lh) * NN_PITCH + (wn0 + li) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 a[WM];
                float b[WN][4];
#pragma unroll
                for (int x = 0; x < WM; ++x) a[x] = *reinterpret_cast<const uint4*>(pa + x * 32 * GR_PITCH + 32 * s);
#pragma unroll
                for (int y = 0; y < WN; ++y)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        b[y][j] = *reinterpret_cast<const float*>(pb + (8 * s + j) * NN_PITCH + y * 128);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int x = 0; x < WM; ++x)
#pragma unroll
                        for (int y = 0; y < WN; ++y) {
                            const unsigned au = j == 0 ? a[x].x : j == 1 ? a[x].y : j == 2 ? a[x].z : a[x].w;
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au), b[y][j], acc[x][y], 0, 0, 0);
                        }
            }
        }
    };

    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < TT; ++t) {
        if (t + 1 < TT) fetch(t + 1);            // global loads in flight under this block's MFMAs
        __builtin_amdgcn_sched_barrier(0);       // (hipcc would sink the loads next to their use in stash())
        compute(t);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < TT) stash(t + 1);            // the other buffer: last read in iteration t-1, before the barrier below
        __syncthreads();
    }

    // epilogue: accumulator r of tile (x,y) <-> row wm0 + 32x + (r&3) + 8(r>>2) + 4 lh, column wn0 + 32y + li
    T* C = reinterpret_cast<T*>(g.C);
    const T* R = reinterpret_cast<const T*>(g.resid);
    // per-cloud bias: rows of this tile lie in clouds c0, c0+1, ... ; boundaries by comparison (no per-row division)
    int c0 = 0, nb = 0x7fffffff;
    const int rpc = g.rows_per_cloud;
    if (g.cbias) { c0 = m0 / rpc; nb = (c0 + 1) * rpc; }
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        const int col = n0 + wn0 + 32 * y + li;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int x = 0; x < WM; ++x) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= g.M) continue;
                float v = acc[x][y][r] + bv;
                if (g.resid) {
                    if (ES == 4) v += reinterpret_cast<const float*>(R)[(size_t)row * g.ldr + col];
                    else v += bf16_to_f32(reinterpret_cast<const unsigned short*>(R)[(size_t)row * g.ldr + col]);
                }
                if (g.cbias) {
                    int c = c0;
                    if (row >= nb) c = c0 + 1 + (row - nb) / rpc;      // rare (tile spans clouds): the division is off the hot path
                    v += g.cbias[(size_t)c * g.N + col];
                }
                if (ES == 4) reinterpret_cast<float*>(C)[(size_t)row * g.ldc + col] = v;
                else reinterpret_cast<unsigned short*>(C)[(size_t)row * g.ldc + col] = f32_to_bf16(v);
            }
        }
    }
}

// bytes per staging load an operand allows: base pointer, row pitch (bytes) and 16 all share the factor
static int vec_bytes(const void* p, int ld_elems, int es) {
    const size_t a = reinterpret_cast<size_t>(p);
    const size_t pitch = (size_t)ld_elems * es;
    for (int v = 16; v > es; v >>= 1)
        if (a % v == 0 && pitch % v == 0) return v;
    return es;
}

template <typename T, int WM, int WN>
static int launch_cfg(const GemmRowsArgs& a, int lb1, int lb2, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int ES = sizeof(T);
    GemmRowsArgs g = a;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = (a.N + BN - 1) / BN;
    constexpr size_t nn = (size_t)32 * (BN + 4) * 4, nt = (size_t)BN * GR_PITCH;
    const size_t lds = 2 * ((size_t)BM * GR_PITCH + (nn > nt ? nn : nt));
    const dim3 grid(g.tiles_m * g.tiles_n), block(256);
#define GR_LAUNCH(L1, L2)                                                                                               \
    do {                                                                                                               \
        auto kern = gemm_rows_kernel<T, WM, WN, L1, L2>;                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                     \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, grid, block, lds, st, g);                                                             \
        return check_launch();                                                                                         \
    } while (0)
    if constexpr (ES == 4) {
        if (lb1 == 1 && lb2 == 0) GR_LAUNCH(1, 0);
        if (lb1 == 2 && lb2 == 0) GR_LAUNCH(2, 0);
        if (lb1 == 1 && lb2 == 1) GR_LAUNCH(1, 1);
        if (lb1 == 1 && lb2 == 2) GR_LAUNCH(1, 2);
        if (lb1 == 2 && lb2 == 1) GR_LAUNCH(2, 1);
        if (lb1 == 2 && lb2 == 2) GR_LAUNCH(2, 2);
    } else {
        if (lb1 == 1 && lb2 == 0) GR_LAUNCH(1, 0);
        if (lb1 == 1 && lb2 == 1) GR_LAUNCH(1, 1);
    }
#undef GR_LAUNCH
    return HSP_ERR_UNSUPPORTED;
}

// tile shape: the one whose tiles fill the 256 CUs most evenly (cost = tiles per CU, rounded up, x tile area; the small
// tile pays ~6 % for its extra operand traffic)
static bool prefer_small_tile(int M, int N) {
    auto cost = [&](int bm, int bn, double pen) {
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        return (double)((tiles + HSP_NUM_CU - 1) / HSP_NUM_CU) * bm * bn * pen;
    };
    return cost(64, 64, 1.06) < cost(128, 128, 1.0);
}

template <typename T>
static int gemm_rows_dispatch(const void* A1, int lda1, const void* B1, int ldb1, int l1, int K1, const void* A2, int lda2,
                              const void* B2, int ldb2, int l2, int K2, int M, int N, const float* bias, const void* resid,
                              int ldr, const float* cbias, int rpc, void* C, int ldc, hspStream_t stream) {
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
    g.va1 = vec_bytes(A1, lda1, ES); g.vb1 = vec_bytes(B1, ldb1, ES);
    if (two) {
        g.A2 = A2; g.B2 = B2; g.lda2 = lda2; g.ldb2 = ldb2; g.K2 = K2;
        g.va2 = vec_bytes(A2, lda2, ES); g.vb2 = vec_bytes(B2, ldb2, ES);
    }
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.bias = bias; g.resid = resid; g.ldr = ldr; g.cbias = cbias;
    g.rows_per_cloud = rpc > 0 ? rpc : 1;
    const int lb1 = l1 == 0 ? 1 : 2, lb2 = two ? (l2 == 0 ? 1 : 2) : 0;
    hipStream_t st = as_stream(stream);
    if (prefer_small_tile(M, N)) return launch_cfg<T, 1, 1>(g, lb1, lb2, st);
    return launch_cfg<T, 2, 2>(g, lb1, lb2, st);
}

}  // namespace hsp

using namespace hsp;

extern "C" int hsp_gemm_rows_f32(const float* A1, int lda1, const float* B1, int ldb1, int b1_layout, int K1,
                                 const float* A2, int lda2, const float* B2, int ldb2, int b2_layout, int K2, int M, int N,
                                 const float* bias, const float* resid, int ldr, const float* cloud_bias,
                                 int rows_per_cloud, float* C, int ldc, hspStream_t stream) {
    return gemm_rows_dispatch<float>(A1, lda1, B1, ldb1, b1_layout, K1, A2, lda2, B2, ldb2, b2_layout, K2, M, N, bias, resid,
                                     ldr, cloud_bias, rows_per_cloud, C, ldc, stream);
}

extern "C" int hsp_gemm_rows_bf16(const hsp_bf16_t* A1, int lda1, const hsp_bf16_t* B1, int ldb1, int K1,
                                  const hsp_bf16_t* A2, int lda2, const hsp_bf16_t* B2, int ldb2, int K2, int M, int N,
                                  const float* bias, const hsp_bf16_t* resid, int ldr, const float* cloud_bias,
                                  int rows_per_cloud, hsp_bf16_t* C, int ldc, hspStream_t stream) {
    return gemm_rows_dispatch<unsigned short>(A1, lda1, B1, ldb1, 0, K1, A2, lda2, B2, ldb2, 0, K2, M, N, bias, resid, ldr,
                                              cloud_bias, rows_per_cloud, C, ldc, stream);
}
