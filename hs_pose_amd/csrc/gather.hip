// gather.hip -- neighbourhood max-pool and row gather/scatter kernels for gfx950.
//
// gather_max: replaces indexing_neighbor_new(...) + torch.max(dim=2) of the ORL branch and of
//             Pool_layer (reference network/fs_net_repo/gcn3d.py:214-216, :236-240); with `qsel`
//             only the rows Pool_layer keeps after its randperm (gcn3d.py:243-245) are computed.
// gather_rows: replaces the nearest-neighbour up-sampling gathers (FaceRecon.py:102-104) and
//             vertices[:, sample_idx] (gcn3d.py:244); can write directly into a column slice of
//             the concatenated feature tensor (FaceRecon.py:107).
// All are pure HBM/L2 streaming kernels: one lane per float4 of a row, 16-byte accesses.
#include "common.h"
#include <type_traits>

namespace hsp {

// running max / first arg-max over the k neighbour rows nb[0..k) of a float4 column group.  8 neighbours per
// pass: their rows are gathered together while the next 8 indices are already in flight (index -> row is a
// dependent pair of L2 round trips; one neighbour at a time, unrolled by 4, cost 10 of them per point at k = 20)
template <typename FT>
__device__ __forceinline__ void gather_max_rows(const FT* __restrict__ fb, const int32_t* __restrict__ nb, int k,
                                                int C, float4& best, int& a0, int& a1, int& a2, int& a3) {
    constexpr int NB = 8;
    int cur[NB], nxt[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) cur[u] = nb[min(u, k - 1)];
    for (int n0 = 0; n0 < k; n0 += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) nxt[u] = nb[min(n0 + NB + u, k - 1)];
        float4 f[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) f[u] = Feat<FT>::ld4(fb + (size_t)cur[u] * C);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int n = n0 + u;
            if (n < k) {                                       // ascending n: the first maximum wins, as torch.max
                if (f[u].x > best.x) { best.x = f[u].x; a0 = n; }
                if (f[u].y > best.y) { best.y = f[u].y; a1 = n; }
                if (f[u].z > best.z) { best.z = f[u].z; a2 = n; }
                if (f[u].w > best.w) { best.w = f[u].w; a3 = n; }
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
    }
}


// one thread per (query row, float4 column group); grid-stride
template <typename FT>
__global__ __launch_bounds__(256) void gather_max_fwd_kernel(const FT* __restrict__ feat,
                                                             const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ qsel, int B, int Nsrc,
                                                             int Nidx, int Nq, int k, int kstride, int C,
                                                             FT* __restrict__ out,
                                                             uint8_t* __restrict__ argmax,
                                                             const float* __restrict__ xyz = nullptr,
                                                             float* __restrict__ xyz_sel = nullptr) {
    const int cq = C >> 2;
    const long long total = (long long)B * Nq * cq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const long long row = e / cq;            // b*Nq + q
        const int q = (int)(row % Nq);
        const int b = (int)(row / Nq);
        const int qi = qsel ? qsel[q] : q;
        if (xyz_sel && g < 3)                    // Pool_layer: the kept points' coordinates ride along (vertices[:, sample_idx], gcn3d.py:244)
            xyz_sel[row * 3 + g] = xyz[((size_t)b * Nidx + qi) * 3 + g];
        const int32_t* nb = idx + ((size_t)b * Nidx + qi) * kstride;
        const FT* fb = feat + (size_t)b * Nsrc * C + (g << 2);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        gather_max_rows<FT>(fb, nb, k, C, best, a0, a1, a2, a3);
        Feat<FT>::st4(out + row * C + (g << 2), best);
        *reinterpret_cast<uchar4*>(argmax + row * C + (g << 2)) =
            make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
    }
}

// ORL global feature in one pass (reference get_ORL_global, gcn3d.py:211-218, before the repeat):
//   fg[b,c] = (1/N) sum_i max_{n<k} feat[b, idx[b,i,n], c],   argmax (B,N,C) uint8 = winning slot
// One workgroup per (cloud, chunk of ORL_ROWS points): a lane owns 4 channels of a point, takes the max
// over the k gathered rows, the chunk's column sums are folded through LDS in a fixed order and written
// to part[b][chunk][C]; orl_finalize_kernel folds the chunks (deterministic).  The (B,N,C) max tensor of
// the reference is never written.
// (Round 3 also built last-arriving-workgroup folds -- a ticket per cloud, write-through partials, the last workgroup folds -- to
// save the ~4.5 us fold launches: bit-identical, +20 us per step fence-free and +60 us with agent-scope fences: the in-kernel fold
// is the same dependent-load chain run by one workgroup per cloud at the kernel's tail.  Removed in round 4.)
#define ORL_ROWS 128         // upper bound of points per chunk
template <typename FT>
__global__ __launch_bounds__(256) void orl_partial_kernel(const FT* __restrict__ feat,
                                                          const int32_t* __restrict__ idx, int N, int k,
                                                          int kstride, int C, uint8_t* __restrict__ argmax,
                                                          float* __restrict__ part, int nchunk, int rows) {
    __shared__ float4 red[256];
    const int cq = C >> 2;
    const int tid = threadIdx.x;
    const int g = tid % cq, rl = tid / cq, RL = 256 / cq;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows, r1 = min(N, r0 + rows);
    const FT* fb = feat + (size_t)b * N * C + (g << 2);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = r0 + rl; i < r1; i += RL) {
        const int32_t* nb = idx + ((size_t)b * N + i) * kstride;
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        gather_max_rows<FT>(fb, nb, k, C, best, a0, a1, a2, a3);
        *reinterpret_cast<uchar4*>(argmax + ((size_t)b * N + i) * C + (g << 2)) =
            make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
        s.x += best.x; s.y += best.y; s.z += best.z; s.w += best.w;
    }
    red[tid] = s;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; ++l) { const float4 v = red[l * cq + g]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(part + ((size_t)b * nchunk + chunk) * C + (g << 2)) = s;
    }
}

// The same global feature with the cloud's column slab resident in LDS (round 5).  orl_partial_kernel gathers k rows of 512+ bytes per
// point through the texture path: 168 MB of L2 requests per N = 1028, C = 128 call, which is what its 17 us are (VALU 13 % busy).
// Here a workgroup owns (cloud, 8 channels): the slab feat[b][:, c0:c0+8] is read ONCE into LDS (48-byte rows: 12-float pitch, so
// that random rows spread over the banks), a thread takes (point, 4 channels), walks the point's k neighbours with ds_read_b128,
// writes the winning slots and keeps the column sums; the workgroup folds them in a fixed order and writes fg itself -- no partial
// buffer, no fold launch.  k = 20, contiguous lists, the slab within 144 KB of LDS (fp32: N <= 2800; bf16 rows stay raw, 24-byte pitch:
// N <= 5600); anything else keeps the chunked form.
//
// The kernel is bound by VALU ISSUE, not by memory: 1028 x 128 x 20 compare / select triples over 1024 SIMDs are ~3.5 us, and one
// workgroup per CU means every other instruction of the pass is on the critical path (a first form with an integer division per staged
// index ran 26 us with every load and store removed).  Hence: lists staged as ready-made LDS byte offsets by a flat coalesced copy
// (contiguous lists only), 512 threads (two waves per SIMD cover each other's LDS latency), max first and the winning slot by a
// descending equality scan (2.5 instead of 3 instructions per element).
#ifdef HSP_ORL_PROF
static __device__ long long* g_orl_prof = nullptr;     // tools/prof_orl_tile.py: clock64 stamps of workgroup 0, wave 0
#define ORL_STAMP(slot) do { if (g_orl_prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_orl_prof[slot] = clock64(); } while (0)
#else
#define ORL_STAMP(slot) do { } while (0)
#endif
#define ORLT_TC 8
#define ORLT_PITCH4 3            // float4 units per LDS row
#define ORLT_WG 512
// max(a, b, c) of plain numbers in ONE instruction (fmaxf compiles to three: it quiets NaNs first).  NaN activations: v_max3_f32
// returns the non-NaN operand, so this kernel MASKS a NaN neighbour (the maximum of the others wins; a NaN-only column takes slot
// 0), where the chunked orl_partial_kernel (compares with >) and torch.max keep a NaN that sits in the first slot.  Neither form
// orders NaNs; a NaN activation is an upstream fault (engine/train.py:91-95 skips such a step on its NaN loss), not a case the
// kernels promise to agree on.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// K is a template constant (the HS layers' k = 20 is the only instance): every neighbour loop is straight-line code -- run-time "n < k"
// guards turned the first form into 180 branches -- and all K rows of a (point, 4 channels) sit in registers (80 of the 128 VGPRs a
// 512-thread workgroup has)
template <typename FT, int K>
__global__ __launch_bounds__(ORLT_WG) void orl_tile_kernel(const FT* __restrict__ feat, const int32_t* __restrict__ idx, int B, int N,
                                                           int C, uint8_t* __restrict__ argmax, float* __restrict__ fg,
                                                           float inv_n) {
    static_assert(K % 2 == 0 && K >= 4, "a wave's 32 K staged indices are K / 2 per lane");
    constexpr int k = K;
    extern __shared__ __attribute__((aligned(16))) float4 orl_tile[];      // N slab rows, then ORLT_WG / 64 strips of 32 k ushorts
    __shared__ float4 red[2 * (ORLT_WG / 64)];
    // workgroups are dealt to the 8 XCDs round-robin: all column tiles of a cloud go to ONE XCD (cloud = xcd + 8 j), so that the
    // 32-byte slices of a feature line and the 8-byte slices of an argmax line meet in one L2 instead of eight
    const int T = C / ORLT_TC, L = blockIdx.x;
    int b, c0;
    if ((B & 7) == 0) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / T); c0 = (slot % T) * ORLT_TC; }
    else { b = L / T; c0 = (L % T) * ORLT_TC; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FT* fb = feat + (size_t)b * N * C + c0;
    ORL_STAMP(0);
    // slab rows: fp32 -> 8 floats on a 48-byte pitch; bf16 -> the 16 raw bytes on a 24-byte pitch (N = 4096 fits in 96 KB)
    constexpr int ROWB = sizeof(FT) == 4 ? 48 : 24, SH = sizeof(FT) == 4 ? 4 : 3;     // row offset = 3 idx << SH
    char* const slab0 = reinterpret_cast<char*>(orl_tile);
    for (int i = tid; i < N; i += ORLT_WG) {
        if constexpr (sizeof(FT) == 4) {
            *reinterpret_cast<float4*>(slab0 + i * ROWB) = Feat<FT>::ld4(fb + (size_t)i * C);
            *reinterpret_cast<float4*>(slab0 + i * ROWB + 16) = Feat<FT>::ld4(fb + (size_t)i * C + 4);
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(fb + (size_t)i * C);
            *reinterpret_cast<uint2*>(slab0 + i * ROWB) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(slab0 + i * ROWB + 8) = make_uint2(v.z, v.w);
        }
    }
    const int q = lane & 1, pw = lane >> 1;                    // a wave: 32 points x two 4-channel groups
    constexpr int EPL = K / 2, PASS = 32 * (ORLT_WG / 64);
    // Neighbour lists: the wave's 32 points of a pass own 32 k CONSECUTIVE indices; the wave copies them coalesced (k / 2 dwords per
    // lane), one pass AHEAD, into its own LDS strip -- as byte offsets of the slab rows.
    unsigned short* strip = reinterpret_cast<unsigned short*>(slab0 + (size_t)N * ROWB) + wave * (32 * k);   // entries 3 idx (< 65536 up to N = 21845)
    const int32_t* ib = idx + (size_t)b * N * k;
    const int last = N * k - 1;
    int nxt[EPL];
    auto fetch = [&](int i0) {                                 // lists of points i0 + 32 wave ... + 32 (past the cloud: any entry)
        const int e0 = (i0 + 32 * wave) * k + lane;
#pragma unroll
        for (int u = 0; u < EPL; ++u) nxt[u] = ib[min(e0 + 64 * u, last)];
    };
    fetch(0);
    __syncthreads();                                           // slab complete
    ORL_STAMP(1);
    const char* slab = slab0 + q * (ROWB / 3);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0; i0 < N; i0 += PASS) {
        __builtin_amdgcn_wave_barrier();                       // (the strip is the wave's own: program order is the only hazard)
#pragma unroll
        for (int u = 0; u < EPL; ++u) strip[lane + 64 * u] = (unsigned short)(nxt[u] * 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ORL_STAMP(2 + 3 * (i0 / PASS));
        if (i0 + PASS < N) fetch(i0 + PASS);
        const int i = i0 + 32 * wave + pw;
        if (i < N) {
            const unsigned short* my = strip + pw * k;
            float4 f[K];
#pragma unroll
            for (int n = 0; n < K; ++n) {
                const char* rp = slab + ((unsigned)my[n] << SH);
                if constexpr (sizeof(FT) == 4) f[n] = *reinterpret_cast<const float4*>(rp);
                else {
                    const uint2 u2 = *reinterpret_cast<const uint2*>(rp);
                    f[n] = make_float4(__uint_as_float(u2.x << 16), __uint_as_float(u2.x & 0xffff0000u), __uint_as_float(u2.y << 16),
                                       __uint_as_float(u2.y & 0xffff0000u));
                }
            }
            ORL_STAMP(3 + 3 * (i0 / PASS));
            float4 best = f[0];
#pragma unroll
            for (int n = 1; n + 1 < K; n += 2) {
                best.x = max3_raw(best.x, f[n].x, f[n + 1].x); best.y = max3_raw(best.y, f[n].y, f[n + 1].y);
                best.z = max3_raw(best.z, f[n].z, f[n + 1].z); best.w = max3_raw(best.w, f[n].w, f[n + 1].w);
            }
            if ((K & 1) == 0) {
                best.x = max3_raw(best.x, f[K - 1].x, f[K - 1].x); best.y = max3_raw(best.y, f[K - 1].y, f[K - 1].y);
                best.z = max3_raw(best.z, f[K - 1].z, f[K - 1].z); best.w = max3_raw(best.w, f[K - 1].w, f[K - 1].w);
            }
            int a0 = 0, a1 = 0, a2 = 0, a3 = 0;               // descending scan: the FIRST slot holding the maximum, as torch.max
#pragma unroll
            for (int n = K - 1; n >= 1; --n) {
                a0 = f[n].x == best.x ? n : a0; a1 = f[n].y == best.y ? n : a1;
                a2 = f[n].z == best.z ? n : a2; a3 = f[n].w == best.w ? n : a3;
            }
            a0 = f[0].x == best.x ? 0 : a0; a1 = f[0].y == best.y ? 0 : a1; a2 = f[0].z == best.z ? 0 : a2; a3 = f[0].w == best.w ? 0 : a3;
            *reinterpret_cast<uchar4*>(argmax + ((size_t)b * N + i) * C + c0 + (q << 2)) =
                make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
            s.x += best.x; s.y += best.y; s.z += best.z; s.w += best.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ORL_STAMP(4 + 3 * (i0 / PASS));
    }
    ORL_STAMP(30);
    // fixed-order fold: within the wave over the 32 point slots of each parity (xor 2 ... 32), then the waves in order
#pragma unroll
    for (int d = 2; d < 64; d <<= 1) {
        s.x += __shfl_xor(s.x, d); s.y += __shfl_xor(s.y, d); s.z += __shfl_xor(s.z, d); s.w += __shfl_xor(s.w, d);
    }
    if (lane < 2) red[wave * 2 + q] = s;
    __syncthreads();
    if (tid < 2) {
        float4 t = red[q];
        for (int w = 1; w < ORLT_WG / 64; ++w) { const float4 v = red[w * 2 + q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(fg + (size_t)b * C + c0 + (q << 2)) = make_float4(t.x * inv_n, t.y * inv_n, t.z * inv_n, t.w * inv_n);
    }
    ORL_STAMP(31);
}

// out[b][c] = scale * sum_chunk part[b][chunk][c]   (also the generic "column sum per cloud" second stage)
__global__ __launch_bounds__(256) void chunk_fold_kernel(const float* __restrict__ part, int B, int nchunk, int C,
                                                         float scale, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * C) return;
    const int b = e / C, c = e - b * C;
    const float* p = part + (size_t)b * nchunk * C + c;
    // 8 independent partial sums: 8 loads in flight per round trip (the fold is a latency chain, not bandwidth)
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int ch = 0;
    for (; ch + 7 < nchunk; ch += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += p[(size_t)(ch + u) * C];
    }
    for (; ch < nchunk; ++ch) s[ch & 7] += p[(size_t)ch * C];
    out[e] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) * scale;
}

// part[b][chunk][c] = sum of x[b][i][c] over the chunk's rows (first stage of a per-cloud column sum)
template <typename FT>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const FT* __restrict__ x, int N, int C,
                                                             float* __restrict__ part, int nchunk, int rows) {
    __shared__ float4 red[256];
    const int cq = C >> 2;
    const int tid = threadIdx.x;
    const int g = tid % cq, rl = tid / cq, RL = 256 / cq;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows, r1 = min(N, r0 + rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = r0 + rl; i < r1; i += RL) {
        const float4 v = Feat<FT>::ld4(x + ((size_t)b * N + i) * C + (g << 2));
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[tid] = s;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; ++l) { const float4 v = red[l * cq + g]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(part + ((size_t)b * nchunk + chunk) * C + (g << 2)) = s;
    }
}

__device__ __forceinline__ float fma_plain(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// colsum + the three coordinate moments of the rows in one pass (HSlayer_surface backward: gt = sum_i g and the STE weight
// gradient g^T xyz, gcn3d.py:85): part[b][chunk][slot][c], slot 0 = sum g, slot 1..3 = sum g * (x, y, z)
template <typename FT>
__global__ __launch_bounds__(256) void colsum_xyz_partial_kernel(const FT* __restrict__ x, const float* __restrict__ xyz, int N, int C,
                                                                 float* __restrict__ part, int nchunk, int rows) {
    __shared__ float4 red[4][256];
    const int cq = C >> 2;
    const int tid = threadIdx.x;
    const int g = tid % cq, rl = tid / cq, RL = 256 / cq;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows, r1 = min(N, r0 + rows);
    float4 s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = r0 + rl; i < r1; i += RL) {
        const float4 v = Feat<FT>::ld4(x + ((size_t)b * N + i) * C + (g << 2));
        const float* p3 = xyz + ((size_t)b * N + i) * 3;
        const float w[3] = {p3[0], p3[1], p3[2]};
        s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            // plain v_fma_f32, spelled out: hipcc packs these into v_pk_fma_f32 with op_sel picking the coordinate out of the
            // loaded (x, y, z) register run, and that form returned wrong LOW halves (the .x / .z sums of the y moment, lanes 16-31
            // of each half-wave) whenever another process's MFMA kernels shared the GPU -- 3-8 % of the replays, 1-3 % off
            // (tools/dbg_ste2.py, tests/test_gpu_dp_shared.py); never alone on the GPU.  The scalar form has not deviated once.
            s[q + 1].x = fma_plain(v.x, w[q], s[q + 1].x); s[q + 1].y = fma_plain(v.y, w[q], s[q + 1].y);
            s[q + 1].z = fma_plain(v.z, w[q], s[q + 1].z); s[q + 1].w = fma_plain(v.w, w[q], s[q + 1].w);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][tid] = s[q];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a = s[q];
            for (int l = 1; l < RL; ++l) { const float4 v = red[q][l * cq + g]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            *reinterpret_cast<float4*>(part + (((size_t)b * nchunk + chunk) * 4 + q) * C + (g << 2)) = a;
        }
    }
}

// the same first stage for widths the float4 form does not take (C not a multiple of 4, or C/4 not dividing 256;
// C <= 256): thread = (row lane, column), consecutive threads read consecutive words
__global__ __launch_bounds__(256) void colsum_partial_scalar_kernel(const float* __restrict__ x, int N, int C,
                                                                    float* __restrict__ part, int nchunk, int rows) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int RL = 256 / C;
    const int col = tid % C, rl = tid / C;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows, r1 = min(N, r0 + rows);
    float s = 0.f;
    if (rl < RL)
        for (int i = r0 + rl; i < r1; i += RL) s += x[((size_t)b * N + i) * C + col];
    red[tid] = s;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; ++l) s += red[l * C + col];
        part[((size_t)b * nchunk + chunk) * C + col] = s;
    }
}

// feat rows assembled from column segments (reference FaceRecon.py:100-107: nearest up-sampling gathers,
// one-hot category columns and torch.cat) in ONE pass: segment s of row (b,i) comes from
//   kind 0: src[(b*N + i)*w + c]      kind 1: src[(b*Ns + idx[b*N+i])*w + c]      kind 2: src[b*w + c]
//   kind 3: (long) src[b] == c ? 1 : 0   -- the one-hot category columns built in place from the (B) float ids
//           (FaceRecon.py:80-85: zeros + scatter_ were three launches)
// (bf16 build: kind 0 / 1 sources are bf16 like the output, kind 2 -- the per-cloud one-hot row -- stays fp32 and is
// rounded on the way in; W is the output ROW PITCH, which may exceed the sum of the widths: padding columns stay untouched)
struct ConcatSeg { const void* src; const int32_t* idx; int width; int kind; int nsrc; int col0; };
struct ConcatDesc { ConcatSeg seg[8]; int nseg; int even; int q0[9]; };      // q0: first 16-byte chunk of each segment (even == 2)

template <typename FT>
__global__ __launch_bounds__(256) void concat_rows_kernel(ConcatDesc d, int B, int N, int W, FT* __restrict__ out) {
    using Pair = typename std::conditional<sizeof(FT) == 4, float2, unsigned>::type;       // two elements
    const long long rows = (long long)B * N;
    if (d.even == 2) {
        // 16-byte form: four rows per workgroup pass, a wave per row; a lane walks the row's 16-byte chunks (all segments
        // flattened) with stride 64, so every lane has 2-5 independent loads in flight
        const int lane = threadIdx.x & 63;
        const int total = d.q0[d.nseg];
        for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
            const int b = (int)(row / N);
            FT* o = out + (size_t)row * W;
            for (int q = lane; q < total; q += 64) {
                int sgi = 0;
                while (q >= d.q0[sgi + 1]) ++sgi;
                const ConcatSeg sg = d.seg[sgi];
                const int c = q - d.q0[sgi];
                if (sg.kind >= 2) {                               // per-cloud fp32 row: element c (chunk = one element here)
                    const float* ps = reinterpret_cast<const float*>(sg.src);
                    Feat<FT>::st(o + sg.col0 + c, sg.kind == 2 ? ps[(size_t)b * sg.width + c] : ((long long)ps[b] == c ? 1.f : 0.f));
                    continue;
                }
                const FT* src = reinterpret_cast<const FT*>(sg.src);
                src += sg.kind == 0 ? (size_t)row * sg.width : ((size_t)b * sg.nsrc + sg.idx[row]) * sg.width;
                reinterpret_cast<uint4*>(o + sg.col0)[c] = reinterpret_cast<const uint4*>(src)[c];
            }
            // padding columns of a pitched row (1286 -> 1288): zeroed, so a consumer that reads whole 16-byte chunks sees no garbage
            const int used = d.seg[d.nseg - 1].col0 + d.seg[d.nseg - 1].width;
            if (lane < W - used) Feat<FT>::st(o + used + lane, 0.f);
        }
        return;
    }
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = (int)(row / N);
        FT* o = out + (size_t)row * W;
        for (int s = 0; s < d.nseg; ++s) {
            const ConcatSeg sg = d.seg[s];
            if (sg.kind >= 2) {
                const float* src = reinterpret_cast<const float*>(sg.src);
                for (int c = threadIdx.x; c < sg.width; c += 256)
                    Feat<FT>::st(o + sg.col0 + c, sg.kind == 2 ? src[(size_t)b * sg.width + c] : ((long long)src[b] == c ? 1.f : 0.f));
                continue;
            }
            const FT* src = reinterpret_cast<const FT*>(sg.src);
            if (sg.kind == 0) src += (size_t)row * sg.width;
            else src += ((size_t)b * sg.nsrc + sg.idx[row]) * sg.width;
            if (d.even) {       // every width / column offset / row stride even: two elements per access
                const Pair* s2 = reinterpret_cast<const Pair*>(src);
                Pair* o2 = reinterpret_cast<Pair*>(o + sg.col0);
                for (int c = threadIdx.x; c < (sg.width >> 1); c += 256) o2[c] = s2[c];
            } else {
                for (int c = threadIdx.x; c < sg.width; c += 256) o[sg.col0 + c] = src[c];
            }
        }
    }
}

// scatter-add of the pooled gradient to the winning source rows (grad_feat pre-zeroed)
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const float* __restrict__ gout, int gbcast,
                                                             const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ qsel,
                                                             const uint8_t* __restrict__ argmax, int B, int Nsrc,
                                                             int Nidx, int Nq, int kstride, int C,
                                                             float* __restrict__ gfeat) {
    const int cq = C >> 2;
    const long long total = (long long)B * Nq * cq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const long long row = e / cq;
        const int q = (int)(row % Nq);
        const int b = (int)(row / Nq);
        const int qi = qsel ? qsel[q] : q;
        const int32_t* nb = idx + ((size_t)b * Nidx + qi) * kstride;
        const float4 gv = *reinterpret_cast<const float4*>(gout + (gbcast ? (size_t)b * C : (size_t)row * C) + (g << 2));
        const uchar4 am = *reinterpret_cast<const uchar4*>(argmax + row * C + (g << 2));
        float* gb = gfeat + (size_t)b * Nsrc * C + (g << 2);
        if (gv.x != 0.f) atomicAdd(gb + (size_t)nb[am.x] * C + 0, gv.x);
        if (gv.y != 0.f) atomicAdd(gb + (size_t)nb[am.y] * C + 1, gv.y);
        if (gv.z != 0.f) atomicAdd(gb + (size_t)nb[am.z] * C + 2, gv.z);
        if (gv.w != 0.f) atomicAdd(gb + (size_t)nb[am.w] * C + 3, gv.w);
    }
}

// COLUMN-TILE LDS-SCATTER form of the pooling backward (default): one workgroup per (cloud, TC columns)
// keeps acc[Nsrc][TC] in LDS, sweeps the queries and adds into the winning source row with LDS atomics,
// then writes every grad_feat row segment once -- no memset, no global atomics.  With a broadcast
// gradient (ORL: d fg / N for every query) the accumulated quantity is an integer COUNT, so that
// branch is exactly reproducible.  MODE 0: arg-max routed (gather_max), MODE 1: plain row scatter
// (gather_rows: nearest-neighbour up-sampling backward, many queries per source row).
// grid (C/TC, B), block NT, dynamic LDS = Nsrc*TC*4.  NT: the kernel is a chain of memory round trips per workgroup (zero the tile;
// per pass of FL points: arg-max bytes + gradients, then the dependent neighbour indices, then the LDS adds; flush), and at the
// stack's widths the grid is 128 workgroups (C = 128, TC = 16, B = 16) -- half the chip, each CU holding ONE 256-thread workgroup
// that walks its 1028 points in 4 passes.  With 1024 threads the same workgroup does it in one: the round trips of a pass are
// in flight for four times the points (launch_scatter_tile picks NT from the grid).
// FT: storage type of gout (per-query gradients) / gfeat / extra; the broadcast row `gbc` (B,C) is always fp32
template <int TC, int MODE, typename FT, int NT>
__global__ __launch_bounds__(NT) void scatter_tile_bwd_kernel(const FT* __restrict__ gout, const float* __restrict__ gbc,
                                                               int gstride,
                                                               int gbcast, const int32_t* __restrict__ idx,
                                                               int idx_shared, const int32_t* __restrict__ qsel,
                                                               const uint8_t* __restrict__ argmax, int Nsrc,
                                                               int Nidx, int Nq, int kstride, int C,
                                                               FT* __restrict__ gfeat, int accumulate,
                                                               const FT* __restrict__ extra) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = reinterpret_cast<float*>(smem);
    int* cnt = reinterpret_cast<int*>(smem);
    constexpr int G = TC / 4;
    constexpr int PL = NT / G;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int j0 = blockIdx.x * TC;
    const int cg = tid % G, pl = tid / G;
    const int j = j0 + cg * 4;
    ORL_STAMP(40);
    for (int q = tid; q < Nsrc * G; q += NT) *reinterpret_cast<float4*>(acc + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    ORL_STAMP(41);
    // FL points per pass: their arg-max bytes / gradients first, then the dependent neighbour-index gathers, then the
    // LDS adds -- two global round trips per pass instead of two per point
#ifndef SCATTER_FL
#define SCATTER_FL 4
#endif
    constexpr int FL = SCATTER_FL;
    for (int q0 = pl; q0 < Nq; q0 += PL * FL) {
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < FL; ++u) {
                const int q = q0 + u * PL;
                if (q < Nq) {
                    const size_t row = (size_t)b * Nq + q;
                    const int m = idx_shared ? idx[q] : idx[row];
                    // rows of a column block of a wider tensor are only 8-byte aligned (stride 1286): two float2 loads
                    float2 g01, g23;
                    if constexpr (sizeof(FT) == 4) {
                        g01 = *reinterpret_cast<const float2*>(gout + row * gstride + j);
                        g23 = *reinterpret_cast<const float2*>(gout + row * gstride + j + 2);
                    } else {                                          // (rows of a bf16 column block: 4-byte aligned)
                        const unsigned u0 = *reinterpret_cast<const unsigned*>(gout + row * gstride + j);
                        const unsigned u1 = *reinterpret_cast<const unsigned*>(gout + row * gstride + j + 2);
                        g01 = make_float2(__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u));
                        g23 = make_float2(__uint_as_float(u1 << 16), __uint_as_float(u1 & 0xffff0000u));
                    }
                    float* a = acc + m * TC + cg * 4;
                    if (g01.x != 0.f) atomicAdd(a + 0, g01.x);
                    if (g01.y != 0.f) atomicAdd(a + 1, g01.y);
                    if (g23.x != 0.f) atomicAdd(a + 2, g23.x);
                    if (g23.y != 0.f) atomicAdd(a + 3, g23.y);
                }
            }
        } else {
            uchar4 am[FL];
            float4 gv[FL];
            const int32_t* nb[FL];
#pragma unroll
            for (int u = 0; u < FL; ++u) {
                const int q = min(q0 + u * PL, Nq - 1);
                const size_t row = (size_t)b * Nq + q;
                const int qi = qsel ? qsel[q] : q;
                nb[u] = idx + ((size_t)b * Nidx + qi) * kstride;
                am[u] = *reinterpret_cast<const uchar4*>(argmax + row * C + j);
                gv[u] = gbcast ? make_float4(0.f, 0.f, 0.f, 0.f) : Feat<FT>::ld4(gout + row * gstride + j);
            }
            int m0[FL], m1[FL], m2[FL], m3[FL];
#pragma unroll
            for (int u = 0; u < FL; ++u) { m0[u] = nb[u][am[u].x]; m1[u] = nb[u][am[u].y]; m2[u] = nb[u][am[u].z]; m3[u] = nb[u][am[u].w]; }
#pragma unroll
            for (int u = 0; u < FL; ++u) {
                if (q0 + u * PL < Nq) {
                    if (gbcast) {
                        atomicAdd(cnt + m0[u] * TC + cg * 4 + 0, 1);
                        atomicAdd(cnt + m1[u] * TC + cg * 4 + 1, 1);
                        atomicAdd(cnt + m2[u] * TC + cg * 4 + 2, 1);
                        atomicAdd(cnt + m3[u] * TC + cg * 4 + 3, 1);
                    } else {
                        if (gv[u].x != 0.f) atomicAdd(acc + m0[u] * TC + cg * 4 + 0, gv[u].x);
                        if (gv[u].y != 0.f) atomicAdd(acc + m1[u] * TC + cg * 4 + 1, gv[u].y);
                        if (gv[u].z != 0.f) atomicAdd(acc + m2[u] * TC + cg * 4 + 2, gv[u].z);
                        if (gv[u].w != 0.f) atomicAdd(acc + m3[u] * TC + cg * 4 + 3, gv[u].w);
                    }
                }
            }
        }
    }
    __syncthreads();
    ORL_STAMP(42);
    // flush: every thread handles FL (row, group) elements at a time with all their global loads issued before the
    // first dependent add (one element per iteration left the 2-3 loads of each store as a serial latency chain:
    // 23 of the kernel's 31 us at B=16 N=1028 C=128)
    float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0 && gbcast) gb = *reinterpret_cast<const float4*>(gbc + (size_t)b * C + j);   // (q % G == cg for all of a thread's q)
    const int total = Nsrc * G;
    for (int q0 = tid; q0 < total; q0 += NT * FL) {
        float4 o[FL], x[FL];
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int q = min(q0 + u * NT, total - 1);
            const int m = q / G, g4 = q - m * G;
            const size_t off = ((size_t)b * Nsrc + m) * C + j0 + g4 * 4;
            o[u] = accumulate ? Feat<FT>::ld4(gfeat + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            x[u] = extra ? Feat<FT>::ld4(extra + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int q = q0 + u * NT;
            if (q < total) {
                const int m = q / G, g4 = q - m * G;
                float4 v;
                if (MODE == 0 && gbcast) {
                    const int4 cv = *reinterpret_cast<const int4*>(cnt + m * TC + g4 * 4);
                    v = make_float4(gb.x * cv.x, gb.y * cv.y, gb.z * cv.z, gb.w * cv.w);
                } else {
                    v = *reinterpret_cast<const float4*>(acc + m * TC + g4 * 4);
                }
                if (accumulate) { v.x += o[u].x; v.y += o[u].y; v.z += o[u].z; v.w += o[u].w; }
                if (extra) { v.x += x[u].x; v.y += x[u].y; v.z += x[u].z; v.w += x[u].w; }
                Feat<FT>::st4(gfeat + ((size_t)b * Nsrc + m) * C + j0 + g4 * 4, v);
            }
        }
    }
    ORL_STAMP(43);
}

// out[b,i,:] += f[b,i,:] + t[b,:]   -- the residual + per-cloud ORL bias of an HS layer in one pass
// (reference: "conv2(cat[feature, f_global]) + feature", gcn3d.py:112,186, with the f_global half of conv2
// reduced to the per-cloud row t = f_global Wb^T)
__global__ __launch_bounds__(256) void residual_bias_kernel(float* __restrict__ out, const float* __restrict__ f,
                                                            const float* __restrict__ t, long long total4, int N,
                                                            int C) {
    const int cq = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const int b = (int)((e / cq) / N);
        float4 o = *reinterpret_cast<const float4*>(out + e * 4);
        const float4 a = *reinterpret_cast<const float4*>(f + e * 4);
        const float4 tb = *reinterpret_cast<const float4*>(t + (size_t)b * C + (g << 2));
        o.x += a.x + tb.x; o.y += a.y + tb.y; o.z += a.z + tb.z; o.w += a.w + tb.w;
        *reinterpret_cast<float4*>(out + e * 4) = o;
    }
}

// out[r][c] = (ga[r][c] (+ gb[r][c])) * [y[r][c] > 0]: the backward of an in-place relu on a tensor with two consumers (fm_0 of
// FaceRecon.py:88: conv_1 and the concat) -- autograd's add and threshold_backward in one pass; ga / gb rows of an even pitch
__global__ __launch_bounds__(256) void add_relu_bwd_kernel(const float* __restrict__ ga, int lda, const float* __restrict__ gb, int ldb,
                                                           const float* __restrict__ y, long long total2, int C2,
                                                           float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total2; e += (long long)gridDim.x * 256) {
        const unsigned row = (unsigned)((unsigned long long)e / (unsigned)C2);
        const int c = (int)(e - (long long)row * C2) * 2;
        float2 a = *reinterpret_cast<const float2*>(ga + (size_t)row * lda + c);
        if (gb) { const float2 b = *reinterpret_cast<const float2*>(gb + (size_t)row * ldb + c); a.x += b.x; a.y += b.y; }
        const float2 yy = *reinterpret_cast<const float2*>(y + e * 2);
        *reinterpret_cast<float2*>(out + e * 2) = make_float2(yy.x > 0.f ? a.x : 0.f, yy.y > 0.f ? a.y : 0.f);
    }
}

static int pick_scatter_cols(int Nsrc, int C) {
    for (int tc = 16; tc >= 4; tc >>= 1)
        if (C % tc == 0 && (size_t)Nsrc * tc * 4 <= 144 * 1024) return tc;
    return 0;
}

template <int MODE, typename FT>
static int launch_scatter_tile(int tc, const FT* gout, const float* gbc, int gstride, int gbcast, const int32_t* idx,
                               int idx_shared, const int32_t* qsel, const uint8_t* argmax, int B, int Nsrc, int Nidx, int Nq,
                               int kstride, int C, FT* gfeat, int accumulate, const FT* extra, hipStream_t st) {
    const size_t lds = (size_t)Nsrc * tc * 4;
    dim3 grid(C / tc, B);
    // threads per workgroup: 1024 while the grid covers at most half the CUs, 512 up to one workgroup per CU (measured at B = 16:
    // N = 1028, C = 128 -- 128 workgroups -- 17.7 -> 15.6 us with 1024 threads, the flush 9 200 -> 3 500 clocks; N = 257, C = 256 --
    // 256 workgroups -- no better with 1024 than with 256: 9.3 vs 8.9 us)
    const long long wgs = (long long)(C / tc) * B;
    const int nt = 2 * wgs <= HSP_NUM_CU ? 1024 : wgs <= HSP_NUM_CU ? 512 : 256;
#define SC_LAUNCH_NT(TC, NT_)                                                                                      \
    {                                                                                                              \
        auto kern = scatter_tile_bwd_kernel<TC, MODE, FT, NT_>;                                                    \
        if (lds > 64 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                 \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(NT_), lds, st, gout, gbc, gstride, gbcast, idx, idx_shared, qsel, argmax, Nsrc, \
                           Nidx, Nq, kstride, C, gfeat, accumulate, extra);                                        \
    }
#define SC_LAUNCH(TC)                                                                                              \
    {                                                                                                              \
        if (nt == 1024) SC_LAUNCH_NT(TC, 1024) else if (nt == 512) SC_LAUNCH_NT(TC, 512) else SC_LAUNCH_NT(TC, 256) \
    }
    if (tc == 16) SC_LAUNCH(16) else if (tc == 8) SC_LAUNCH(8) else SC_LAUNCH(4)
#undef SC_LAUNCH_NT
#undef SC_LAUNCH
    return check_launch();
}

// gather form of the same backward over the reverse-edge index (csr.hip): one thread per
// (source row m, float4 column group) walks the edges e = i*k + n pointing at m and adds
// grad_out[b,i,c] (or the per-cloud broadcast row) where argmax[b,i,c] == n.  No atomics.
__global__ __launch_bounds__(256) void gather_max_bwd_csr_kernel(const float* __restrict__ gout, int gbcast,
                                                                 const uint8_t* __restrict__ argmax,
                                                                 const int32_t* __restrict__ rev_off,
                                                                 const int32_t* __restrict__ rev_edge, int B,
                                                                 int Nsrc, int Nq, int k, int C,
                                                                 float* __restrict__ gfeat) {
    const int cq = C >> 2;
    const long long total = (long long)B * Nsrc * cq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const long long row = e / cq;            // b*Nsrc + m
        const int m = (int)(row % Nsrc);
        const int b = (int)(row / Nsrc);
        const int32_t* off = rev_off + (size_t)b * (Nsrc + 1);
        const int32_t* edge = rev_edge + (size_t)b * Nq * k;
        const int o0 = off[m], o1 = off[m + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gbcast) {
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int p = o0; p < o1; ++p) {
                const int ed = edge[p];
                const int i = ed / k, n = ed - i * k;
                const uchar4 am = *reinterpret_cast<const uchar4*>(argmax + ((size_t)b * Nq + i) * C + (g << 2));
                c0 += am.x == n; c1 += am.y == n; c2 += am.z == n; c3 += am.w == n;
            }
            const float4 gv = *reinterpret_cast<const float4*>(gout + (size_t)b * C + (g << 2));
            // (count * g): equals the serial sum of count copies of g up to one rounding
            acc = make_float4(gv.x * c0, gv.y * c1, gv.z * c2, gv.w * c3);
        } else {
            for (int p = o0; p < o1; ++p) {
                const int ed = edge[p];
                const int i = ed / k, n = ed - i * k;
                const size_t qrow = (size_t)b * Nq + i;
                const uchar4 am = *reinterpret_cast<const uchar4*>(argmax + qrow * C + (g << 2));
                const float4 gv = *reinterpret_cast<const float4*>(gout + qrow * C + (g << 2));
                if (am.x == n) acc.x += gv.x;
                if (am.y == n) acc.y += gv.y;
                if (am.z == n) acc.z += gv.z;
                if (am.w == n) acc.w += gv.w;
            }
        }
        *reinterpret_cast<float4*>(gfeat + row * C + (g << 2)) = acc;
    }
}

__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const float* __restrict__ feat,
                                                              const int32_t* __restrict__ idx, int idx_shared,
                                                              int B, int Nsrc, int Nq, int C,
                                                              float* __restrict__ out, int out_stride) {
    const int cq = C >> 2;
    const long long total = (long long)B * Nq * cq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const long long row = e / cq;
        const int q = (int)(row % Nq);
        const int b = (int)(row / Nq);
        const int m = idx_shared ? idx[q] : idx[row];
        const float4 f = *reinterpret_cast<const float4*>(feat + ((size_t)b * Nsrc + m) * C + (g << 2));
        float* o = out + (size_t)row * out_stride + (g << 2);
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = f.w;   // out_stride need not be a multiple of 4
    }
}

// rows of 3 floats (xyz): scalar variant
__global__ __launch_bounds__(256) void gather_rows_fwd_scalar_kernel(const float* __restrict__ feat,
                                                                     const int32_t* __restrict__ idx,
                                                                     int idx_shared, int B, int Nsrc, int Nq, int C,
                                                                     float* __restrict__ out, int out_stride) {
    const long long total = (long long)B * Nq * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long long row = e / C;
        const int q = (int)(row % Nq);
        const int b = (int)(row / Nq);
        const int m = idx_shared ? idx[q] : idx[row];
        out[(size_t)row * out_stride + c] = feat[((size_t)b * Nsrc + m) * C + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ gout, int gstride,
                                                              const int32_t* __restrict__ idx, int idx_shared,
                                                              int B, int Nsrc, int Nq, int C,
                                                              float* __restrict__ gfeat) {
    const long long total = (long long)B * Nq * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long long row = e / C;
        const int q = (int)(row % Nq);
        const int b = (int)(row / Nq);
        const int m = idx_shared ? idx[q] : idx[row];
        const float g = gout[(size_t)row * gstride + c];
        if (g != 0.f) atomicAdd(gfeat + ((size_t)b * Nsrc + m) * C + c, g);
    }
}

// points per chunk for the two-stage per-cloud reductions: enough chunks to fill the chip, at least one
// row per row-lane, at most ORL_ROWS (the workspace is sized for the smallest chunk count = ORL_ROWS rows)
static bool colsum_vec4(int C) { return !(C & 3) && !(256 % (C >> 2)); }

static int chunk_rows(int B, int N, int C) {
    const int RL = colsum_vec4(C) ? 256 / (C >> 2) : (C <= 256 ? 256 / C : 1);
    long long r = ((long long)B * N + 511) / 512;           // ~512 workgroups ...
    if (r < (N + 31) / 32) r = (N + 31) / 32;               // ... but at most 32 chunks per cloud for the serial fold
    if (r < RL) r = RL;
    if (r > ORL_ROWS) r = ORL_ROWS;
    return (int)r;
}

static int stream_grid(long long threads) {
    long long g = (threads + 255) / 256;
    const long long cap = (long long)HSP_NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- max over the points of a cloud (the heads' torch.max(x, 2)[0]) ---------------------------------------------------
// grid (ceil(C/64), B), 1024 threads = 16 row groups x 64 channels: wave g scans rows g, g+16, ... of its 64-channel strip
// (256 B per wave per row), the 16 partial (value, first row) pairs fold through LDS.
__global__ __launch_bounds__(1024) void points_max_fwd_kernel(const float* __restrict__ x, int N, int C,
                                                              float* __restrict__ out, int32_t* __restrict__ arg) {
    __shared__ float s_v[16][64];
    __shared__ int s_i[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
    float best = -INFINITY;
    int bi = INT_MAX;
    if (c < C) {
        const float* col = x + (size_t)b * N * C + c;
        int n = g;
        for (; n + 48 < N; n += 64) {                     // four rows in flight per lane
            const float v0 = col[(size_t)n * C], v1 = col[(size_t)(n + 16) * C], v2 = col[(size_t)(n + 32) * C],
                        v3 = col[(size_t)(n + 48) * C];
            if (v0 > best || (v0 != v0 && best == best)) { best = v0; bi = n; }
            if (v1 > best || (v1 != v1 && best == best)) { best = v1; bi = n + 16; }
            if (v2 > best || (v2 != v2 && best == best)) { best = v2; bi = n + 32; }
            if (v3 > best || (v3 != v3 && best == best)) { best = v3; bi = n + 48; }
        }
        for (; n < N; n += 16) {
            const float v = col[(size_t)n * C];
            if (v > best || (v != v && best == best)) { best = v; bi = n; }
        }
    }
    s_v[g][lane] = best;
    s_i[g][lane] = bi;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            const float v = s_v[j][lane];
            const int i = s_i[j][lane];
            const bool vnan = v != v, bnan = best != best;
            if ((vnan && !bnan) || (!bnan && v > best) || (vnan == bnan && v == best && i < bi) ||
                (vnan && bnan && i < bi)) { best = v; bi = i; }
        }
        out[(size_t)b * C + c] = best;
        arg[(size_t)b * C + c] = bi == INT_MAX ? 0 : bi;
    }
}

// grad_x (B,N,C) written in one pass: grad_out[b,c] on the winning row, 0 elsewhere (no memset + scatter)
__global__ __launch_bounds__(256) void points_max_bwd_kernel(const float* __restrict__ gout,
                                                             const int32_t* __restrict__ arg, int N, int C4,
                                                             long long total, float* __restrict__ gx) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        const long long r = t / C4;
        const int n = (int)(r % N);
        const size_t bc = (size_t)(r / N) * C4 + c4;
        const int4 a = reinterpret_cast<const int4*>(arg)[bc];
        const float4 g = reinterpret_cast<const float4*>(gout)[bc];
        float4 o;
        o.x = a.x == n ? g.x : 0.f;
        o.y = a.y == n ? g.y : 0.f;
        o.z = a.z == n ? g.z : 0.f;
        o.w = a.w == n ? g.w : 0.f;
        reinterpret_cast<float4*>(gx)[t] = o;
    }
}

}  // namespace hsp

using namespace hsp;

template <typename FT>
static int gather_max_fwd_impl(const FT* feat, const int32_t* idx, const int32_t* qsel, int B, int Nsrc,
                               int Nidx, int Nq, int k, int kstride, int C, FT* out, uint8_t* argmax,
                               hspStream_t stream) {
    if (!feat || !idx || !out || !argmax || B <= 0 || Nsrc <= 0 || Nidx <= 0 || Nq <= 0 || k <= 0 || kstride < k || C <= 0)
        return HSP_ERR_BAD_ARG;
    if (!qsel && Nq != Nidx) return HSP_ERR_BAD_ARG;
    if ((C & 3) || k > 255) return HSP_ERR_UNSUPPORTED;
    const long long total = (long long)B * Nq * (C >> 2);
    hipLaunchKernelGGL(gather_max_fwd_kernel<FT>, dim3(stream_grid(total)), dim3(256), 0, as_stream(stream), feat, idx,
                       qsel, B, Nsrc, Nidx, Nq, k, kstride, C, out, argmax);
    return check_launch();
}
extern "C" int hsp_gather_max_fwd(const float* feat, const int32_t* idx, const int32_t* qsel, int B, int Nsrc,
                                  int Nidx, int Nq, int k, int kstride, int C, float* out, uint8_t* argmax,
                                  hspStream_t stream) {
    return gather_max_fwd_impl<float>(feat, idx, qsel, B, Nsrc, Nidx, Nq, k, kstride, C, out, argmax, stream);
}
extern "C" int hsp_gather_max_fwd_bf16(const hsp_bf16_t* feat, const int32_t* idx, const int32_t* qsel, int B, int Nsrc,
                                       int Nidx, int Nq, int k, int kstride, int C, hsp_bf16_t* out, uint8_t* argmax,
                                       hspStream_t stream) {
    return gather_max_fwd_impl<bf16_t>(feat, idx, qsel, B, Nsrc, Nidx, Nq, k, kstride, C, out, argmax, stream);
}

/* Pool_layer in one launch (gcn3d.py:236-245): max over the k listed neighbours of the kept rows qsel + the kept rows' coordinates */
extern "C" int hsp_pool_fwd(const float* feat, const float* xyz, const int32_t* idx, const int32_t* qsel, int B, int N, int Nq,
                            int k, int kstride, int C, float* out, uint8_t* argmax, float* xyz_sel, hspStream_t stream) {
    if (!feat || !xyz || !idx || !qsel || !out || !argmax || !xyz_sel || B <= 0 || N <= 0 || Nq <= 0 || k <= 0 || kstride < k || C < 12)
        return HSP_ERR_BAD_ARG;
    if ((C & 3) || k > 255) return HSP_ERR_UNSUPPORTED;
    const long long total = (long long)B * Nq * (C >> 2);
    hipLaunchKernelGGL(gather_max_fwd_kernel<float>, dim3(stream_grid(total)), dim3(256), 0, as_stream(stream), feat, idx, qsel, B,
                       N, N, Nq, k, kstride, C, out, argmax, xyz, xyz_sel);
    return check_launch();
}

template <typename FT>
static int gather_max_bwd_impl(const void* grad_out, int grad_bcast, const int32_t* idx, const int32_t* qsel,
                               const uint8_t* argmax, int B, int Nsrc, int Nidx, int Nq, int kstride, int C,
                               FT* grad_feat, int accumulate, const FT* extra, hspStream_t stream) {
    if (!grad_out || !idx || !argmax || !grad_feat || B <= 0 || Nsrc <= 0 || Nidx <= 0 || Nq <= 0 || kstride <= 0 || C <= 0)
        return HSP_ERR_BAD_ARG;
    if (!qsel && Nq != Nidx) return HSP_ERR_BAD_ARG;
    if (C & 3) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    // the per-query gradient has the feature type; the broadcast row (ORL: d fg / N per cloud) is always fp32
    if (const int tc = pick_scatter_cols(Nsrc, C))
        return launch_scatter_tile<0, FT>(tc, grad_bcast ? nullptr : reinterpret_cast<const FT*>(grad_out),
                                          grad_bcast ? reinterpret_cast<const float*>(grad_out) : nullptr, C, grad_bcast, idx, 0,
                                          qsel, argmax, B, Nsrc, Nidx, Nq, kstride, C, grad_feat, accumulate, extra, st);
    if constexpr (sizeof(FT) == 4) {
        if (extra) return HSP_ERR_UNSUPPORTED;             // the global-atomic fallback has no fused add
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(grad_feat, 0, (size_t)B * Nsrc * C * sizeof(float), st);
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
        }
        const long long total = (long long)B * Nq * (C >> 2);
        hipLaunchKernelGGL(gather_max_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(grad_out), grad_bcast, idx, qsel, argmax, B, Nsrc, Nidx, Nq, kstride, C,
                           grad_feat);
        return check_launch();
    }
    return HSP_ERR_UNSUPPORTED;                            // bf16: the LDS tile form only (Nsrc * 16 bytes <= 144 KiB)
}
extern "C" int hsp_gather_max_bwd(const float* grad_out, int grad_bcast, const int32_t* idx, const int32_t* qsel,
                                  const uint8_t* argmax, int B, int Nsrc, int Nidx, int Nq, int kstride, int C,
                                  float* grad_feat, int accumulate, const float* extra, hspStream_t stream) {
    return gather_max_bwd_impl<float>(grad_out, grad_bcast, idx, qsel, argmax, B, Nsrc, Nidx, Nq, kstride, C, grad_feat,
                                      accumulate, extra, stream);
}
extern "C" int hsp_gather_max_bwd_bf16(const void* grad_out, int grad_bcast, const int32_t* idx, const int32_t* qsel,
                                       const uint8_t* argmax, int B, int Nsrc, int Nidx, int Nq, int kstride, int C,
                                       hsp_bf16_t* grad_feat, int accumulate, const hsp_bf16_t* extra, hspStream_t stream) {
    return gather_max_bwd_impl<bf16_t>(grad_out, grad_bcast, idx, qsel, argmax, B, Nsrc, Nidx, Nq, kstride, C, grad_feat,
                                       accumulate, extra, stream);
}

extern "C" int hsp_gather_rows_fwd(const float* feat, const int32_t* idx, int idx_shared, int B, int Nsrc, int Nq,
                                   int C, float* out, int out_stride, hspStream_t stream) {
    if (!feat || !idx || !out || B <= 0 || Nsrc <= 0 || Nq <= 0 || C <= 0 || out_stride < C) return HSP_ERR_BAD_ARG;
    hipStream_t st = as_stream(stream);
    if ((C & 3) == 0) {
        const long long total = (long long)B * Nq * (C >> 2);
        hipLaunchKernelGGL(gather_rows_fwd_kernel, dim3(stream_grid(total)), dim3(256), 0, st, feat, idx, idx_shared, B,
                           Nsrc, Nq, C, out, out_stride);
    } else {
        const long long total = (long long)B * Nq * C;
        hipLaunchKernelGGL(gather_rows_fwd_scalar_kernel, dim3(stream_grid(total)), dim3(256), 0, st, feat, idx,
                           idx_shared, B, Nsrc, Nq, C, out, out_stride);
    }
    return check_launch();
}

extern "C" int hsp_gather_rows_bwd(const float* grad_out, int grad_stride, const int32_t* idx, int idx_shared, int B,
                                   int Nsrc, int Nq, int C, float* grad_feat, hspStream_t stream) {
    if (!grad_out || !idx || !grad_feat || B <= 0 || Nsrc <= 0 || Nq <= 0 || C <= 0 || grad_stride < C)
        return HSP_ERR_BAD_ARG;
    hipStream_t st = as_stream(stream);
    if ((C & 3) == 0 && (grad_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 7) == 0)
        if (const int tc = pick_scatter_cols(Nsrc, C))
            return launch_scatter_tile<1, float>(tc, grad_out, nullptr, grad_stride, 0, idx, idx_shared, nullptr, nullptr, B,
                                                 Nsrc, Nq, Nq, 1, C, grad_feat, 0, nullptr, st);
    hipError_t e = hipMemsetAsync(grad_feat, 0, (size_t)B * Nsrc * C * sizeof(float), st);
    if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    const long long total = (long long)B * Nq * C;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, st, grad_out, grad_stride, idx,
                       idx_shared, B, Nsrc, Nq, C, grad_feat);
    return check_launch();
}

// gather form of hsp_gather_rows_bwd over the reverse index of the (B,Nq) row map: a workgroup row-lane owns one
// SOURCE row m and sums the gradient rows of the queries that selected it, in ascending query order (no atomics,
// bit-reproducible), reading every gradient row segment whole.  The column-tile scatter form reads a 64-byte
// slice of every 5 KB gradient row per tile: 30-45 us where this takes under 10.
template <typename FT>
__global__ __launch_bounds__(256) void gather_rows_bwd_csr_kernel(const FT* __restrict__ gout, int gstride,
                                                                  const int32_t* __restrict__ rev_off,
                                                                  const int32_t* __restrict__ rev_edge, int B,
                                                                  int Nsrc, int Nq, int C,
                                                                  FT* __restrict__ gfeat) {
    const int pairs = C >> 1;                               // float2 columns (gradient rows are 8-byte aligned)
    const int tpr = pairs < 256 ? pairs : 256;              // threads per row
    const int RB = 256 / tpr;
    const int rl = threadIdx.x / tpr, t = threadIdx.x - rl * tpr;
    const long long row = (long long)blockIdx.x * RB + rl;  // b*Nsrc + m
    if (rl >= RB || row >= (long long)B * Nsrc) return;
    const int m = (int)(row % Nsrc), b = (int)(row / Nsrc);
    const int32_t* off = rev_off + (size_t)b * (Nsrc + 1);
    const int32_t* edge = rev_edge + (size_t)b * Nq;
    const int o0 = off[m], o1 = off[m + 1];
    const FT* gb = gout + (size_t)b * Nq * gstride;
    for (int p = t; p < pairs; p += tpr) {
        float2 acc = make_float2(0.f, 0.f);
        for (int e = o0; e < o1; e += 4) {                  // 4 rows in flight, added in edge order
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = edge[min(e + u, o1 - 1)];
                if constexpr (sizeof(FT) == 4) {
                    v[u] = *reinterpret_cast<const float2*>(gb + (size_t)q * gstride + 2 * p);
                } else {
                    const unsigned w = *reinterpret_cast<const unsigned*>(gb + (size_t)q * gstride + 2 * p);
                    v[u] = make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < o1) { acc.x += v[u].x; acc.y += v[u].y; }
        }
        if constexpr (sizeof(FT) == 4) *reinterpret_cast<float2*>(gfeat + (size_t)row * C + 2 * p) = acc;
        else *reinterpret_cast<unsigned*>(gfeat + (size_t)row * C + 2 * p) = f32_to_bf16_bits(acc.x) | (f32_to_bf16_bits(acc.y) << 16);
    }
}

template <typename FT>
static int gather_rows_bwd_csr_impl(const FT* grad_out, int grad_stride, const int32_t* rev_off,
                                    const int32_t* rev_edge, int B, int Nsrc, int Nq, int C, FT* grad_feat,
                                    hspStream_t stream) {
    if (!grad_out || !rev_off || !rev_edge || !grad_feat || B <= 0 || Nsrc <= 0 || Nq <= 0 || C <= 0 || grad_stride < C)
        return HSP_ERR_BAD_ARG;
    if ((C & 1) || (grad_stride & 1) || (reinterpret_cast<uintptr_t>(grad_out) & (2 * sizeof(FT) - 1)) ||
        (256 % ((C >> 1) < 256 ? (C >> 1) : 256)))
        return HSP_ERR_UNSUPPORTED;                         // two-element columns, whole rows per workgroup
    const int tpr = (C >> 1) < 256 ? (C >> 1) : 256;
    const int RB = 256 / tpr;
    const long long rows = (long long)B * Nsrc;
    hipLaunchKernelGGL(gather_rows_bwd_csr_kernel<FT>, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), 0, as_stream(stream),
                       grad_out, grad_stride, rev_off, rev_edge, B, Nsrc, Nq, C, grad_feat);
    return check_launch();
}
extern "C" int hsp_gather_rows_bwd_csr(const float* grad_out, int grad_stride, const int32_t* rev_off,
                                       const int32_t* rev_edge, int B, int Nsrc, int Nq, int C, float* grad_feat,
                                       hspStream_t stream) {
    return gather_rows_bwd_csr_impl<float>(grad_out, grad_stride, rev_off, rev_edge, B, Nsrc, Nq, C, grad_feat, stream);
}
extern "C" int hsp_gather_rows_bwd_csr_bf16(const hsp_bf16_t* grad_out, int grad_stride, const int32_t* rev_off,
                                            const int32_t* rev_edge, int B, int Nsrc, int Nq, int C, hsp_bf16_t* grad_feat,
                                            hspStream_t stream) {
    return gather_rows_bwd_csr_impl<bf16_t>(grad_out, grad_stride, rev_off, rev_edge, B, Nsrc, Nq, C, grad_feat, stream);
}

extern "C" int hsp_gather_max_bwd_csr(const float* grad_out, int grad_bcast, const uint8_t* argmax,
                                      const int32_t* rev_off, const int32_t* rev_edge, int B, int Nsrc, int Nq, int k,
                                      int C, float* grad_feat, hspStream_t stream) {
    if (!grad_out || !argmax || !rev_off || !rev_edge || !grad_feat || B <= 0 || Nsrc <= 0 || Nq <= 0 || k <= 0 || C <= 0)
        return HSP_ERR_BAD_ARG;
    if (C & 3) return HSP_ERR_UNSUPPORTED;
    const long long total = (long long)B * Nsrc * (C >> 2);
    hipLaunchKernelGGL(gather_max_bwd_csr_kernel, dim3(stream_grid(total)), dim3(256), 0, as_stream(stream), grad_out,
                       grad_bcast, argmax, rev_off, rev_edge, B, Nsrc, Nq, k, C, grad_feat);
    return check_launch();
}

extern "C" int hsp_points_max_fwd(const float* x, int B, int N, int C, float* out, int32_t* argrow, hspStream_t stream) {
    if (!x || !out || !argrow || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(points_max_fwd_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, as_stream(stream), x, N, C, out,
                       argrow);
    return check_launch();
}

extern "C" int hsp_points_max_bwd(const float* grad_out, const int32_t* argrow, int B, int N, int C, float* grad_x,
                                  hspStream_t stream) {
    if (!grad_out || !argrow || !grad_x || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (C & 3) return HSP_ERR_UNSUPPORTED;
    const long long total = (long long)B * N * (C >> 2);
    hipLaunchKernelGGL(points_max_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, as_stream(stream), grad_out,
                       argrow, N, C >> 2, total, grad_x);
    return check_launch();
}

#ifdef HSP_ORL_PROF
extern "C" int hsp_debug_set_orl_prof(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(hsp::g_orl_prof), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t hsp_orl_workspace_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C <= 0 || (!colsum_vec4(C) && C > 256)) return 0;
    const int rows = chunk_rows(B, N, C);
    return (size_t)B * ((N + rows - 1) / rows) * C * sizeof(float);
}

template <typename FT>
static int orl_global_fwd_impl(const FT* feat, const int32_t* idx, int B, int N, int k, int kstride, int C,
                               float* fg, uint8_t* argmax, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!feat || !idx || !fg || !argmax || B <= 0 || N <= 0 || k <= 0 || kstride < k || C <= 0) return HSP_ERR_BAD_ARG;
    if ((C & 3) || (256 % (C >> 2)) || k > 255) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_orl_workspace_bytes(B, N, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const size_t lds_tile = (size_t)N * (sizeof(FT) == 4 ? 48 : 24) + (ORLT_WG / 2) * (size_t)k * sizeof(short);   // slab + the waves' list strips
    if ((C % ORLT_TC) == 0 && lds_tile <= 144 * 1024 && N >= 128 && N <= 21845 && k == 20 && kstride == k) {   // fg written by the kernel itself
        auto kern = orl_tile_kernel<FT, 20>;
        if (lds_tile > 64 * 1024) {                          // per device, so on every such launch (a process may drive several GPUs)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
        }
        hipLaunchKernelGGL(kern, dim3(C / ORLT_TC * B), dim3(ORLT_WG), lds_tile, st, feat, idx, B, N, C, argmax, fg, 1.0f / (float)N);
        return check_launch();
    }
    const int rows = chunk_rows(B, N, C);
    const int nchunk = (N + rows - 1) / rows;
    float* part = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(orl_partial_kernel<FT>, dim3(nchunk, B), dim3(256), 0, st, feat, idx, N, k, kstride, C, argmax, part, nchunk, rows);
    hipLaunchKernelGGL(chunk_fold_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, nchunk, C, 1.0f / (float)N, fg);
    return check_launch();
}
extern "C" int hsp_orl_global_fwd(const float* feat, const int32_t* idx, int B, int N, int k, int kstride, int C,
                                  float* fg, uint8_t* argmax, void* ws, size_t ws_bytes, hspStream_t stream) {
    return orl_global_fwd_impl<float>(feat, idx, B, N, k, kstride, C, fg, argmax, ws, ws_bytes, stream);
}
extern "C" int hsp_orl_global_fwd_bf16(const hsp_bf16_t* feat, const int32_t* idx, int B, int N, int k, int kstride, int C,
                                       float* fg, uint8_t* argmax, void* ws, size_t ws_bytes, hspStream_t stream) {
    return orl_global_fwd_impl<bf16_t>(feat, idx, B, N, k, kstride, C, fg, argmax, ws, ws_bytes, stream);
}

extern "C" int hsp_colsum_rows(const float* x, int B, int N, int C, float* out, void* ws, size_t ws_bytes,
                               hspStream_t stream) {
    if (!x || !out || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (!colsum_vec4(C) && C > 256) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_orl_workspace_bytes(B, N, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int rows = chunk_rows(B, N, C);
    const int nchunk = (N + rows - 1) / rows;
    float* part = reinterpret_cast<float*>(ws);
    if (colsum_vec4(C))
        hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(nchunk, B), dim3(256), 0, st, x, N, C, part, nchunk, rows);
    else
        hipLaunchKernelGGL(colsum_partial_scalar_kernel, dim3(nchunk, B), dim3(256), 0, st, x, N, C, part, nchunk, rows);
    hipLaunchKernelGGL(chunk_fold_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, nchunk, C, 1.0f, out);
    return check_launch();
}

/* out4 (B, 4, C): slot 0 = sum_i x[b][i][:] (the colsum of hsp_colsum_rows), slots 1..3 = sum_i x[b][i][:] * xyz[b][i][0..2]
 * (per-cloud halves of the STE weight gradient g^T xyz of HSlayer_surface, gcn3d.py:85).  C a multiple of 4 with C/4 | 256;
 * ws >= 4 * hsp_orl_workspace_bytes(B, N, C) */
template <typename FT>
static int colsum_rows_xyz_impl(const FT* x, const float* xyz, int B, int N, int C, float* out4, void* ws, size_t ws_bytes,
                                hspStream_t stream) {
    if (!x || !xyz || !out4 || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (!colsum_vec4(C)) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < 4 * hsp_orl_workspace_bytes(B, N, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int rows = chunk_rows(B, N, C);
    const int nchunk = (N + rows - 1) / rows;
    float* part = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(colsum_xyz_partial_kernel<FT>, dim3(nchunk, B), dim3(256), 0, st, x, xyz, N, C, part, nchunk, rows);
    hipLaunchKernelGGL(chunk_fold_kernel, dim3((B * 4 * C + 255) / 256), dim3(256), 0, st, part, B, nchunk, 4 * C, 1.0f, out4);
    return check_launch();
}
extern "C" int hsp_colsum_rows_xyz(const float* x, const float* xyz, int B, int N, int C, float* out4, void* ws, size_t ws_bytes,
                                   hspStream_t stream) {
    return colsum_rows_xyz_impl<float>(x, xyz, B, N, C, out4, ws, ws_bytes, stream);
}
extern "C" int hsp_colsum_rows_xyz_bf16(const hsp_bf16_t* x, const float* xyz, int B, int N, int C, float* out4, void* ws,
                                        size_t ws_bytes, hspStream_t stream) {
    return colsum_rows_xyz_impl<bf16_t>(x, xyz, B, N, C, out4, ws, ws_bytes, stream);
}

extern "C" int hsp_colsum_rows_bf16(const hsp_bf16_t* x, int B, int N, int C, float* out, void* ws, size_t ws_bytes,
                                    hspStream_t stream) {
    if (!x || !out || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (!colsum_vec4(C)) return HSP_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < hsp_orl_workspace_bytes(B, N, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int rows = chunk_rows(B, N, C);
    const int nchunk = (N + rows - 1) / rows;
    float* part = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(nchunk, B), dim3(256), 0, st, x, N, C, part, nchunk, rows);
    hipLaunchKernelGGL(chunk_fold_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, nchunk, C, 1.0f, out);
    return check_launch();
}

template <typename FT>
static int concat_rows_impl(int nseg, const void* const* src, const int32_t* const* idx, const int* width,
                            const int* kind, const int* nsrc, int B, int N, FT* out, int out_pitch, hspStream_t stream) {
    if (nseg <= 0 || nseg > 8 || !src || !width || !kind || !out || B <= 0 || N <= 0) return HSP_ERR_BAD_ARG;
    ConcatDesc d;
    d.nseg = nseg;
    int col = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!src[s] || width[s] <= 0 || kind[s] < 0 || kind[s] > 3 || (kind[s] == 1 && (!idx || !idx[s]))) return HSP_ERR_BAD_ARG;
        d.seg[s].src = src[s];
        d.seg[s].idx = idx ? idx[s] : nullptr;
        d.seg[s].width = width[s];
        d.seg[s].kind = kind[s];
        d.seg[s].nsrc = nsrc ? nsrc[s] : 0;
        d.seg[s].col0 = col;
        col += width[s];
    }
    if (out_pitch < col) return HSP_ERR_BAD_ARG;
    d.even = (out_pitch & 1) ? 0 : 1;
    for (int s = 0; s < nseg; ++s)
        if (kind[s] < 2 && ((width[s] & 1) || (d.seg[s].col0 & 1) || (reinterpret_cast<uintptr_t>(src[s]) & (2 * sizeof(FT) - 1))))
            d.even = 0;
    if (reinterpret_cast<uintptr_t>(out) & (2 * sizeof(FT) - 1)) d.even = 0;
    if (d.even) {               // 16-byte form: pitch, column offsets, widths and bases all multiples of 16 bytes
        constexpr int EP = 16 / (int)sizeof(FT);
        bool q = (out_pitch % EP) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        for (int s = 0; s < nseg && q; ++s)
            if (kind[s] < 2 && ((width[s] % EP) || (d.seg[s].col0 % EP) || (reinterpret_cast<uintptr_t>(src[s]) & 15))) q = false;
        if (q) {
            d.even = 2;
            int acc = 0;
            for (int s = 0; s < nseg; ++s) { d.q0[s] = acc; acc += kind[s] >= 2 ? width[s] : width[s] / EP; }
            d.q0[nseg] = acc;
        }
    }
    const long long rows = (long long)B * N;
    int grid = (int)(rows < 8192 ? rows : 8192);
    if (d.even == 2) grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipLaunchKernelGGL(concat_rows_kernel<FT>, dim3(grid), dim3(256), 0, as_stream(stream), d, B, N, out_pitch, out);
    return check_launch();
}
extern "C" int hsp_concat_rows(int nseg, const float* const* src, const int32_t* const* idx, const int* width,
                               const int* kind, const int* nsrc, int B, int N, float* out, hspStream_t stream) {
    int col = 0;
    for (int s = 0; s < nseg && s < 8 && width; ++s) col += width[s];
    return concat_rows_impl<float>(nseg, reinterpret_cast<const void* const*>(src), idx, width, kind, nsrc, B, N, out, col, stream);
}
/* the same with an explicit output row pitch (>= sum of widths: padding columns are zeroed when every segment is 16-byte
 * aligned -- the form the feat assembly takes -- and left untouched otherwise) and, in the bf16
 * form, bf16 kind-0 / kind-1 sources with fp32 kind-2 (per-cloud) sources */
extern "C" int hsp_concat_rows_pitched(int nseg, const float* const* src, const int32_t* const* idx, const int* width,
                                       const int* kind, const int* nsrc, int B, int N, float* out, int out_pitch,
                                       hspStream_t stream) {
    return concat_rows_impl<float>(nseg, reinterpret_cast<const void* const*>(src), idx, width, kind, nsrc, B, N, out, out_pitch, stream);
}
extern "C" int hsp_concat_rows_bf16(int nseg, const void* const* src, const int32_t* const* idx, const int* width,
                                    const int* kind, const int* nsrc, int B, int N, hsp_bf16_t* out, int out_pitch,
                                    hspStream_t stream) {
    return concat_rows_impl<bf16_t>(nseg, src, idx, width, kind, nsrc, B, N, out, out_pitch, stream);
}

/* out (R, C) = (ga + gb) * [y > 0]: relu backward fused with the gradient sum of a two-consumer tensor (FaceRecon.py:88);
 * ga / gb: rows of pitch lda / ldb (even, 8-byte aligned bases), gb may be NULL; y, out dense; C even */
extern "C" int hsp_add_relu_bwd(const float* ga, int lda, const float* gb, int ldb, const float* y, int R, int C, float* out,
                                hspStream_t stream) {
    if (!ga || !y || !out || R <= 0 || C <= 0 || lda < C || (gb && ldb < C)) return HSP_ERR_BAD_ARG;
    if ((C & 1) || (lda & 1) || (gb && (ldb & 1)) || (reinterpret_cast<size_t>(ga) & 7) || (reinterpret_cast<size_t>(gb) & 7))
        return HSP_ERR_UNSUPPORTED;
    const long long total2 = (long long)R * (C >> 1);
    hipLaunchKernelGGL(add_relu_bwd_kernel, dim3(stream_grid(total2)), dim3(256), 0, as_stream(stream), ga, lda, gb, ldb, y, total2,
                       C >> 1, out);
    return check_launch();
}

extern "C" int hsp_residual_bias(float* out, const float* f, const float* t, int B, int N, int C, hspStream_t stream) {
    if (!out || !f || !t || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (C & 3) return HSP_ERR_UNSUPPORTED;
    const long long total4 = (long long)B * N * (C >> 2);
    hipLaunchKernelGGL(residual_bias_kernel, dim3(stream_grid(total4)), dim3(256), 0, as_stream(stream), out, f, t, total4,
                       N, C);
    return check_launch();
}
