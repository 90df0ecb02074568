/*
 * hsp_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped, never timed as product).
 *
 * Plain-C restatement of the integer/index-producing parts of the HS-Pose hot path, written
 * from the reference's behaviour (paths relative to /root/reference):
 *   - KNN over xyz / feature rows      network/fs_net_repo/gcn3d.py:15-24   (get_neighbor_index)
 *   - top-1 nearest source point       network/fs_net_repo/gcn3d.py:27-36   (get_nearest_index)
 *   - Chamfer nn-search fwd / bwd      tools/pyTorchChamferDistance/chamfer_distance.cpp:59-87, :114-177
 *   - farthest point sampling          tools/eval_utils.py:73-84, :107-119
 *
 * The reference computes distances with ATen CPU ops.  Their fp32 summation orders were pinned
 * in this container (torch 2.10.0 CPU, MKL sgemm, AVX512 build) and are restated here exactly:
 *   inner[i][j]  = k-ordered fmaf chain starting from 0        (== torch.bmm, bit for bit, K=3..256)
 *   quad[i]      = ATen row-sum order: C<8 -> 4 interleaved scalar partials; C>=8 -> 8-lane
 *                  vector partials (x4 ILP, cascade levels), then a sequential horizontal sum
 *   dist[i][j]   = ((inner * -2) + quad[j]) + quad[i]           (gcn3d.py:21, left to right)
 *   nn1 d[i][j]  = (s_norm[j] + t_norm[i]) - (2 * inner)        (gcn3d.py:34)
 * Selection: the k+1 smallest by (distance, index) lexicographic order -- i.e. ascending
 * distance, LOWEST INDEX FIRST on exact ties (torch.topk leaves tie order unspecified; this is
 * the rule the HIP kernels implement) -- then rank 0 is dropped (gcn3d.py:22-23), not "self".
 *
 * Parity status: pinned against the imported reference by oracle/gen_golden.py (fixtures under
 * tests/golden/), see tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int ceil_log2_i64(int64_t x) {
    int r = 0;
    int64_t v = 1;
    if (x <= 1) return 0;
    while (v < x) { v <<= 1; r++; }
    return r;
}

/* ATen multi_row_sum<acc, 4>: cascade sum over "rows" of 4 interleaved lanes-of-W accumulators.
 * data is viewed as size rows, each row = 4 groups of W floats (row stride 4*W, group stride W). */
#define ORL_MAXW 8
static void aten_multi_row_sum(const float *data, int64_t size, int W, float out[4][ORL_MAXW]) {
    const int num_levels = 4;
    int level_power = ceil_log2_i64(size) / num_levels;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    float acc[4][4][ORL_MAXW];
    memset(acc, 0, sizeof(acc));
    int64_t i = 0;
    for (; i + level_step <= size;) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < W; l++) acc[0][k][l] += data[(i * 4 + k) * W + l];
        for (int j = 1; j < num_levels; ++j) {
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < W; l++) { acc[j][k][l] += acc[j - 1][k][l]; acc[j - 1][k][l] = 0.f; }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < W; l++) acc[0][k][l] += data[(i * 4 + k) * W + l];
    for (int j = 1; j < num_levels; ++j)
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < W; l++) acc[0][k][l] += acc[j][k][l];
    for (int k = 0; k < 4; k++)
        for (int l = 0; l < W; l++) out[k][l] = acc[0][k][l];
}

/* row sum of one row of C already-squared floats in ATen's CPU order */
static float aten_row_sum(const float *sq, int C) {
    float ps[4][ORL_MAXW];
    if (C < 8) { /* scalar_inner_sum: W = 1 */
        int64_t size_ilp = C / 4;
        aten_multi_row_sum(sq, size_ilp, 1, ps);
        for (int64_t i = size_ilp * 4; i < C; i++) ps[0][0] += sq[i];
        for (int k = 1; k < 4; k++) ps[0][0] += ps[k][0];
        return ps[0][0];
    }
    const int W = 8;
    int64_t vec_size = C / W;
    int64_t size_ilp = vec_size / 4;
    aten_multi_row_sum(sq, size_ilp, W, ps);
    for (int64_t m = size_ilp * 4; m < vec_size; m++)
        for (int l = 0; l < W; l++) ps[0][l] += sq[m * W + l];
    for (int k = 1; k < 4; k++)
        for (int l = 0; l < W; l++) ps[0][l] += ps[k][l];
    float fin = 0.f;
    for (int64_t k = vec_size * W; k < C; k++) fin += sq[k];
    for (int l = 0; l < W; l++) fin += ps[0][l];
    return fin;
}

/* quad[r] = sum_c x[r][c]^2 in the reference's order (gcn3d.py:20 / :32-33) */
void hsp_oracle_quad(const float *x, int64_t rows, int C, float *quad) {
    float *sq = (float *)malloc(sizeof(float) * (size_t)C);
    for (int64_t r = 0; r < rows; r++) {
        for (int c = 0; c < C; c++) {
            volatile float p = x[r * C + c] * x[r * C + c];
            sq[c] = p;
        }
        quad[r] = aten_row_sum(sq, C);
    }
    free(sq);
}

static inline float dot_chain(const float *a, const float *b, int C) {
    float acc = 0.f;
    for (int c = 0; c < C; c++) acc = fmaf(a[c], b[c], acc);
    return acc;
}

/* distance matrix row i of cloud x (N,C): d[j] = ((inner*-2) + quad[j]) + quad[i] */
static void knn_dist_row(const float *x, const float *quad, int N, int C, int i, float *d) {
    for (int j = 0; j < N; j++) {
        float inner = dot_chain(x + (size_t)i * C, x + (size_t)j * C, C);
        volatile float t1 = inner * -2.0f;
        volatile float t2 = t1 + quad[j];
        volatile float t3 = t2 + quad[i];
        d[j] = t3;
    }
}

/* select the m smallest of d[0..N) by (d, index); result ascending in sel[0..m) */
static void select_smallest(const float *d, int N, int m, int32_t *sel) {
    float *bd = (float *)malloc(sizeof(float) * (size_t)m);
    int cnt = 0;
    for (int j = 0; j < N; j++) {
        float v = d[j];
        if (cnt == m && !(v < bd[m - 1])) continue; /* strict <: earlier (lower) index wins ties */
        int p = (cnt < m) ? cnt : m - 1;
        while (p > 0 && v < bd[p - 1]) { bd[p] = bd[p - 1]; sel[p] = sel[p - 1]; p--; }
        bd[p] = v; sel[p] = j;
        if (cnt < m) cnt++;
    }
    free(bd);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * torch.topk(dist, m, largest=False) on the CPU, INCLUDING its order among exactly equal distances.
 * ATen (aten/src/ATen/native/cpu/TopKImpl.h) fills a queue of (value, index) pairs and, for m * 64 > N, calls
 *   std::nth_element(q, q + m - 1, q + N, cmp);  std::sort(q, q + m - 1, cmp);      cmp(x, y) = x.value < y.value  (no NaNs here)
 * and for m * 64 <= N  std::partial_sort(q, q + m, q + N, cmp).  The comparator never looks at the index, so which of several
 * equal distances lands where is decided by libstdc++'s algorithms -- deterministic, restated below line by line
 * (bits/stl_algo.h, bits/stl_heap.h: introselect / introsort with median-of-three to first, unguarded partition, insertion
 * sorts with threshold 16, heap select).  Pinned against torch.topk itself on tie-rich rows (oracle/gen_golden_exact.py,
 * tests/golden/exact_topk_ties.npz).
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct { float v; int32_t i; } tk_t;
#define TK_LT(a, b) ((a).v < (b).v)
static inline void tk_swap(tk_t *a, tk_t *b) { tk_t t = *a; *a = *b; *b = t; }
static int tk_lg(long n) { int k = 0; while (n > 1) { n >>= 1; k++; } return k; }

static void tk_move_median_to_first(tk_t *result, tk_t *a, tk_t *b, tk_t *c) {
    if (TK_LT(*a, *b)) {
        if (TK_LT(*b, *c)) tk_swap(result, b);
        else if (TK_LT(*a, *c)) tk_swap(result, c);
        else tk_swap(result, a);
    } else if (TK_LT(*a, *c)) tk_swap(result, a);
    else if (TK_LT(*b, *c)) tk_swap(result, c);
    else tk_swap(result, b);
}
static tk_t *tk_unguarded_partition(tk_t *first, tk_t *last, tk_t *pivot) {
    for (;;) {
        while (TK_LT(*first, *pivot)) ++first;
        --last;
        while (TK_LT(*pivot, *last)) --last;
        if (!(first < last)) return first;
        tk_swap(first, last);
        ++first;
    }
}
static tk_t *tk_partition_pivot(tk_t *first, tk_t *last) {
    tk_t *mid = first + (last - first) / 2;
    tk_move_median_to_first(first, first + 1, mid, last - 1);
    return tk_unguarded_partition(first + 1, last, first);
}
static void tk_unguarded_linear_insert(tk_t *last) {
    tk_t val = *last;
    tk_t *next = last - 1;
    while (TK_LT(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void tk_insertion_sort(tk_t *first, tk_t *last) {
    if (first == last) return;
    for (tk_t *i = first + 1; i != last; ++i) {
        if (TK_LT(*i, *first)) {
            tk_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(tk_t));
            *first = val;
        } else tk_unguarded_linear_insert(i);
    }
}
/* heap primitives (max-heap under TK_LT), as bits/stl_heap.h */
static void tk_push_heap(tk_t *first, long hole, long top, tk_t value) {
    long parent = (hole - 1) / 2;
    while (hole > top && TK_LT(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void tk_adjust_heap(tk_t *first, long hole, long len, tk_t value) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (TK_LT(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    tk_push_heap(first, hole, top, value);
}
static void tk_make_heap(tk_t *first, tk_t *last) {
    const long len = last - first;
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        tk_t value = first[parent];
        tk_adjust_heap(first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}
static void tk_pop_heap(tk_t *first, tk_t *last, tk_t *result) {
    tk_t value = *result;
    *result = *first;
    tk_adjust_heap(first, 0, last - first, value);
}
static void tk_heap_select(tk_t *first, tk_t *middle, tk_t *last) {
    tk_make_heap(first, middle);
    for (tk_t *i = middle; i < last; ++i)
        if (TK_LT(*i, *first)) tk_pop_heap(first, middle, i);
}
static void tk_sort_heap(tk_t *first, tk_t *last) {
    while (last - first > 1) { --last; tk_pop_heap(first, last, last); }
}
static void tk_introselect(tk_t *first, tk_t *nth, tk_t *last, int depth_limit) {
    while (last - first > 3) {
        if (depth_limit == 0) {
            tk_heap_select(first, nth + 1, last);
            tk_swap(first, nth);
            return;
        }
        --depth_limit;
        tk_t *cut = tk_partition_pivot(first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    tk_insertion_sort(first, last);
}
static void tk_introsort_loop(tk_t *first, tk_t *last, int depth_limit) {
    while (last - first > 16) {
        if (depth_limit == 0) {                        /* std::__partial_sort(first, last, last) */
            tk_heap_select(first, last, last);
            tk_sort_heap(first, last);
            return;
        }
        --depth_limit;
        tk_t *cut = tk_partition_pivot(first, last);
        tk_introsort_loop(cut, last, depth_limit);
        last = cut;
    }
}
static void tk_sort(tk_t *first, tk_t *last) {
    if (first == last) return;
    tk_introsort_loop(first, last, tk_lg(last - first) * 2);
    if (last - first > 16) {
        tk_insertion_sort(first, first + 16);
        for (tk_t *i = first + 16; i != last; ++i) tk_unguarded_linear_insert(i);
    } else tk_insertion_sort(first, last);
}
/* sel[0..m) = indices torch.topk(d, m, largest=False, sorted=True) returns on the CPU */
static void topk_smallest_aten(const float *d, int N, int m, int32_t *sel, tk_t *q) {
    for (int j = 0; j < N; j++) { q[j].v = d[j]; q[j].i = j; }
    if ((long)m * 64 <= N) {                           /* std::partial_sort */
        tk_heap_select(q, q + m, q + N);
        tk_sort_heap(q, q + m);
    } else {
        if (q + (m - 1) != q + N) tk_introselect(q, q + (m - 1), q + N, tk_lg(N) * 2);
        tk_sort(q, q + (m - 1));
    }
    for (int j = 0; j < m; j++) sel[j] = q[j].i;
}
/* stand-alone form for the fixture generator / tests: rows (R,N) of distances -> (R,m) indices */
void hsp_oracle_topk_smallest(const float *d, int R, int N, int m, int32_t *sel) {
    tk_t *q = (tk_t *)malloc(sizeof(tk_t) * (size_t)N);
    for (int r = 0; r < R; r++) topk_smallest_aten(d + (size_t)r * N, N, m, sel + (size_t)r * m, q);
    free(q);
}

/* get_neighbor_index (gcn3d.py:15-24).  x (B,N,C) fp32 row-major -> idx (B,N,k) int32.
 * drop_first=1 reproduces the reference ([:, :, 1:] after topk(k+1)). Also returns (optional,
 * may be NULL) the selected distances dsel (B,N,k) for near-tie diagnostics. */
int hsp_oracle_knn(const float *x, int B, int N, int C, int k, int drop_first, int32_t *idx, float *dsel) {
    int m = k + (drop_first ? 1 : 0);
    if (m > N || k <= 0) return -1;
    float *quad = (float *)malloc(sizeof(float) * (size_t)N);
    float *d = (float *)malloc(sizeof(float) * (size_t)N);
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    for (int b = 0; b < B; b++) {
        const float *xb = x + (size_t)b * N * C;
        hsp_oracle_quad(xb, N, C, quad);
        for (int i = 0; i < N; i++) {
            knn_dist_row(xb, quad, N, C, i, d);
            select_smallest(d, N, m, sel);
            for (int r = 0; r < k; r++) {
                int s = sel[r + (drop_first ? 1 : 0)];
                idx[((size_t)b * N + i) * k + r] = s;
                if (dsel) dsel[((size_t)b * N + i) * k + r] = d[s];
            }
        }
    }
    free(quad); free(d); free(sel);
    return 0;
}

/* the same search with torch.topk's OWN order among exactly equal distances (ATen TopKImpl.h -> libstdc++, restated above):
 * what the reference returns on tiled clouds (datasets/load_data.py:314-316), where duplicates make ties the normal case.
 * Pinned against the imported reference by tests/golden/exact_stack_tiled_1028.npz (tests/test_oracle_golden.py). */
int hsp_oracle_knn_topk(const float *x, int B, int N, int C, int k, int drop_first, int32_t *idx) {
    int m = k + (drop_first ? 1 : 0);
    if (m > N || k <= 0) return -1;
    float *quad = (float *)malloc(sizeof(float) * (size_t)N);
    float *d = (float *)malloc(sizeof(float) * (size_t)N);
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)m);
    tk_t *q = (tk_t *)malloc(sizeof(tk_t) * (size_t)N);
    for (int b = 0; b < B; b++) {
        const float *xb = x + (size_t)b * N * C;
        hsp_oracle_quad(xb, N, C, quad);
        for (int i = 0; i < N; i++) {
            knn_dist_row(xb, quad, N, C, i, d);
            topk_smallest_aten(d, N, m, sel, q);
            for (int r = 0; r < k; r++) idx[((size_t)b * N + i) * k + r] = sel[r + (drop_first ? 1 : 0)];
        }
    }
    free(quad); free(d); free(sel); free(q);
    return 0;
}

/* get_nearest_index (gcn3d.py:27-36). tgt (B,Nt,C), src (B,Ns,C) -> idx (B,Nt) int32 */
int hsp_oracle_nn1(const float *tgt, int Nt, const float *src, int Ns, int B, int C, int32_t *idx) {
    float *sq = (float *)malloc(sizeof(float) * (size_t)Ns);
    float *tq = (float *)malloc(sizeof(float) * (size_t)Nt);
    for (int b = 0; b < B; b++) {
        const float *t = tgt + (size_t)b * Nt * C;
        const float *s = src + (size_t)b * Ns * C;
        hsp_oracle_quad(s, Ns, C, sq);
        hsp_oracle_quad(t, Nt, C, tq);
        for (int i = 0; i < Nt; i++) {
            float best = 0.f; int bi = 0;
            for (int j = 0; j < Ns; j++) {
                float inner = dot_chain(t + (size_t)i * C, s + (size_t)j * C, C);
                volatile float a = sq[j] + tq[i];
                volatile float two = 2.0f * inner;
                volatile float dd = a - two;
                if (j == 0 || dd < best) { best = dd; bi = j; }
            }
            idx[(size_t)b * Nt + i] = bi;
        }
    }
    free(sq); free(tq);
    return 0;
}

/* Chamfer nn search, one direction (chamfer_distance.cpp:59-87): differences first, fp32
 * products and sums (the reference's double `d` only widens an fp32 expression), strict <. */
static void chamfer_nn(int b, int n, int m, const float *a, const float *c, float *dist, int32_t *idx) {
    for (int i = 0; i < b; i++)
        for (int j = 0; j < n; j++) {
            const float x1 = a[(i * n + j) * 3 + 0], y1 = a[(i * n + j) * 3 + 1], z1 = a[(i * n + j) * 3 + 2];
            float best = 0.f; int besti = 0;
            for (int k = 0; k < m; k++) {
                volatile float x2 = c[(i * m + k) * 3 + 0] - x1;
                volatile float y2 = c[(i * m + k) * 3 + 1] - y1;
                volatile float z2 = c[(i * m + k) * 3 + 2] - z1;
                volatile float xx = x2 * x2, yy = y2 * y2, zz = z2 * z2;
                volatile float s1 = xx + yy;
                volatile float d = s1 + zz;
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[i * n + j] = best;
            idx[i * n + j] = besti;
        }
}

void hsp_oracle_chamfer_fwd(const float *x1, const float *x2, int B, int n, int m,
                            float *dist1, float *dist2, int32_t *idx1, int32_t *idx2) {
    chamfer_nn(B, n, m, x1, x2, dist1, idx1);
    chamfer_nn(B, m, n, x2, x1, dist2, idx2);
}

/* chamfer_distance.cpp:114-177 -- serial scatter, fp32 */
void hsp_oracle_chamfer_bwd(const float *x1, const float *x2, const int32_t *idx1, const int32_t *idx2,
                            const float *gd1, const float *gd2, int B, int n, int m, float *gx1, float *gx2) {
    memset(gx1, 0, sizeof(float) * (size_t)B * n * 3);
    memset(gx2, 0, sizeof(float) * (size_t)B * m * 3);
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < n; j++) {
            int j2 = idx1[i * n + j];
            float g = gd1[i * n + j] * 2;
            for (int d = 0; d < 3; d++) {
                float df = x1[(i * n + j) * 3 + d] - x2[(i * m + j2) * 3 + d];
                gx1[(i * n + j) * 3 + d] += g * df;
                gx2[(i * m + j2) * 3 + d] -= g * df;
            }
        }
        for (int j = 0; j < m; j++) {
            int j2 = idx2[i * m + j];
            float g = gd2[i * m + j] * 2;
            for (int d = 0; d < 3; d++) {
                float df = x2[(i * m + j) * 3 + d] - x1[(i * n + j2) * 3 + d];
                gx2[(i * m + j) * 3 + d] += g * df;
                gx1[(i * n + j2) * 3 + d] -= g * df;
            }
        }
    }
}

/* Farthest point sampling (tools/eval_utils.py:107-119): float64 Euclidean (sqrt) distances
 * (eval_utils.py:73-84), start at index 0, running min, argmax takes the FIRST maximum.
 * pts (B,N,3) double -> sel (B,n_samples) int32.  Batched over B clouds independently. */
void hsp_oracle_fps_f64(const double *pts, int B, int N, int n_samples, int32_t *sel) {
    double *dts = (double *)malloc(sizeof(double) * (size_t)N);
    for (int b = 0; b < B; b++) {
        const double *p = pts + (size_t)b * N * 3;
        int cur = 0;
        for (int j = 0; j < N; j++) {
            double dx = p[j * 3] - p[0], dy = p[j * 3 + 1] - p[1], dz = p[j * 3 + 2] - p[2];
            dts[j] = sqrt(dx * dx + dy * dy + dz * dz);
        }
        for (int s = 0; s < n_samples; s++) {
            sel[(size_t)b * n_samples + s] = cur;
            int arg = 0; double best = -1.0;
            for (int j = 0; j < N; j++) {
                double dx = p[j * 3] - p[cur * 3], dy = p[j * 3 + 1] - p[cur * 3 + 1], dz = p[j * 3 + 2] - p[cur * 3 + 2];
                double d = sqrt(dx * dx + dy * dy + dz * dz);
                if (d < dts[j]) dts[j] = d;
                if (dts[j] > best) { best = dts[j]; arg = j; }
            }
            cur = arg;
        }
    }
    free(dts);
}

/* The same helper called on a float32 array: numpy then computes diff, diff**2, the sum over the three coordinates
 * ((x*x + y*y) + z*z, sequential) and np.sqrt all in fp32 (eval_utils.py:73-84).  The square root matters: it maps
 * neighbouring fp32 values of d^2 to the SAME fp32 distance about half the time, and np.argmax then takes the first of
 * them -- a squared-distance rule picks a different point there (pinned by tests/golden/fps_*: the lattice cloud). */
void hsp_oracle_fps_f32(const float *pts, int B, int N, int n_samples, int32_t *sel) {
    float *dts = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; b++) {
        const float *p = pts + (size_t)b * N * 3;
        int cur = 0;
        for (int j = 0; j < N; j++) dts[j] = INFINITY;
        for (int s = 0; s < n_samples; s++) {
            sel[(size_t)b * n_samples + s] = cur;
            int arg = 0; float best = -1.0f;
            for (int j = 0; j < N; j++) {
                volatile float dx = p[j * 3] - p[cur * 3], dy = p[j * 3 + 1] - p[cur * 3 + 1], dz = p[j * 3 + 2] - p[cur * 3 + 2];
                volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                volatile float s1 = xx + yy;
                volatile float d2 = s1 + zz;
                volatile float d = sqrtf(d2);
                if (d < dts[j]) dts[j] = d;
                if (dts[j] > best) { best = dts[j]; arg = j; }
            }
            cur = arg;
        }
    }
    free(dts);
}
