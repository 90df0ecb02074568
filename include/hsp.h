/*
 * hsp.h -- C-ABI of libhsp.so: the MI355X (gfx950) hybrid-scope point-cloud feature-extractor kernels.
 *
 * This is the drop-in boundary for the HS-Pose hot path.  The reference has no
 * C/FFI boundary for its live path -- the boundary there is the Python module API of
 * network/fs_net_repo/gcn3d.py -- so every entry point below cites the reference Python function
 * whose device work it replaces.  The only FFI the reference does have is the pybind module of its
 * (dead) Chamfer extension, tools/pyTorchChamferDistance/chamfer_distance.cpp:180-185; hsp_chamfer_*
 * keep that module's argument roles.
 *
 * Conventions
 *   - every pointer is DEVICE memory, row-major contiguous, fp32 / int32 / uint8 as typed;
 *   - the caller owns every buffer; kernels allocate nothing.  Ops that need scratch take a
 *     caller-supplied workspace (`ws`, `ws_bytes`) sized by the matching *_workspace_bytes() query;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is stream-ordered
 *     and re-entrant, safe to capture into a hipGraph;
 *   - return value: 0 = HSP_OK, negative = error code (see hsp_error_string); nothing throws;
 *   - index tensors are int32 (the Python mirror widens to int64 at its edge because the reference
 *     API returns int64, gcn3d.py:22-23);
 *   - "cloud" = one (N,3) point set of the batch; B clouds are independent in every op.
 */
#ifndef HSP_H_
#define HSP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSP_OK 0
#define HSP_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, k out of range ...       */
#define HSP_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels were built for             */
#define HSP_ERR_WORKSPACE (-3)    /* ws == NULL or ws_bytes too small                          */
#define HSP_ERR_LAUNCH (-4)       /* hipLaunch / hipMemsetAsync reported an error              */

#define HSP_MAX_K 32              /* largest neighbour count (k + drop_first <= 33)            */

typedef void *hspStream_t;
typedef uint16_t hsp_bf16_t;      /* bfloat16 bits (the upper half of an fp32), for the *_bf16 entry points */

int hsp_version(void);
const char *hsp_error_string(int code);
/* last HIP error text recorded by a failing launch on this thread ("" if none) */
const char *hsp_last_hip_error(void);

/* ---- neighbour search ------------------------------------------------------------------------
 * replaces get_neighbor_index(vertices, k)            network/fs_net_repo/gcn3d.py:15-24
 * x (B,N,C) -> idx (B,N,k): the k rows nearest to each row under the reference's expanded fp32
 * distance ((inner*-2)+quad[j])+quad[i], inner = k-ordered fma chain (== torch.bmm on CPU), quad in
 * ATen's row-sum order; ascending distance, lowest index first on exact ties; with drop_first=1 the
 * rank-0 entry of the (k+1) nearest is dropped (the reference's [:, :, 1:]), which is NOT "exclude
 * self".  C==3 runs the LDS-resident xyz kernel, any other C the f32-MFMA distance-tile kernel.
 */
size_t hsp_knn_workspace_bytes(int B, int N, int C, int k);
int hsp_knn_f32(const float *x, int B, int N, int C, int k, int drop_first, int32_t *idx,
                void *ws, size_t ws_bytes, hspStream_t stream);
/* get_neighbor_index with torch.topk's OWN order among exactly equal distances (ATen's CPU topk = libstdc++'s
 * nth_element + sort, or partial_sort for (k + drop_first) * 64 <= N, under a comparator that only sees the value: restated in
 * csrc/knn_exact.hip and oracle/hsp_oracle.c, pinned against torch.topk on tie-rich rows).  hsp_knn_f32 breaks ties by the lowest
 * index; on tie-free rows the two agree.  Used by the eval-mode forward (exact scope).  k + drop_first + 1 <= 33.
 * quad_mode (C != 3): how torch.sum(x ** 2, dim=2) of gcn3d.py:20 rounds -- 0: x is a contiguous (B,N,C) tensor (ATen's
 * vectorised row sum), 1: x is the transposed VIEW of a (B,C,N) tensor, which is what FaceRecon.py:94-95 hands conv_3 (ATen then
 * sums the channels as an outer reduction: cascade for the first 32 floor(N/32) points, four interleaved cascades for the rest).
 * The search flags the rows that hold two equal distances among their k + drop_first + 1 nearest and a fixed-grid pass replays only
 * those (round 5; while B N N floats stay below 512 MB the search also leaves the distance matrix in ws, so a replayed row reads its
 * distances instead of recomputing ~0.5 MB of feature rows).
 * ws: hsp_knn_exact_workspace_bytes; tie_rows (may be NULL): a device int that is incremented once per row that held a tie. */
size_t hsp_knn_exact_workspace_bytes(int B, int N, int C, int k, int drop_first);
int hsp_knn_exact_f32(const float *x, int B, int N, int C, int k, int drop_first, int quad_mode, int32_t *idx, void *ws,
                      size_t ws_bytes, int *tie_rows, hspStream_t stream);
/* get_neighbor_index on COORDINATES (C == 3), torch.topk's order among equal distances included, for up to two list lengths of
 * the same search: idx (B,N,k) and, when k2 > 0, idx2 (B,N,k2), k2 <= k -- the layers' k-list and Pool_layer's 4-list of one
 * resolution (gcn3d.py:236; on a tiled cloud, datasets/load_data.py:314-316, the short list is NOT the prefix of the long one:
 * ATen takes std::partial_sort for (k2 + drop) * 64 <= N and nth_element + sort otherwise).  The search itself is hsp_knn_f32's
 * xyz kernel run one rank past the answer; it flags the rows that hold two equal distances among those k + drop + 1 nearest, and
 * a fixed-grid pass replays only the flagged rows through libstdc++'s routines (csrc/knn_exact.hip).  A tie-free batch pays one
 * small launch.  This is what every xyz search of the package goes through, training included.
 * N <= 10 240 (the replay keeps one row of candidates in LDS; beyond it HSP_ERR_UNSUPPORTED is returned before any launch and the
 * caller falls back on hsp_knn_f32's (distance, index) order).
 * ws: hsp_knn_xyz_workspace_bytes (the row flags); tie_rows (may be NULL): device int, += number of flagged rows -- counted by
 * the SEPARATE replay pass only: for N <= 576 with B N < 131 072 the selection kernel replays its flagged rows inline and leaves
 * tie_rows untouched. */
size_t hsp_knn_xyz_workspace_bytes(int B, int N);
int hsp_knn_xyz_f32(const float *xyz, int B, int N, int k, int k2, int drop_first, int32_t *idx, int32_t *idx2, void *ws,
                    size_t ws_bytes, int *tie_rows, hspStream_t stream);
/* The coordinate work of the stack's two coarse levels in ONE launch (FaceRecon.py:91-101, gcn3d.py:236,243-245).  Pool_layer keeps
 * rows sel1 (N1 of them, int32, device) of the input cloud and then rows sel2 (N2) of that level; the draws are host-side and known
 * before the forward, and everything later asked of the two clouds depends on coordinates only:
 *   v1 (B,N1,3), v2 (B,N2,3)                      the levels' vertices
 *   idx1 (B,N1,k1), idx1_pool (B,N1,kpool)        get_neighbor_index(v1, k1) and Pool_layer's own list (kpool = 0: none, pass NULL)
 *   idx2 (B,N2,k2)                                get_neighbor_index(v2, k2)
 *   up1, up2 (B,N0)                               get_nearest_index(vertices, v1 / v2)
 * Four independent small searches that cost a launch each (6-13 us, mostly latency); here they are block ranges of one grid.  Same
 * results as hsp_knn_xyz_f32 / hsp_nn1_f32 (torch.topk's order among equal distances included).  64 <= N2 <= N1 <= 576, else
 * HSP_ERR_UNSUPPORTED (the caller keeps the separate calls). */
int hsp_geometry_levels_f32(const float *xyz, int B, int N0, const int32_t *sel1, int N1, const int32_t *sel2, int N2, int k1,
                            int kpool, int k2, int drop_first, float *v1, float *v2, int32_t *idx1, int32_t *idx1_pool,
                            int32_t *idx2, int32_t *up1, int32_t *up2, hspStream_t stream);
/* hsp_knn_xyz_f32 on the input cloud (idx0 (B,N0,k0), idx0_pool (B,N0,kpool0)) AND hsp_geometry_levels_f32 in two launches instead of
 * three: the level-0 search's tie pass depends on that search's flags only, so it rides in the levels' launch as a further block
 * range (a flagged row is ~14 us of latency during which the chip would otherwise idle).  576 < N0 <= 1088 and the levels' limits,
 * else HSP_ERR_UNSUPPORTED (the caller keeps the separate calls).  ws: hsp_geometry_all_workspace_bytes (the level-0 flags). */
size_t hsp_geometry_all_workspace_bytes(int B, int N0);
int hsp_geometry_all_f32(const float *xyz, int B, int N0, int k0, int kpool0, const int32_t *sel1, int N1, const int32_t *sel2, int N2,
                         int k1, int kpool, int k2, int drop_first, int32_t *idx0, int32_t *idx0_pool, float *v1, float *v2,
                         int32_t *idx1, int32_t *idx1_pool, int32_t *idx2, int32_t *up1, int32_t *up2, void *ws, size_t ws_bytes,
                         hspStream_t stream);
/* hsp_knn_f32 with the |x|^2 order chosen as above */
int hsp_knn_quadmode_f32(const float *x, int B, int N, int C, int k, int drop_first, int32_t *idx, void *ws, size_t ws_bytes,
                         int quad_mode, hspStream_t stream);
/* |x|^2 per row in the transposed-view order (quad_mode 1) */
int hsp_quad_outer_f32(const float *x, int B, int N, int C, float *quad, hspStream_t stream);
/* PoseNet9D.py:25: centred = pts - mean over the N points, mean (B,3) in ATen's summation order for a contiguous (B,N,3) tensor
 * (an outer sum over 3 columns: four interleaved 16-element cascades per column) divided by N */
int hsp_center_cloud_f32(const float *pts, int B, int N, float *centred, float *mean, hspStream_t stream);

/* replaces get_nearest_index(target, source)          network/fs_net_repo/gcn3d.py:27-36
 * tgt (B,Nt,3), src (B,Ns,3) -> idx (B,Nt): top-1 source row, d = (s2[j]+t2[i]) - 2*inner. */
int hsp_nn1_f32(const float *tgt, int Nt, const float *src, int Ns, int B, int32_t *idx,
                hspStream_t stream);

/* ---- receptive-field graph convolution -------------------------------------------------------
 * replaces HSlayer_surface.graph_conv                 gcn3d.py:92-107   (+ directions of :49-59)
 * xyz (B,N,3), idx (B,N,k), dirs (3, S*K) = the RAW support-direction parameter; the kernels apply
 * F.normalize(dim=0) (gcn3d.py:100,166: D / max(||D||_col, 1e-12)) themselves, and the backward entry
 * points return the gradient w.r.t. the RAW parameter (normalisation Jacobian included).
 * out (B,N,K) = mean_s max_n relu(R[b,i,n,:] . D^[:, s*K+c]);
 * argrow (B,N,S*K) uint16 = the SOURCE ROW idx[b,i,n*] of the winning neighbour (N <= 65535): all a
 * backward needs -- neither idx nor the slot n is read again.
 * R = normalize(xyz[idx]-xyz) is recomputed in-kernel, never materialised.
 */
int hsp_rf_surface_fwd(const float *xyz, const int32_t *idx, const float *dirs, int B, int N, int k,
                       int S, int K, float *out, uint16_t *argrow, hspStream_t stream);
/* grad_dirs (3,S*K) is OVERWRITTEN with d(loss)/d(raw directions).
 * ws: hsp_rf_bwd_scatter_workspace_bytes(B, S*K). */
size_t hsp_rf_bwd_scatter_workspace_bytes(int B, int SC);
int hsp_rf_surface_bwd(const float *xyz, const float *dirs, const uint16_t *argrow, const float *grad_out,
                       int B, int N, int S, int K, float *grad_dirs, void *ws, size_t ws_bytes,
                       hspStream_t stream);

/* replaces HS_layer.graph_conv after its fm GEMM      gcn3d.py:158-181  (gather of :39-47 fused)
 * fm (B,N,(S+1)*C) = feature_map @ weights + bias: columns [0,C) centre, [C+s*C+c] support s.
 * out (B,N,C) = fm[b,i,c] + mean_s max_n relu(R.D^[:,sC+c]) * fm[b, idx[b,i,n], C+sC+c]
 * argrow (B,N,S*C) uint16 as above.  The (B,N,k,S*C) tensors of the reference are never formed.
 * fwin (B,N,S*C) fp32 or NULL: the winner's support value fm[b, argrow[b,i,j], C+j], which
 * hsp_rf_conv_bwd_scatter reads as a stream (inference passes NULL).
 */
int hsp_rf_conv_wants_fwin(int N, int S, int C);
int hsp_rf_conv_fwd(const float *xyz, const int32_t *idx, const float *dirs, const float *fm, int B,
                    int N, int k, int S, int C, float *out, uint16_t *argrow, float *fwin,
                    hspStream_t stream);
/* Backward, COLUMN-TILE LDS-SCATTER form (the default of the Python mirror): a (cloud, 16-column) tile
 * of grad_fm plus the cloud's xyz live in LDS; gradients are routed to row argrow[b,i,j] with ds_add_f32
 * (immune to the in-degree hubs of feature-space graphs) and every row segment is written once.
 * grad_fm (B,N,(S+1)*C) and grad_dirs (3,S*C) are OVERWRITTEN.  The LDS adds make grad_fm
 * order-dependent in the last bits; hsp_rf_conv_bwd is the bit-reproducible twin.
 * fwin: the forward's (B,N,S*C) winner support values (then fm may be NULL), or NULL: the values are
 * gathered from fm (B,N,(S+1)*C).  hsp_rf_conv_wants_fwin(N,S,C) says which is faster (fwin once a cloud's
 * fm outgrows the L2 share it gets).
 * ws: hsp_rf_bwd_scatter_workspace_bytes(B, S*C). */
int hsp_rf_conv_bwd_scatter(const float *xyz, const float *dirs, const float *fm, const float *fwin,
                            const uint16_t *argrow,
                            const float *grad_out, int B, int N, int S, int C, float *grad_fm,
                            float *grad_dirs, void *ws, size_t ws_bytes, hspStream_t stream);
/* Backward, GATHER form over rev_off/rev_edge = hsp_rev_build(idx) of the SAME idx the forward used:
 * every grad_fm row is summed in ascending edge order (no atomics, bit-reproducible).
 * ws: hsp_rf_bwd_workspace_bytes(S*C). */
size_t hsp_rf_bwd_workspace_bytes(int SC);
int hsp_rf_conv_bwd(const float *xyz, const float *dirs, const float *fm, const uint16_t *argrow,
                    const float *grad_out, const int32_t *rev_off, const int32_t *rev_edge, int B, int N,
                    int k, int S, int C, float *grad_fm, float *grad_dirs, void *ws, size_t ws_bytes,
                    hspStream_t stream);

/* ---- reverse-edge (CSR) index of a neighbour graph --------------------------------------------
 * replaces the accumulate-scatter (_index_put_impl_) that autograd runs for the gather of
 * indexing_neighbor_new (gcn3d.py:39-47): idx (B,Nq,kstride), first k columns used ->
 * rev_off (B,Nsrc+1), rev_edge (B,Nq*k): for source row m the ascending edge ids e = i*k + n with
 * idx[b,i,n] == m are rev_edge[b][rev_off[b][m] .. rev_off[b][m+1]).
 */
int hsp_rev_build(const int32_t *idx, int B, int Nq, int Nsrc, int k, int kstride, int32_t *rev_off,
                  int32_t *rev_edge, hspStream_t stream);

/* ---- neighbourhood max-pool (ORL global branch, Pool_layer) ----------------------------------
 * replaces indexing_neighbor_new(...) + max(dim=2)    gcn3d.py:214-216, :236-240
 * feat (B,Nsrc,C); idx (B,Nidx,kstride) of which the first k columns are used; qsel (Nq) optional
 * row selector shared by the batch (Pool_layer's randperm slice, gcn3d.py:243-245; NULL = identity,
 * then Nq must equal Nidx).  out (B,Nq,C) = max_{n<k} feat[b, idx[b, q', n], :], q' = qsel ? qsel[q] : q;
 * argmax (B,Nq,C) uint8.
 */
int hsp_gather_max_fwd(const float *feat, const int32_t *idx, const int32_t *qsel, int B, int Nsrc,
                       int Nidx, int Nq, int k, int kstride, int C, float *out, uint8_t *argmax,
                       hspStream_t stream);
/* grad_feat (B,Nsrc,C) is OVERWRITTEN.  grad_out is (B,Nq,C), or (B,C) broadcast over q when
 * grad_bcast != 0 (the ORL mean-over-points branch; integer counts in LDS => exactly reproducible).
 * Column-tile LDS scatter when a (Nsrc x 16-column) tile fits LDS, else memset + global atomics. */
/* Pool_layer (gcn3d.py:220-246) in ONE launch: out (B,Nq,C) = max over the first k listed neighbours of the kept rows qsel
 * (Nq ints, shared by the batch) + argmax, and xyz_sel (B,Nq,3) = xyz[:, qsel] (the reference's vertices[:, sample_idx]). */
int hsp_pool_fwd(const float *feat, const float *xyz, const int32_t *idx, const int32_t *qsel, int B, int N, int Nq, int k,
                 int kstride, int C, float *out, uint8_t *argmax, float *xyz_sel, hspStream_t stream);
int hsp_gather_max_bwd(const float *grad_out, int grad_bcast, const int32_t *idx, const int32_t *qsel,
                       const uint8_t *argmax, int B, int Nsrc, int Nidx, int Nq, int kstride, int C,
                       float *grad_feat, int accumulate /* !=0: add into grad_feat instead of overwriting */,
                       const float *extra /* optional (B,Nsrc,C) tensor added in the same pass, or NULL */,
                       hspStream_t stream);
/* the same result in GATHER form over hsp_rev_build(idx, k) (qsel == NULL case, Nq rows of idx):
 * each grad_feat row written once, no atomics. */
int hsp_gather_max_bwd_csr(const float *grad_out, int grad_bcast, const uint8_t *argmax, const int32_t *rev_off,
                           const int32_t *rev_edge, int B, int Nsrc, int Nq, int k, int C, float *grad_feat,
                           hspStream_t stream);

/* ---- max over the points of a cloud (the heads' global feature) ------------------------------
 * replaces torch.max(x, 2, keepdim=True)[0]    PoseR.py:30 / :61, PoseTs.py:35, FaceRecon.py:98
 * x (B,N,C) point-major; out (B,C) = max_n x[b,n,c]; argrow (B,C) int32 = the FIRST row attaining
 * it (NaN counts as the maximum, as in ATen).  bwd: grad_x (B,N,C) is OVERWRITTEN with grad_out on
 * the winning row and 0 elsewhere (C % 4 == 0). */
int hsp_points_max_fwd(const float *x, int B, int N, int C, float *out, int32_t *argrow, hspStream_t stream);
int hsp_points_max_bwd(const float *grad_out, const int32_t *argrow, int B, int N, int C, float *grad_x,
                       hspStream_t stream);

/* ORL global feature in one pass (get_ORL_global, gcn3d.py:211-218, before the repeat):
 * fg (B,C) = mean_i max_{n<k} feat[b, idx[b,i,n], :], argmax (B,N,C) uint8; the (B,N,C) max tensor is never
 * written.  idx (B,N,kstride).  ws: hsp_orl_workspace_bytes(B,N,C).  Backward: hsp_gather_max_bwd(grad_bcast=1). */
size_t hsp_orl_workspace_bytes(int B, int N, int C);
int hsp_orl_global_fwd(const float *feat, const int32_t *idx, int B, int N, int k, int kstride, int C,
                       float *fg, uint8_t *argmax, void *ws, size_t ws_bytes, hspStream_t stream);
/* The same feature with the REFERENCE's summation order: ATen's cascade sum over the points (16-row level-0 chunks, levels dumped
 * every 16 / 256 / 4096 rows, remainder last) then a division by N -- torch.mean(dim=1) of gcn3d.py:217 bit for bit
 * (tests/golden/exact_*.npz).  ws: hsp_orl_exact_workspace_bytes(B,N,C).  argmax as above. */
size_t hsp_orl_exact_workspace_bytes(int B, int N, int C);
int hsp_orl_global_exact_f32(const float *feat, const int32_t *idx, int B, int N, int k, int kstride, int C, float *fg,
                             uint8_t *argmax, void *ws, size_t ws_bytes, hspStream_t stream);
/* eval-mode BatchNorm1d of FaceRecon.py:90-95 on point rows x (R,C) in ATen's own operation order (the reference applies it to
 * a transposed view: the generic TensorIterator path): y = ((x - running_mean) * invstd) * weight + bias, every operation
 * rounded to fp32, optionally followed by relu.  invstd (C) = 1 / sqrt(running_var + eps) as the HOST's ATen evaluates it
 * (its sqrt is MKL VML's, not correctly rounded: one channel in ~180 differs), or NULL: the correctly rounded value from
 * running_var.  C % 4 == 0. */
int hsp_bn_eval_f32(const float *x, long long R, int C, const float *running_mean, const float *running_var,
                    const float *invstd, const float *weight, const float *bias, float eps, int relu, float *y,
                    hspStream_t stream);
/* The HS layer's out product in the reference's order of operations (gcn3d.py:111-112 / :185-186 then :90 / :156): every product a
 * k-ordered fp32 fma chain from 0 (what the reference's CPU GEMM runs for K <= 256; K = 512 is two such chains added), additions
 * in the reference's order.  two_chain == 0 (2C <= 256): out = ((chain(F Wa^T) continued by chain(fg[cloud] Wb^T)) + F) + tail;
 * two_chain == 1: out = (((chain(F Wa^T)) + cloud_t[cloud]) + F) + tail with cloud_t (B,C) = chain(fg Wb^T) from
 * hsp_gemm_wave_f32.  tail = ste (M,C) (the STE product) or the K = 3 chain xyz3 . w3 (surface layer; relu allowed).
 * C % 32 == 0, 16-byte aligned rows, rows_per_cloud >= 32. */
int hsp_layer_out_exact_f32(const float *F, int ldf, const float *Wa, int ldwa, const float *fg, int ldfg, const float *Wb,
                            int ldwb, const float *cloud_t, int two_chain, const float *ste, int ldste, const float *xyz3,
                            const float *w3, int relu, int M, int C, int rows_per_cloud, float *out, int ldo,
                            hspStream_t stream);
/* out (B,C) = sum_i x[b,i,:]  (the per-cloud column sum autograd takes of a gradient that was
 * broadcast over the points, e.g. d/d(fg Wb^T)); deterministic two-stage.  ws as above. */
int hsp_colsum_rows(const float *x, int B, int N, int C, float *out, void *ws, size_t ws_bytes,
                    hspStream_t stream);

/* out (B,N,C) += f (B,N,C) + t (B,C) broadcast over the points: the "+ feature" residual and the per-cloud
 * half of conv2 of an HS layer (gcn3d.py:112,186) in one pass. */
/* out (R, C) = (ga + gb) * [y > 0]: the backward of relu(conv_0(...)) (FaceRecon.py:88) fused with the sum of the two gradients
 * that reach fm_0 (conv_1 and the concat); ga / gb rows of an even pitch lda / ldb (8-byte aligned), gb may be NULL; C even */
int hsp_add_relu_bwd(const float *ga, int lda, const float *gb, int ldb, const float *y, int R, int C, float *out,
                     hspStream_t stream);
int hsp_residual_bias(float *out, const float *f, const float *t, int B, int N, int C, hspStream_t stream);

/* ---- feature assembly ---------------------------------------------------------------------------
 * replaces the nearest-up-sampling gathers + one-hot repeat + torch.cat     FaceRecon.py:100-107
 * out (B,N,sum width): column segment s of row (b,i) is
 *   kind 0: src[s][(b*N+i)*w .. ]   kind 1: src[s][(b*nsrc[s] + idx[s][b*N+i])*w ..]   kind 2: src[s][b*w ..]
 *   kind 3: (long) src[s][b] == c ? 1 : 0  -- one-hot columns from the (B) float category ids (FaceRecon.py:80-85)
 * (host arrays of nseg <= 8 entries; device pointers inside).  The backward of kind-1 segments is
 * hsp_gather_rows_bwd with grad_stride = total width.
 */
int hsp_concat_rows(int nseg, const float *const *src, const int32_t *const *idx, const int *width,
                    const int *kind, const int *nsrc, int B, int N, float *out, hspStream_t stream);

/* ---- row gather (nearest up-sample, vertex select) -------------------------------------------
 * replaces indexing_neighbor_new(t, nearest).squeeze(2)    FaceRecon.py:102-104 ; vertices[:, sample_idx]
 * feat (B,Nsrc,C); idx (B,Nq) or, when idx_shared != 0, (Nq) shared by the batch.
 * out rows are written at out + (b*Nq+q)*out_stride (out_stride >= C lets the caller write straight
 * into a slice of the concatenated feature tensor, FaceRecon.py:107).
 */
int hsp_gather_rows_fwd(const float *feat, const int32_t *idx, int idx_shared, int B, int Nsrc, int Nq,
                        int C, float *out, int out_stride, hspStream_t stream);
/* grad_feat (B,Nsrc,C) OVERWRITTEN; grad_out rows at grad_out + (b*Nq+q)*grad_stride. */
int hsp_gather_rows_bwd(const float *grad_out, int grad_stride, const int32_t *idx, int idx_shared, int B,
                        int Nsrc, int Nq, int C, float *grad_feat, hspStream_t stream);
/* the same backward in gather form over (rev_off, rev_edge) = hsp_rev_build(idx as (B,Nq,1), k = 1): every source row
 * sums its queries' gradient rows in ascending query order (no atomics, bit-reproducible), whole row segments at
 * a time -- the faster form when grad_out is a column block of a much wider tensor (the 1286-wide feature). */
int hsp_gather_rows_bwd_csr(const float *grad_out, int grad_stride, const int32_t *rev_off, const int32_t *rev_edge,
                            int B, int Nsrc, int Nq, int C, float *grad_feat, hspStream_t stream);

/* ---- dense per-point products (forward / input-gradient GEMMs with fused tails) ----------------
 * replaces the matmuls / Conv1d(k=1) of an HS layer and of the heads and the element-wise tail around them:
 *   feature_map @ self.weights + self.bias                                  gcn3d.py:171   ("nn" B, bias)
 *   STE_layer(x) + conv2(cat[feature, f_global]) + feature                  gcn3d.py:149,156,186 (two sources,
 *                                                  resid = feature, cloud_bias = f_global Wb^T per cloud)
 *   their input gradients (g Wa ; g Wste + gfm W^T)                          autograd of the above
 *   Conv1d(Cin, Cout, 1) of the heads                                       PoseR.py:16-39, PoseTs.py:18-45, FaceRecon.py:37-68
 * C (M,N) = alpha * (A1 (M,K1) op(B1) [+ A2 (M,K2) op(B2)]) [+ bias (N)] [+ resid (M,N)] [+ cloud_bias[row / rows_per_cloud] (N)]
 *           [+ xyz3[row] . w3[col]]     (xyz3 (M,3), w3 (N,3) fp32: the K = 3 STE of HSlayer_surface on raw coordinates,
 *                                        gcn3d.py:85 -- coordinates never pass through bf16, gcn3d.py:57,59)
 * b?_layout 0 = "nt": B is (N,K), k contiguous (a Linear / Conv1d weight);  1 = "nn": B is (K,N) (HS_layer.weights).
 * A2 == NULL: single source.  Leading dimensions in elements; any K, any alignment (aligned operands stage 16 bytes
 * per load).  fp32: v_mfma_f32_32x32x2_f32 (exact fp32).  bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate, "nt"
 * operands only, bias / cloud_bias fp32, resid bf16, C bf16 (round to nearest even) or fp32.
 */
int hsp_gemm_rows_f32(const float *A1, int lda1, const float *B1, int ldb1, int b1_layout, int K1,
                      const float *A2, int lda2, const float *B2, int ldb2, int b2_layout, int K2, int M, int N,
                      const float *bias, const float *resid, int ldr, const float *cloud_bias, int rows_per_cloud,
                      float alpha, const float *xyz3, const float *w3, float *C, int ldc, void *ws, size_t ws_bytes,
                      hspStream_t stream);
/* ws: hsp_gemm_rows_workspace_bytes(M,N,K1,K2, 4 | 2) -- 0 unless few output tiles meet a deep K (the input-gradient
 * products), which then split K over up to 16 workgroups per tile and fold the partial tiles in a second launch; with
 * ws == NULL (or too small) the product runs unsplit */
size_t hsp_gemm_rows_workspace_bytes(int M, int N, int K1, int K2, int elem_bytes);
int hsp_gemm_rows_bf16(const hsp_bf16_t *A1, int lda1, const hsp_bf16_t *B1, int ldb1, int K1,
                       const hsp_bf16_t *A2, int lda2, const hsp_bf16_t *B2, int ldb2, int K2, int M, int N,
                       const float *bias, const hsp_bf16_t *resid, int ldr, const float *cloud_bias,
                       int rows_per_cloud, float alpha, const float *xyz3, const float *w3, void *C, int ldc,
                       int c_is_f32 /* != 0: C is fp32 (an output that feeds BatchNorm keeps its mantissa) */,
                       void *ws, size_t ws_bytes, hspStream_t stream);

/* The same product (same arguments, same result contract; reference gcn3d.py:149,171,186 and their input gradients) by the
 * LDS-free wave-level kernel of csrc/gemm_wave.hip: every wave is an independent worker that streams both operands from
 * L2 / L1 straight into MFMA operand layout (16-byte buffer loads, a ring of 4 steps of 8 k in flight that does not drain
 * between the sources of a product nor between the tiles of a wave), no LDS, no barriers; the work is cut per WAVE -- a
 * contiguous run of (32|64) x (32|64|128) tiles each, the tiles that do not divide by the wave count dealt out as 32 x 32
 * blocks -- so there are no partial sums and the result does not depend on the cut.  Covers K1, K2 multiples of 32, N a
 * multiple of 32, 16-byte aligned rows (hsp_gemm_wave_supported; anything else: hsp_gemm_rows_f32); fp32 on
 * v_mfma_f32_32x32x2_f32.  cfg: 0 = automatic tile / cut; otherwise RB | NCB << 4 | waves_per_simd << 16 | order << 28
 * (tuning: rows / 32 and columns / 32 of the wave tile, occupancy, 1 = row panels fastest in the tile order); bit 29 of cfg:
 * relu on the result (the xyz3 + residual + per-cloud-bias form only: relu(conv_0(...)) of FaceRecon.py:88 in the epilogue). */
int hsp_gemm_wave_supported(int M, int N, int K1, int K2, int cfg);
/* host-only: out[10] = rows / 32 and columns / 32 of the wave tile, waves per SIMD, tiles along M and N, whole tiles per wave,
 * waves, first leftover tile, 32 x 32 blocks of the leftover tiles, 0; returns 0 when the shape is not covered */
int hsp_gemm_wave_plan_info(int M, int N, int K1, int K2, int cfg, int *out);
int hsp_gemm_wave_f32(const float *A1, int lda1, const float *B1, int ldb1, int b1_layout, int K1,
                      const float *A2, int lda2, const float *B2, int ldb2, int b2_layout, int K2, int M, int N,
                      const float *bias, const float *resid, int ldr, const float *cloud_bias, int rows_per_cloud,
                      float alpha, const float *xyz3, const float *w3, float *C, int ldc, int cfg, hspStream_t stream);

/* ---- fp32 dense products on the bf16 matrix cores, fp32-accurate (csrc/gemm_x3.hip) ---------------------------------------
 * replaces the same reference lines as hsp_gemm_rows_f32 (gcn3d.py:149,171,186 and their input gradients; the Conv1d(k=1)
 * layers of PoseR.py:16-39, PoseTs.py:18-45, FaceRecon.py:37-68):
 *   C = alpha * (A1 W1^T (+ A2 W2^T)) (+ bias) (+ resid) (+ cloud_bias[row / rows_per_cloud])        A*: fp32 rows (M, K*)
 * Every fp32 operand is split EXACTLY into three bf16 slices (x = hi + mid + lo, truncations) and six slice products per k
 * -- each exact in fp32 -- are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are below 2^-23 |a||b| per
 * product (the rounding an fp32 fma chain commits per step).  The weight-side operand comes ALREADY SPLIT in (N, K) form:
 * three bf16 planes (N, ldp), plane p at P + p * ps elements, columns k >= K zero, ldp >= K rounded up to 32 -- written once
 * per step for every weight by hsp_split_params_x3 (HspSplitDesc: src (rows, cols) fp32 with pitch ld; transpose = 0: src is
 * (N, K); 1: src is (K, N), i.e. the product wants src^T; kp = plane row pitch, ps = plane stride, both in elements and even; tile0 =
 * number of 32 (n) x 64 (k) output pieces, ceil(N / 32) * ceil(K / 64) each, of the entries before; table in DEVICE memory).  Covers N >= 64 and 16-byte aligned
 * activation rows (hsp_gemm_x3_supported; anything else: hsp_gemm_rows_f32); epilogues: none, bias, resid + cloud_bias.
 * Deep-K products with few tiles split K over workgroups (ws: hsp_gemm_x3_workspace_bytes) and fold in a fixed order. */
typedef struct HspSplitDesc { const float *src; void *dst; int rows, cols, ld, transpose, kp, tile0; long long ps; } HspSplitDesc;
int hsp_split_params_x3(const HspSplitDesc *table_dev, int n, int total_tiles, hspStream_t stream);
int hsp_gemm_x3_supported(int M, int N, int K1, int K2);
size_t hsp_gemm_x3_workspace_bytes(int M, int N, int K1, int K2);
int hsp_gemm_x3_f32(const float *A1, int lda1, const hsp_bf16_t *P1, int ldp1, long long ps1, int K1,
                    const float *A2, int lda2, const hsp_bf16_t *P2, int ldp2, long long ps2, int K2, int M, int N,
                    const float *bias, const float *resid, int ldr, const float *cloud_bias, int rows_per_cloud,
                    float alpha, float *C, int ldc, void *ws, size_t ws_bytes, hspStream_t stream);
/* the layer's out product (residual + per-cloud bias) that also leaves the FIRST PASS of the train-mode BatchNorm that follows it
 * (FaceRecon.py:90-95): bn_part[(M + 63) / 64][2][N] = per 64-row tile, sum (c - s[n]) and sum (c - s[n])^2 over the tile's rows
 * of the result, with the shift s[n] = resid[0][n] + cloud_bias[0][n] (this step's data only, so a replayed graph and an eager
 * step agree bit for bit) written to bn_shift (N floats); hsp_bn_relu_fwd_partials folds them (nblk = (M + 63) / 64 <= 512) */
int hsp_gemm_x3_bn_f32(const float *A1, int lda1, const hsp_bf16_t *P1, int ldp1, long long ps1, int K1,
                       const float *A2, int lda2, const hsp_bf16_t *P2, int ldp2, long long ps2, int K2, int M, int N,
                       const float *resid, int ldr, const float *cloud_bias, int rows_per_cloud, float *C, int ldc,
                       float *bn_shift, float *bn_part, hspStream_t stream);
/* the same for a Linear / Conv1d(k=1) WITH BIAS that a train-mode BatchNorm follows (the heads: PoseR.py:27-30, PoseTs.py:32-36,
 * FaceRecon.py:37-47): C = A W^T + bias plus bn_part[tiles][2][N], tiles = hsp_gemm_x3_bn_tiles(M, N) (64- or 128-row tiles by
 * shape, <= 512), shift = bias (written to bn_shift) */
int hsp_gemm_x3_bn_tiles(int M, int N);
int hsp_gemm_x3_bias_bn_f32(const float *A1, int lda1, const hsp_bf16_t *P1, int ldp1, long long ps1, int K1, int M, int N,
                            const float *bias, float *C, int ldc, float *bn_shift, float *bn_part, hspStream_t stream);

/* the per-CLOUD products of the ORL branch (gcn3d.py:186: the f_global half of conv2, one row per cloud of the batch), one
 * launch each, fp32 fma chains in a fixed order:
 *   hsp_small_rows_f32:  out (M, N) = alpha * A (M, K) op(W), M <= 64 (<= 16 unless K is a multiple of 128); w_layout 0: W is (N, K) (t = fg Wb^T), 1: W is (K, N)
 *                        (gfg = gt Wb / N); K <= 2048
 *   hsp_small_outer_f32: out (Ma, Nb) = a^T c over the B <= 64 rows of a (B, Ma), c (B, Nb) (gWb = gt^T fg) */
int hsp_small_rows_f32(const float *A, int lda, const float *W, int ldw, int w_layout, int M, int N, int K, float alpha,
                       float *out, int ldo, hspStream_t stream);
int hsp_small_outer_f32(const float *a, int lda, const float *c, int ldc, int B, int Ma, int Nb, float *out, int ldo,
                        const float *mom, int ldm, int Cm, float *gste, hspStream_t stream);
/* (mom != NULL: the same launch also writes gste (Cm, 3) = sum over the B rows of mom (B, >= 3 Cm) [j * Cm + c] -- the STE weight
 * gradient of HSlayer_surface, g^T xyz (gcn3d.py:85), from the per-cloud coordinate moments of hsp_colsum_rows_xyz:
 * out4 (B, 4, C), slot 0 = sum_i g[b][i][:], slots 1..3 = sum_i g[b][i][:] * xyz[b][i][0..2]; ws >= 4 * hsp_orl_workspace_bytes) */
int hsp_colsum_rows_xyz(const float *x, const float *xyz, int B, int N, int C, float *out4, void *ws, size_t ws_bytes,
                        hspStream_t stream);
int hsp_colsum_rows_xyz_bf16(const hsp_bf16_t *x, const float *xyz, int B, int N, int C, float *out4, void *ws, size_t ws_bytes,
                             hspStream_t stream);

/* fp32 master parameters -> bf16 working copies for the *_bf16 entry points, every tensor of a step in one launch:
 * entry e copies src (rows, cols; row pitch ld) to dst (rows, cols) and / or dstT (cols, rows) -- either may be NULL --
 * rounding to nearest even.  tile0 = number of 32 x 32 tiles of the entries before e; table_dev lives in DEVICE memory. */
typedef struct HspCastDesc { const float *src; void *dst; void *dstT; int rows, cols, ld, tile0; } HspCastDesc;
int hsp_cast_params_bf16(const HspCastDesc *table_dev, int n, int total_tiles, hspStream_t stream);

/* ---- weight-gradient GEMM ---------------------------------------------------------------------
 * replaces the parameter-gradient matmuls autograd runs for `feature_map @ self.weights + self.bias`
 * (gcn3d.py:171) and the 1x1 Conv1d layers (gcn3d.py:85,149,186):
 *   C[m][n] = sum_k A[k][m]*B[k][n],  A (K,M) row stride lda, B (K,N) row stride ldb, C row stride ldc;
 *   colsum_B (N) = sum_k B[k][n] when non-NULL (the bias gradient).  K is the point-row count (deep),
 * M and N multiples of 64; split-K over independent waves on v_mfma_f32_32x32x2_f32, partials folded in
 * a fixed order (deterministic).  ws: hsp_wgrad_workspace_bytes(M,N,K).
 */
size_t hsp_wgrad_workspace_bytes(int M, int N, int K);
/* split forms for a backward that computes several parameter gradients (an HS layer has three): hsp_wgrad_partial_* runs the
 * split-K launch only and describes the pending fold in *pending (a HOST struct; the workspace must stay alive and untouched
 * until the fold); hsp_wgrad_fold folds up to HSP_FOLD_MAX_WGRAD pending problems in ONE launch (same fixed order, same
 * results). */
#define HSP_FOLD_MAX_WGRAD 24
#define HSP_FOLD_MAX_DIRS 8
typedef struct HspWgradPending {
    const void *part, *cs_part;   /* split-K partials in the problem's workspace */
    void *C, *colsum;             /* outputs (colsum may be NULL) */
    int nparts, M, N, ldc;
} HspWgradPending;
int hsp_wgrad_fold(const HspWgradPending *pending, int n, hspStream_t stream);
/* the pending fold of a receptive-field layer's support-direction gradient (hsp_rf_*_bwd*_partial below): nparts per-cloud
 * partials (3, SC) in the call's workspace -> grad_dirs (3, SC) through the Jacobian of F.normalize(directions, dim=0)
 * (gcn3d.py:100,166).  A HOST struct; the workspace must stay alive and untouched until the fold. */
typedef struct HspDirsPending {
    const void *part, *dirs;      /* partials; the layer's RAW directions parameter (3, SC) */
    void *grad_dirs;              /* output (3, SC) */
    int nparts, SC;
} HspDirsPending;
/* every fold a backward pass left pending in ONE launch: nw <= HSP_FOLD_MAX_WGRAD parameter-gradient folds and
 * nd <= HSP_FOLD_MAX_DIRS direction-gradient folds (autograd of gcn3d.py:149,171,186 and :166 produce them; nothing before the
 * optimizer reads them).  Same summation order as the stand-alone folds: same bits. */
int hsp_step_fold(const HspWgradPending *wgrads, int nw, const HspDirsPending *dirs, int nd, hspStream_t stream);
/* hsp_rf_surface_bwd / hsp_rf_conv_bwd_scatter (and their bf16-storage twins) WITHOUT the direction-gradient fold: the tile
 * launch only; *pending describes the fold that hsp_step_fold runs later (grad_dirs is not valid until then). */
int hsp_rf_surface_bwd_partial(const float *xyz, const float *dirs, const uint16_t *argrow, const float *grad_out, int B, int N,
                               int S, int K, float *grad_dirs, void *ws, size_t ws_bytes, HspDirsPending *pending,
                               hspStream_t stream);
int hsp_rf_conv_bwd_scatter_partial(const float *xyz, const float *dirs, const float *fm, const float *fwin,
                                    const uint16_t *argrow, const float *grad_out, int B, int N, int S, int C, float *grad_fm,
                                    float *grad_dirs, void *ws, size_t ws_bytes, HspDirsPending *pending, hspStream_t stream);
int hsp_rf_surface_bwd_partial_bf16(const float *xyz, const float *dirs, const uint16_t *argrow, const hsp_bf16_t *grad_out,
                                    int B, int N, int S, int K, float *grad_dirs, void *ws, size_t ws_bytes,
                                    HspDirsPending *pending, hspStream_t stream);
int hsp_rf_conv_bwd_scatter_partial_bf16(const float *xyz, const float *dirs, const hsp_bf16_t *fm, const hsp_bf16_t *fwin,
                                         const uint16_t *argrow, const hsp_bf16_t *grad_out, int B, int N, int S, int C,
                                         hsp_bf16_t *grad_fm, float *grad_dirs, void *ws, size_t ws_bytes,
                                         HspDirsPending *pending, hspStream_t stream);
/* two pending weight gradients from ONE split-K launch (an HS layer's backward has two that depend only on the incoming
 * gradient and are each too small to fill the chip: g^T F and g^T X -- autograd of gcn3d.py:186 and :149); pending[2] */
int hsp_wgrad_partial_pair_f32(const float *A0, int lda0, const float *B0, int ldb0, int M0, int N0, int K0, float *C0, int ldc0,
                               void *ws0, size_t ws_bytes0, const float *A1, int lda1, const float *B1, int ldb1, int M1, int N1,
                               int K1, float *C1, int ldc1, void *ws1, size_t ws_bytes1, HspWgradPending *pending,
                               hspStream_t stream);
int hsp_wgrad_f32(const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C, int ldc,
                  float *colsum_B, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_wgrad_partial_f32(const float *A, int lda, const float *B, int ldb, int M, int N, int K, float *C, int ldc,
                          float *colsum_B, void *ws, size_t ws_bytes, HspWgradPending *pending, hspStream_t stream);

/* ---- BatchNorm1d (train mode) + ReLU over point rows -------------------------------------------
 * replaces F.relu(bn(x.transpose(1,2)).transpose(1,2))          FaceRecon.py:27-29, :90-95
 * x (R,C) point rows (R = B*N), batch statistics over R (biased variance for the normalisation, eps),
 * running_mean/var updated in place with `momentum` (unbiased variance) and *num_batches_tracked += 1
 * when the pointers are non-NULL -- nn.BatchNorm1d semantics.  y = relu ? max(0, bn(x)) : bn(x).
 * save_mean / save_invstd (C) feed the backward.  ws: hsp_bn_workspace_bytes(R, C).
 */
size_t hsp_bn_workspace_bytes(int R, int C);
int hsp_bn_relu_fwd(const float *x, int R, int C, const float *gamma, const float *beta, float eps,
                    float momentum, int relu, float *y, float *save_mean, float *save_invstd,
                    float *running_mean, float *running_var, long long *num_batches_tracked, void *ws,
                    size_t ws_bytes, hspStream_t stream);
/* the affine + ReLU part alone with given statistics (eval mode: mean = running_mean,
 * invstd = 1/sqrt(running_var + eps)) */
int hsp_bn_relu_apply(const float *x, int R, int C, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, int relu, float *y, hspStream_t stream);
/* dx (R,C), dgamma (C), dbeta (C) OVERWRITTEN; x is the forward INPUT (the ReLU mask is recomputed). */
int hsp_bn_relu_bwd(const float *x, const float *dy, int R, int C, const float *gamma, const float *beta,
                    const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                    float *dbeta, void *ws, size_t ws_bytes, hspStream_t stream);
/* the same for a tensor with TWO consumers: dy rows of pitch ldy (>= C, even -- 8-byte aligned rows: a column block of a wider gradient
 * tensor, e.g. of a dense (B, N, 1286) one, is consumed in place) plus an optional second incoming gradient dy2 (pitch ldy2, NULL: none), added as they are read */
int hsp_bn_relu_fwd_partials(const float *x, int R, int C, const float *gamma, const float *beta, float eps, float momentum,
                             int relu, float *y, float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                             long long *num_batches_tracked, const float *partial, int nblk, const float *shift,
                             hspStream_t stream);
int hsp_bn_relu_bwd2(const float *x, const float *dy, int ldy, const float *dy2, int ldy2, int R, int C, const float *gamma,
                     const float *beta, const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                     float *dbeta, void *ws, size_t ws_bytes, hspStream_t stream);

/* ---- depth -> point cloud front end -------------------------------------------------------------
 * replaces the device work of PC_sample(obj_mask, Depth, camK, coor2d)    network/point_sample/pc_sample.py:8-77
 * mask (B,HW) fp32 (object mask, already arg-maxed if it was a 2-channel prediction), depth (B,HW).
 * hsp_pc_compact: pix (B,HW) int32 = ids of the pixels with mask*(depth>0) > 0 in row-major order (the order
 * of torch boolean indexing), count (B) int32.  The HOST then draws np.random.choice(count[b], S, replace =
 * count[b] < S) per image exactly like the reference (pc_sample.py:57-66) and
 * hsp_pc_gather back-projects the chosen pixels: pc (B,S,3) = ((u-cx)*d/fx, (v-cy)*d/fy, d) / 1000,
 * coor2d (B,2,HW), camK (B,3,3), choose (B,S) int32.
 */
size_t hsp_pc_compact_workspace_bytes(int B, int HW);   /* per-chunk counts of the two-launch compaction */
int hsp_pc_compact(const float *mask, const float *depth, int B, int HW, int32_t *pix, int32_t *count,
                   void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_pc_gather(const float *depth, const float *coor2d, const float *camK, const int32_t *pix,
                  const int32_t *choose, int B, int HW, int S, float *pc, hspStream_t stream);

/* dataset-side variant: replaces PoseDataset._depth_to_pcl(depth, K, xymap, mask) / 1000.0
 * datasets/load_data.py:322-333, :275 (numpy float64 arithmetic, fp32 result).  camK (B,9) DOUBLE, as the
 * loader holds it; mask/compaction via hsp_pc_compact; choose = the loader's _sample_points ids (:308-320). */
int hsp_depth_to_pcl(const float *depth, const float *xymap, const double *camK, const int32_t *pix,
                     const int32_t *choose, int B, int HW, int S, float *pc, hspStream_t stream);

/* ---- pose matrix assembly -----------------------------------------------------------------------
 * replaces generate_RT([p_green,p_red],[f_green,f_red], T, 'vec', sym)     tools/geom_utils.py:232-244
 * (with to_R_matrices / get_vertical_rot_vec_in_batch / get_rot_mat_y_first, tools/rot_utils.py:39-100)
 * p_green, p_red (B,3) unit axes, f_green, f_red (B) confidences, T (B,3), sym (B, sym_stride) (column 0 == 1
 * zeroes the red confidence) -> out (B,4,4).
 */
int hsp_generate_rt(const float *p_green, const float *p_red, const float *f_green, const float *f_red,
                    const float *T, const float *sym, int sym_stride, int B, float *out, hspStream_t stream);

/* ---- optimizer step (training driver) ---------------------------------------------------------------
 * All parameters / gradients / optimizer state of a parameter group live in flat fp32 buffers; `rows` (device)
 * lists the dim-0 slices of every >= 2-D tensor (gc = 1: gradient centralisation applies) and chunks of the
 * 1-D tensors (gc = 0).
 * hsp_sumsq_f32: out[0] = sum x^2 (the squared total norm of torch.nn.utils.clip_grad_norm_, engine/train.py:99).
 * hsp_ranger_step replaces Ranger.step() tools/torch_utils/solver/ranger2020.py:135-246 for the whole group:
 *   clip (coef = min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6)); gnorm_sq NULL = no clipping), gradient
 *   centralisation (before the moments, or with gc_after on the generalised gradient), RAdam moments,
 *   adaptive (N_sma > threshold, decided by the host from the step count) or plain momentum step with
 *   step_lr = step_size * lr, weight decay, and on lookahead steps slow += la_alpha * (p - slow); p = slow.
 */
typedef struct HspRowDesc { long long offset; int len; int gc; } HspRowDesc;
size_t hsp_sumsq_workspace_bytes(long long n);
int hsp_sumsq_f32(const float *x, long long n, float *out, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_ranger_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, float *slow,
                    const HspRowDesc *rows, int nrows, float beta1, float beta2, float eps, float weight_decay,
                    float step_lr, int adaptive, int lookahead, float la_alpha, int gc_after,
                    const float *gnorm_sq, float max_norm, hspStream_t stream);

/* ---- Chamfer distance -------------------------------------------------------------------------
 * replaces cd.forward_cuda / cd.backward_cuda    tools/pyTorchChamferDistance/chamfer_distance.cpp:27-56
 * xyz1 (B,n,3), xyz2 (B,m,3) -> dist1 (B,n), dist2 (B,m) squared NN distances, idx1/idx2 int32 arg-min
 * (first minimum wins, chamfer_distance.cpp:78).  bwd: gx1/gx2 OVERWRITTEN with +-2g(p-q) sums
 * (chamfer_distance.cpp:141-175).
 */
int hsp_chamfer_fwd(const float *xyz1, const float *xyz2, int B, int n, int m, float *dist1, float *dist2,
                    int32_t *idx1, int32_t *idx2, hspStream_t stream);
int hsp_chamfer_bwd(const float *xyz1, const float *xyz2, const int32_t *idx1, const int32_t *idx2,
                    const float *gd1, const float *gd2, int B, int n, int m, float *gx1, float *gx2,
                    hspStream_t stream);

/* ---- farthest point sampling -----------------------------------------------------------------
 * replaces farthest_point_sampling(points, n)         tools/eval_utils.py:107-119 (per cloud)
 * xyz (B,N,3) -> sel (B,n_samples): start at 0, running min of the Euclidean distances, first maximum wins.
 * The arithmetic follows the dtype the numpy helper is called with (eval_utils.py:73-84):
 *   hsp_fps_f32 -- a float32 cloud: (x*x + y*y) + z*z and a correctly rounded sqrt, all in fp32;
 *   hsp_fps_f64 -- a float64 cloud (what tools/eval_utils.py:122-140 passes): the same in fp64.
 * The sqrt is part of the contract: it creates exact ties that the first-maximum rule resolves by index.
 * ws: hsp_fps_workspace_bytes(B,N) for f32, twice that for f64.
 */
size_t hsp_fps_workspace_bytes(int B, int N);
int hsp_fps_f32(const float *xyz, int B, int N, int n_samples, int32_t *sel, void *ws, size_t ws_bytes,
                hspStream_t stream);
int hsp_fps_f64(const double *xyz, int B, int N, int n_samples, int32_t *sel, void *ws, size_t ws_bytes,
                hspStream_t stream);

/* ---- bf16 feature storage (BASELINE configs[3]: dense clouds, bf16 features / weights / fm / gradients) -----------
 * Twins of the entry points above for feature tensors stored as bfloat16 (hsp_bf16_t = the upper 16 bits of an fp32).
 * Same argument roles and semantics; every kernel still computes in fp32 (loads widen, stores round to nearest even).
 * What stays fp32: xyz, support directions and theta (the reference hard-casts them with .float(), gcn3d.py:57,59),
 * arg-max / index tensors, BatchNorm statistics and affine parameters, every parameter gradient, per-cloud (B,C) rows
 * (ORL global feature and its gradient).  The fp32 master parameters are rounded once per step by hsp_cast_params_bf16.
 * hsp_knn_bf16: inner products on v_mfma_f32_32x32x16_bf16 (exact products, fp32 accumulation), |x|^2 in fp32, the fp32
 * path's distance expression and selection rule; C % 32 == 0.  ws: B*N*4 bytes.
 */
int hsp_knn_bf16(const hsp_bf16_t *x, int B, int N, int C, int k, int drop_first, int32_t *idx, void *ws,
                 size_t ws_bytes, hspStream_t stream);
int hsp_rf_surface_fwd_bf16(const float *xyz, const int32_t *idx, const float *dirs, int B, int N, int k, int S, int K,
                            hsp_bf16_t *out, uint16_t *argrow, hspStream_t stream);
int hsp_rf_surface_bwd_bf16(const float *xyz, const float *dirs, const uint16_t *argrow, const hsp_bf16_t *grad_out, int B,
                            int N, int S, int K, float *grad_dirs, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_rf_conv_wants_fwin_bf16(int N, int S, int C);
int hsp_rf_conv_fwd_bf16(const float *xyz, const int32_t *idx, const float *dirs, const hsp_bf16_t *fm, int B, int N, int k,
                         int S, int C, hsp_bf16_t *out, uint16_t *argrow, hsp_bf16_t *fwin, hspStream_t stream);
int hsp_rf_conv_bwd_scatter_bf16(const float *xyz, const float *dirs, const hsp_bf16_t *fm, const hsp_bf16_t *fwin,
                                 const uint16_t *argrow, const hsp_bf16_t *grad_out, int B, int N, int S, int C,
                                 hsp_bf16_t *grad_fm, float *grad_dirs, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_gather_max_fwd_bf16(const hsp_bf16_t *feat, const int32_t *idx, const int32_t *qsel, int B, int Nsrc, int Nidx,
                            int Nq, int k, int kstride, int C, hsp_bf16_t *out, uint8_t *argmax, hspStream_t stream);
/* grad_out: (B,Nq,C) bf16, or with grad_bcast != 0 the fp32 (B,C) per-cloud row; LDS tile form only */
int hsp_gather_max_bwd_bf16(const void *grad_out, int grad_bcast, const int32_t *idx, const int32_t *qsel,
                            const uint8_t *argmax, int B, int Nsrc, int Nidx, int Nq, int kstride, int C,
                            hsp_bf16_t *grad_feat, int accumulate, const hsp_bf16_t *extra, hspStream_t stream);
int hsp_orl_global_fwd_bf16(const hsp_bf16_t *feat, const int32_t *idx, int B, int N, int k, int kstride, int C, float *fg,
                            uint8_t *argmax, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_colsum_rows_bf16(const hsp_bf16_t *x, int B, int N, int C, float *out, void *ws, size_t ws_bytes,
                         hspStream_t stream);
/* out_pitch: row pitch of out in elements (>= sum of widths; padding columns are zeroed when every segment is 16-byte
 * aligned -- the feat assembly -- and left untouched otherwise).  bf16 form: kind 0 / 1
 * sources are bf16, kind 2 (per-cloud rows) and kind 3 (the (B) float
 * category ids, expanded to one-hot columns in place) fp32 */
int hsp_concat_rows_pitched(int nseg, const float *const *src, const int32_t *const *idx, const int *width, const int *kind,
                            const int *nsrc, int B, int N, float *out, int out_pitch, hspStream_t stream);
int hsp_concat_rows_bf16(int nseg, const void *const *src, const int32_t *const *idx, const int *width, const int *kind,
                         const int *nsrc, int B, int N, hsp_bf16_t *out, int out_pitch, hspStream_t stream);
int hsp_gather_rows_bwd_csr_bf16(const hsp_bf16_t *grad_out, int grad_stride, const int32_t *rev_off,
                                 const int32_t *rev_edge, int B, int Nsrc, int Nq, int C, hsp_bf16_t *grad_feat,
                                 hspStream_t stream);
int hsp_wgrad_bf16(const hsp_bf16_t *A, int lda, const hsp_bf16_t *B, int ldb, int M, int N, int K, float *C, int ldc,
                   float *colsum_B, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_wgrad_partial_bf16(const hsp_bf16_t *A, int lda, const hsp_bf16_t *B, int ldb, int M, int N, int K, float *C, int ldc,
                           float *colsum_B, void *ws, size_t ws_bytes, HspWgradPending *pending, hspStream_t stream);
/* "mixed": x fp32 (the pre-BatchNorm layer output is kept in fp32: with |mean| >> std per channel a bf16 x would leave
 * the normalised value only a few significant bits), y / dy / dx bf16 */
int hsp_bn_relu_fwd_mixed(const float *x, int R, int C, const float *gamma, const float *beta, float eps, float momentum,
                          int relu, hsp_bf16_t *y, float *save_mean, float *save_invstd, float *running_mean,
                          float *running_var, long long *num_batches_tracked, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_bn_relu_apply_mixed(const float *x, int R, int C, const float *mean, const float *invstd, const float *gamma,
                            const float *beta, int relu, hsp_bf16_t *y, hspStream_t stream);
int hsp_bn_relu_bwd_mixed(const float *x, const hsp_bf16_t *dy, int R, int C, const float *gamma, const float *beta,
                          const float *save_mean, const float *save_invstd, int relu, hsp_bf16_t *dx, float *dgamma,
                          float *dbeta, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_bn_relu_fwd_bf16(const hsp_bf16_t *x, int R, int C, const float *gamma, const float *beta, float eps, float momentum,
                         int relu, hsp_bf16_t *y, float *save_mean, float *save_invstd, float *running_mean,
                         float *running_var, long long *num_batches_tracked, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_bn_relu_apply_bf16(const hsp_bf16_t *x, int R, int C, const float *mean, const float *invstd, const float *gamma,
                           const float *beta, int relu, hsp_bf16_t *y, hspStream_t stream);
int hsp_bn_relu_bwd_bf16(const hsp_bf16_t *x, const hsp_bf16_t *dy, int R, int C, const float *gamma, const float *beta,
                         const float *save_mean, const float *save_invstd, int relu, hsp_bf16_t *dx, float *dgamma,
                         float *dbeta, void *ws, size_t ws_bytes, hspStream_t stream);

/* ---- training losses of the PoseNet_only stage -----------------------------------------------------------
 * replaces, for one batch, the four loss modules as network/HSPose.py:84-160 wires them:
 *   fs_net_loss          losses/fs_net_loss.py:95-235      Rot1 Rot1_cos Rot2 Rot2_cos Rot_r_a Tran Size R_con
 *   recon_6face_loss     losses/recon_loss.py:464-649      recon_per_p recon_p_f recon_point_{vote,r,t,s,self}
 *   geo_transform_loss   losses/geometry_loss.py:123-150   geo_point
 *   prop_rot_loss        losses/prop_loss.py:156-276       Prop_pm Prop_sym_recon Prop_sym_rt
 * terms (19) in that order, each already multiplied by its FLAGS weight.  PC (B,N,3), gt_R (B,3,3), gt_t / gt_s /
 * mean_shape (B,3), sym (B,4), obj_id (B) fp32; network outputs recon (B,N,3), face_normal (B,N,6,3), face_dis / face_f
 * (B,N,6) in the network's face order (y+ x+ z+ x- z- y-), p_green / p_red / pred_T / pred_s (B,3), f_green / f_red (B).
 * The axis confidences are variables only in R_con and the face confidences only in recon_p_f (HSPose.py detaches them
 * elsewhere).  `cfg` is a HOST struct.  The workspace written by _fwd is read by _bwd (keep it until then).
 * _bwd: grad_terms (19) = d(objective)/d(term); every d_* buffer is OVERWRITTEN with the gradient w.r.t. that network output;
 * d_mom_scratch: B*54 floats.  Five launches in all; sums in a fixed order (bit-reproducible). */
#define HSP_LOSS_TERMS 19
typedef struct HspLossCfg {       /* config/config.py:64-93 */
    float rot_1_w, rot_2_w, rot_regular, tran_w, size_w, r_con_w;
    float recon_n_w, recon_d_w, recon_f_w, recon_v_w, recon_bb_r_w, recon_bb_t_w, recon_bb_s_w, recon_bb_self_w;
    float geo_p_w, prop_pm_w, prop_sym_w;
    int smooth_l1;                /* fsnet_loss_type: 0 = 'l1', 1 = 'smoothl1' (beta 0.5) */
} HspLossCfg;
size_t hsp_pose_losses_workspace_bytes(int B);
int hsp_pose_losses_fwd(const float *PC, const float *gt_R, const float *gt_t, const float *gt_s, const float *mean_shape,
                        const float *sym, const float *obj_id, const float *recon, const float *face_normal,
                        const float *face_dis, const float *face_f, const float *p_green, const float *p_red,
                        const float *f_green, const float *f_red, const float *pred_T, const float *pred_s, int B, int N,
                        const HspLossCfg *cfg, float *terms, void *ws, size_t ws_bytes, hspStream_t stream);
int hsp_pose_losses_bwd(const float *PC, const float *gt_R, const float *gt_t, const float *gt_s, const float *mean_shape,
                        const float *sym, const float *obj_id, const float *recon, const float *face_normal,
                        const float *face_dis, const float *face_f, const float *p_green, const float *p_red,
                        const float *f_green, const float *f_red, const float *pred_T, const float *pred_s, int B, int N,
                        const HspLossCfg *cfg, const float *grad_terms, const void *ws, size_t ws_bytes,
                        float *d_mom_scratch, float *d_recon, float *d_face_normal, float *d_face_dis, float *d_face_f,
                        float *d_green, float *d_red, float *d_f_green, float *d_f_red, float *d_T, float *d_s,
                        hspStream_t stream);

/* replaces HSPose.data_augment                          network/HSPose.py:185-256 over
 *          defor_3D_bb_in_batch / _rt_in_batch / _bc_in_batch / defor_3D_pc   datasets/data_augmentation.py:70-190
 * one launch for a training batch: box scaling in the object frame (aug_bb; x and z share the mean factor under rotational
 * symmetry), rigid perturbation (aug_rt_t, aug_rt_r), box-cage taper for bowls (1) / mugs (5) with the size taken from the
 * tapered model's extent * nocs_scale, per-point radial jitter noise * (p - t).  Each applies to the clouds whose draw is
 * below its probability.  draws (6,B): u_bb, u_rt, u_bc, ey_up, ey_down (uniforms in [0,1), mapped to [0.8,1.2)), u_pc -- the
 * caller draws them in the reference's order; noise (B,N,3) = rand * FLAGS.aug_pc_r (the reference's CPU draw, uploaded).
 * PC (B,N,3), model_point (B,M,3); gt_s / s_out are size residuals to mean_shape.  Outputs may not alias the inputs. */
int hsp_pose_augment(const float *PC, const float *gt_R, const float *gt_t, const float *gt_s, const float *mean_shape,
                     const float *sym, const float *aug_bb, const float *aug_rt_t, const float *aug_rt_r,
                     const float *model_point, const float *nocs_scale, const float *obj_id, const float *draws,
                     const float *noise, int B, int N, int M, float p_bb, float p_rt, float p_bc, float p_pc, float *PC_out,
                     float *R_out, float *t_out, float *s_out, hspStream_t stream);

/* the face head's output split (PoseNet9D.py:31-35) in one launch each way: face (R, 30) -> unit normals (R, 6, 3) =
 * face[:, :18] / its per-face 3-norm (no epsilon, as the reference), distances (R, 6) = face[:, 18:24], confidences (R, 6) =
 * sigmoid(face[:, 24:]); backward: g_face (R, 30) from the three incoming gradients (each may be NULL = zero). */
int hsp_face_split_fwd(const float *face, long long R, float *normals, float *dis, float *conf, hspStream_t stream);
int hsp_face_split_bwd(const float *face, const float *g_normals, const float *g_dis, const float *g_conf, long long R,
                       float *g_face, hspStream_t stream);

/* a rotation head's output split (PoseNet9D.py:40-46): h (B, 4) -> axis (B, 3) = h[:, 1:] / (||h[:, 1:]|| + 1e-6), confidence (B) =
 * sigmoid(h[:, 0]); backward: g_h (B, 4) from g_axis / g_conf (each may be NULL = zero). */
int hsp_axis_conf_fwd(const float *h, int B, float *axis, float *conf, hspStream_t stream);
int hsp_axis_conf_bwd(const float *h, const float *g_axis, const float *g_conf, int B, float *g_h, hspStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HSP_H_ */
