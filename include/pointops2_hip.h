/*
 * pointops2_hip.h — C ABI of libpointops2_hip.so, the MI355X (gfx950) implementation of the
 * Stratified Transformer's pointops2 hot path.
 *
 * PART 1 are the reference's own raw-pointer launchers (the "true C ABI" of lib/pointops2,
 * SURVEY.md §8b seam B2): same names, same argument order and meaning.  Every pointer is a
 * DEVICE pointer to a contiguous fp32 / int32 array.  Ownership follows the reference: the caller
 * allocates every output (and zero-fills it where the reference does, see each entry); the library
 * borrows the pointers for the duration of the launch, allocates nothing and returns void.
 *
 * Differences from the reference, all "stricter is compatible":
 *   - launches go to the stream set with pointops2_set_stream() (thread-local; default: the NULL
 *     stream, which is what the reference's <<<grid, block, 0>>> launches use);
 *   - instead of `throw "d != 16 and d != 32"` (attention_cuda_kernel_v2.cu:116) an unsupported
 *     argument records an error readable with pointops2_last_error(); the call is then a no-op;
 *   - outputs are fully written by the kernels (the caller's zero-fill is harmless, not required),
 *     except where noted "accumulates".
 *
 * PART 2 are the additional native entry points this build adds on the same path (segment softmax
 * = the torch_scatter.scatter_softmax call of the model, CSC transposition used by the backward
 * kernels, scratch memory for the bucketed exact FPS).
 */
#ifndef POINTOPS2_HIP_H
#define POINTOPS2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* runtime plumbing                                                                           */
/* ------------------------------------------------------------------------------------------ */
/* hipStream_t as void*.  Thread-local.  NULL = legacy default stream. */
void pointops2_set_stream(void *hip_stream);
void *pointops2_get_stream(void);
/* NULL when the last call on this thread succeeded; otherwise a static message.  Reading clears. */
const char *pointops2_last_error(void);
/* library/ABI version, bumped when a signature changes */
int pointops2_abi_version(void);
/* Diagnostic: how long (ticks of the 100 MHz clock, default 2 s) a workgroup of the round sampler waits at its grid barrier before
 * the sampler gives up and pointops2_last_error() reports the call's indices invalid (tests force the path with a tiny value). */
void pointops2_diag_set_fps_patience(unsigned long long ticks_100mhz);

/* ------------------------------------------------------------------------------------------ */
/* PART 1 — the reference's launcher set                                                      */
/* ------------------------------------------------------------------------------------------ */

/* sampling/sampling_cuda_kernel.h:14  — furthest point sampling per batch element.
 * n = largest batch element (selects the reference's block size, which fixes its tie rule);
 * xyz [N,3]; offset/new_offset [b] cumulative ends; tmp [N] scratch PRE-FILLED with 1e10;
 * idx [new_offset[b-1]] out. */
void furthestsampling_cuda_launcher(int b, int n, const float *xyz, const int *offset,
                                    const int *new_offset, float *tmp, int *idx);

/* knnquery/knnquery_cuda_kernel.h:14 — exact kNN, ascending; dist2 = SQUARED distances. nsample <= 100. */
void knnquery_cuda_launcher(int m, int nsample, const float *xyz, const float *new_xyz,
                            const int *offset, const int *new_offset, int *idx, float *dist2);

/* grouping/grouping_cuda_kernel.h:14-15 — gather rows / scatter-add (grad_input accumulates). */
void grouping_forward_cuda_launcher(int m, int nsample, int c, const float *input, const int *idx, float *output);
void grouping_backward_cuda_launcher(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input);

/* interpolation/interpolation_cuda_kernel.h:34-35 — k-NN weighted sum (output / grad_input accumulate). */
void interpolation_forward_cuda_launcher(int n, int c, int k, const float *input, const int *idx, const float *weight, float *output);
void interpolation_backward_cuda_launcher(int n, int c, int k, const float *grad_output, const int *idx, const float *weight, float *grad_input);

/* attention/attention_cuda_kernel.h:17-21 — pair-indexed (v1) forms; outputs ACCUMULATE (atomics in
 * the reference), so the caller's zero-fill is required. */
void attention_step1_forward_cuda_launcher(int N, int M, int h, int C, const float *q, const float *k,
                                           const int *index0, const int *index1, float *attn);
void attention_step1_backward_cuda_launcher(int N, int M, int h, int C, const float *grad_out,
                                            const int *index0, const int *index1, const float *q,
                                            const float *k, float *grad_q, float *grad_k);
void attention_step2_forward_cuda_launcher(int N, int M, int h, int C, const float *attn, const float *v,
                                           const int *index0, const int *index1, float *output);
void attention_step2_backward_cuda_launcher(int N, int M, int h, int C, const float *grad_out,
                                            const int *index0, const int *index1, const float *attn,
                                            const float *v, float *grad_attn, float *grad_v);

/* attention_v2/attention_cuda_kernel_v2.h:19-23 — CSR forms.  index0_offsets [N+1]; index1 [M];
 * n_max = longest segment (<= 1024, pointops.py:150).  C/h must be 16 or 32.
 * backward: grad_q fully written; grad_k ACCUMULATES (pre-zeroed by the caller). */
void attention_step1_forward_cuda_launcher_v2(int N, int M, int h, int C, const unsigned int n_max,
                                              const float *q, const float *k, const int *index0_offsets,
                                              const int *index1, float *attn);
void attention_step1_backward_cuda_launcher_v2(int N, int M, int h, int C, const unsigned int n_max,
                                               const float *grad_out, const int *index0_offsets,
                                               const int *index1, const float *q, const float *k,
                                               float *grad_q, float *grad_k);
void attention_step2_forward_cuda_launcher_v2(int N, int M, int h, int C, const float *attn, const float *v,
                                              const int *index0, const int *index1, float *output);
void attention_step2_backward_cuda_launcher_v2(int N, int M, int h, int C, const float *grad_out,
                                               const int *index0, const int *index1, const float *attn,
                                               const float *v, float *grad_attn, float *grad_v);

/* rpe/relative_pos_encoding_cuda_kernel.h:17-21 — v1 single-table forms (outputs accumulate). */
void dot_prod_with_idx_forward_cuda_launcher(int N, int M, int h, int hdim, const float *q, const int *index,
                                             const float *table, const int *rel_idx, float *output);
void dot_prod_with_idx_backward_cuda_launcher(int N, int M, int h, int hdim, const float *grad_out,
                                              const float *q, const int *index, const float *table,
                                              const int *rel_idx, float *grad_q, float *grad_table);
void attention_step2_with_rel_pos_value_forward_cuda_launcher(int N, int M, int h, int hdim, const float *attn,
                                                              const float *v, const int *index0, const int *index1,
                                                              const float *table, const int *rel_idx, float *output);
void attention_step2_with_rel_pos_value_backward_cuda_launcher(int N, int M, int h, int hdim, const float *grad_out,
                                                               const int *index0, const int *index1, const float *attn,
                                                               const float *v, const float *table, const int *rel_idx,
                                                               float *grad_attn, float *grad_v, float *grad_table);

/* rpe_v2/relative_pos_encoding_cuda_kernel_v2.h:22-29 — CSR forms.  table [L,h,hdim,3]; rel_idx [M,3].
 * hdim must be 16 or 32.  backward: grad_q / grad_attn fully written; grad_k / grad_v / table grads
 * ACCUMULATE (pre-zeroed by the caller).  The table length L is not part of the reference signature
 * but the fast kernels stage the tables in LDS: announce it with pointops2_set_table_rows(L) before any
 * *_v3 bias or *_v2 rel-pos-value launcher (forward and backward).  Without it the launchers are still
 * callable with the reference's arguments alone: generic kernels then read the tables from global
 * memory and accumulate with atomics as the reference does (same results, several times slower). */
void dot_prod_with_idx_forward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max, int T, const float *q,
                                                const int *index_q, const float *k, const int *index_k,
                                                const float *table_q, const float *table_k, const int *rel_idx,
                                                const int *rel_idx_offsets, const int *sort_indices, float *output);
void dot_prod_with_idx_backward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max, int T, const float *grad_out,
                                                 const float *q, const int *index_q, const float *k, const int *index_k,
                                                 const float *table_q, const float *table_k, const int *rel_idx,
                                                 const int *rel_idx_offsets, const int *sort_indices, float *grad_q,
                                                 float *grad_k, float *grad_table_q, float *grad_table_k);
void dot_prod_with_idx_forward_cuda_launcher_v3(int N, int M, int h, int hdim, int n_max, const float *q,
                                                const int *index_q_offsets, const float *k, const int *index_k,
                                                const float *table_q, const float *table_k, const int *rel_idx,
                                                float *output);
void dot_prod_with_idx_backward_cuda_launcher_v3(int N, int M, int h, int hdim, int n_max, const float *grad_out,
                                                 const float *q, const int *index_q_offsets, const float *k,
                                                 const int *index_k, const float *table_q, const float *table_k,
                                                 const int *rel_idx, float *grad_q, float *grad_k,
                                                 float *grad_table_q, float *grad_table_k);
void attention_step2_with_rel_pos_value_forward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max,
                                                                 const float *attn, const float *v,
                                                                 const int *index0_offsets, const int *index1,
                                                                 const float *table, const int *rel_idx, float *output);
void attention_step2_with_rel_pos_value_backward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max,
                                                                  const float *grad_out, const int *index0_offsets,
                                                                  const int *index1, const float *attn, const float *v,
                                                                  const float *table, const int *rel_idx,
                                                                  float *grad_attn, float *grad_v, float *grad_table);

/* subtraction/subtraction_cuda_kernel.h:14-15, aggregation/aggregation_cuda_kernel.h:14-15 — Point-Transformer
 * vector-attention ops bound by pointops_api.cpp:23-26 and called by no model of the reference (out of scope):
 * exported so that the reference's shim sources link; a call records an error and does nothing. */
void subtraction_forward_cuda_launcher(int n, int nsample, int c, const float *input1, const float *input2, const int *idx, float *output);
void subtraction_backward_cuda_launcher(int n, int nsample, int c, const int *idx, const float *grad_output, float *grad_input1,
                                        float *grad_input2);
void aggregation_forward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input, const float *position, const float *weight,
                                       const int *idx, float *output);
void aggregation_backward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input, const float *position, const float *weight,
                                        const int *idx, const float *grad_output, float *grad_input, float *grad_position,
                                        float *grad_weight);

/* ------------------------------------------------------------------------------------------ */
/* PART 2 — additional entry points of this build                                             */
/* ------------------------------------------------------------------------------------------ */

/* Rows L of the [L,h,hdim,3] tables for the next *_v2/_v3 rel-pos call on this thread. */
void pointops2_set_table_rows(int L);

/* Bucketed exact FPS (same index sequence as the reference, ~30x fewer bytes per step): taken by
 * furthestsampling_cuda_launcher when the caller has lent a scratch buffer of at least
 * pointops2_fps_workspace_bytes(b, N) bytes with pointops2_set_workspace() and announced the total
 * point count N = offset[b-1] of the next call with pointops2_set_point_count() (the reference
 * signature only carries the largest batch element).  Otherwise the single-workgroup scan runs. */
void pointops2_set_workspace(void *device_ptr, size_t bytes);
void pointops2_set_point_count(int N);
/* Grid-accelerated exact kNN (same idx/dist2 as the reference's full scan, ties replayed literally):
 * taken by knnquery_cuda_launcher when a workspace of pointops2_knn_workspace_bytes(n, m, b) bytes is
 * lent and the candidate count n (pointops2_set_point_count) and batch count b are announced. */
void pointops2_set_batch_count(int b);
size_t pointops2_knn_workspace_bytes(int n, int m, int b);
size_t pointops2_fps_workspace_bytes(int b, int N);
/* FPS is deterministic, so a request for fewer samples of the same cloud is a prefix of a longer one
 * (the model asks for n/8+1 and then n/4+1 samples of the same points, stratified_transformer.py:289,103).
 * If the workspace still holds the state of the previous call on the SAME xyz/offset, pass that call's
 * idx/new_offset here: the next furthestsampling_cuda_launcher copies those samples and continues
 * instead of starting over.  One-shot (cleared by the launch). */
void pointops2_set_fps_resume(const int *prev_idx, const int *prev_new_offset);
/* One-shot hint for the next furthestsampling_cuda_launcher: unordered != 0 = the caller knows that the cloud is NOT the output of an
 * earlier FPS in selection order (a raw scene), so the identity-prefix probe (six small launches, ~60 us in front of the sampler) is
 * skipped.  The result is the same with or without the hint, and with a wrong hint. */
void pointops2_set_fps_hint(int unordered);

/* Key-major ("CSC") transposition of a CSR pair list, used by the backward kernels instead of
 * global float atomics.  When set (thread-local, cleared with NULLs), the *_backward_* launchers
 * above gather by key; otherwise they build a temporary one themselves in `workspace`.
 *   csc_offsets [N+1], csc_pair [M] (pair id m, ascending per key), csc_query [M] (query of m). */
size_t pointops2_csc_workspace_bytes(int N, int M);
void pointops2_csc_build(int N, int M, const int *index0_offsets, const int *index1,
                         int *csc_offsets, int *csc_pair, int *csc_query,
                         void *workspace, size_t workspace_bytes);
void pointops2_set_csc(const int *csc_offsets, const int *csc_pair, const int *csc_query);
/* Rows of k / v when they outnumber the CSR's query rows N (a rank of a sharded scene holds all keys but only
 * its own queries); read by pointops2_csc_build (then csc_offsets has n+1 entries) and by the *_backward_*
 * launchers' key-side kernels.  0 (default) = N, the reference's implicit assumption. */
void pointops2_set_key_rows(int n);

/* torch_scatter.scatter_softmax(src [M,h], index_0, dim=0) over CSR segments
 * (model/stratified_transformer.py:205) and its backward. */
void segment_softmax_forward_launcher(int N, int M, int h, const float *src, const int *offsets, float *out);
void segment_softmax_backward_launcher(int N, int M, int h, const float *out, const float *grad_out,
                                       const int *offsets, float *grad_src);
/* ---- on-device index build of one stage (model/stratified_transformer.py:10-65, 186-190, 312-317) ----
 * All arrays are caller-allocated device memory; ws is scratch of pointops2_index_workspace_bytes(N) bytes.
 *   bbox:       out6 = {min x,y,z, max x,y,z} of xyz [N,3]
 *   partition:  one grid_sample(): cluster [N] dense window id (torch.unique rank of the voxel id), order [N] point
 *               ids sorted by (window, id), starts [N+2] bucket boundaries into order, n_windows [1].
 *               size = window edge, shift = value added to xyz before binning (0, or size/2 for the shifted
 *               partitions); key_bits = significant bits of the voxel ids (0 = all 64)
 *   window_coord: wc [N,3] = ((xyz [+ window/2]) - xyz_min) // window  (fp32 floor division), the mask operand of :28-34
 *   sampled_buckets: the FPS subset (sample_idx [m]) bucketed by a large-window partition: ls [m] point ids in
 *               (window, id) order, ls_starts [N+1]; `sampled` [N] must be zero-filled by the caller
 *   pairs_count: offsets [N+1] = exclusive scan of keys per query (offsets[N] = M)
 *   pairs_fill:  index_0 / index_1 [M], rel_idx [M,3] in the canonical order (dense keys ascending, then
 *               stratified keys ascending) */
void pointops2_bbox_launcher(int N, const float *xyz, float *out6);
size_t pointops2_index_workspace_bytes(int N);
void pointops2_window_partition_launcher(int N, int b, const float *xyz, const int *offset, const float *bbox6, float size,
                                         float shift, int key_bits, int *cluster, int *order, int *starts, int *n_windows,
                                         void *ws, size_t ws_bytes);
/* the four partitions of a stage in one sort (variant 0 small, 1 small shifted by window/2, 2 large = 2 window, 3 large shifted by
 * window): cluster / order [4][N], starts [4][N+2], n_windows [4], same contents as four partition calls.  The key is of fixed width
 * (ten bits per voxel coordinate: no bounding-box read-back); *overflow = 1 when a coordinate does not fit - the outputs are then
 * meaningless (but in range) and the caller builds the partitions one by one. */
/* Rows in window order: order [N] = the rows of a CSR pair list sorted by their first partner (rows of one window become neighbours).
 * pointops2_set_row_order(order, N) makes the operators' pair walkers (A1 / A2 / A4 forward and backward, by query and by key) take
 * their rows in that order, eight runs over the eight XCDs, whenever a launch walks exactly N rows: same results (the sums of a row
 * do not change), the gathers of neighbouring waves hit the second-level cache.  nullptr (the default): rows by index. */
size_t pointops2_row_order_workspace_bytes(int N);
void pointops2_row_order_launcher(int N, int M, const int *offsets, const int *index1, int *order, void *ws, size_t ws_bytes);
void pointops2_set_row_order(const int *order, int n_rows);
size_t pointops2_partitions4_workspace_bytes(int N);
void pointops2_window_partitions4_launcher(int N, int b, const float *xyz, const int *offset, const float *bbox6, float window,
                                           int *cluster, int *order, int *starts, int *n_windows, int *overflow, void *ws,
                                           size_t ws_bytes);
void pointops2_window_coord_launcher(int N, const float *xyz, const float *bbox6, float window, int shifted, float *wc);
void pointops2_sampled_buckets_launcher(int N, int m, const int *sample_idx, const int *l_order, const int *l_starts,
                                        const int *l_n_windows, int *sampled, int *ls, int *ls_starts, void *ws, size_t ws_bytes);
void pointops2_pairs_count_launcher(int N, const int *s_cluster, const int *s_starts, const int *l_cluster, const int *ls,
                                    const int *ls_starts, const float *wc, int *offsets, void *ws, size_t ws_bytes);
void pointops2_pairs_fill_launcher(int N, const float *xyz, float window, float quant, const int *s_cluster, const int *s_order,
                                   const int *s_starts, const int *l_cluster, const int *ls, const int *ls_starts, const float *wc,
                                   const int *offsets, int *index_0, int *index_1, int *rel_idx);

/* expands CSR offsets to the per-pair query id (index_0); segment bounds are clamped into [0, M], so offsets that do not
 * describe an M-pair list leave entries unwritten but never write outside index0[0, M) */
void csr_expand_launcher(int N, int M, const int *offsets, int *index0);

/* *bad (device int) = 0 iff offsets [N+1] describes the per-pair query ids index [M] exactly (offsets[0] = 0,
 * offsets[N] = M, ordered segments, every pair of segment i carries the id i), non-zero otherwise.  Reads nothing
 * outside offsets[0..N] and index[0..M) whatever the offsets hold.  index: int32 or int64 (index_is_int64); the model's
 * scatter_softmax call site (model/stratified_transformer.py:205) passes int64. */
void pointops2_csr_matches_launcher(int N, int M, const int *offsets, const void *index, int index_is_int64, int *bad);

/* ---- optional fused path (no counterpart in the reference's launcher set) -----------------------
 * attn[m,hh] = softmax over the query's pairs of (<q,k[j]> + <q,Tq(m)> + <k[j],Tk(m)>): A1 + A2 + add + A3 of
 * WindowAttention.forward (model/stratified_transformer.py:183-205) in one kernel.  d = 16 only; needs
 * pointops2_set_table_rows(L).  attn [M,h] is fully written (no zero-fill needed). */
void window_logits_softmax_forward_launcher(int N, int M, int h, int hdim, const float *q, const int *index_q_offsets,
                                             const float *k, const int *index_k, const float *table_q,
                                             const float *table_k, const int *rel_idx, float *attn);

/* Backward of the whole sequence for that module: grad_logit [M,h] (scratch/output, fully written), grad_q [N,h,16],
 * grad_k / grad_v [rows of k, h, 16] fully written; the three table gradients [L,h,16,3] are ACCUMULATED (zero-fill them).
 * attn = the forward's softmax output.  Needs pointops2_set_table_rows(L), pointops2_set_csc(...) and, when k / v have
 * other rows than q, pointops2_set_key_rows. */
void window_attention_backward_launcher(int N, int M, int h, int hdim, const float *grad_out, const float *q, const float *k,
                                        const float *v, const float *attn, const int *index0_offsets, const int *index1,
                                        const float *table_q, const float *table_k, const float *table_v,
                                        const int *rel_idx, float *grad_logit, float *grad_q, float *grad_k, float *grad_v,
                                        float *grad_table_q, float *grad_table_k, float *grad_table_v);

/* ---- window-centric ("cell") attention: SURVEY 8f-1, second step ------------------------------------------------
 * A cell = the queries of one (small window, large window) intersection of a block pattern; they share one candidate key
 * list (the small window's points, then the sampled points of the large window), so a cell is a dense n_q x n_k tile of
 * pairs (model/stratified_transformer.py:15-18, :20-38).  The plan of a pattern is built from the arrays the index build
 * already has (window partitions, bucketed samples, window coordinates):
 *   pass 1  pointops2_cell_plan_count_launcher  cells, their order, sizes and scans; counts[8] = {cells, tile entries P,
 *           key slots K, largest key count, parents (cells before the cut)} (device; the caller reads P and K to allocate the arrays of pass 2);
 *           max_queries > 0 cuts every cell into pieces of at most that many queries (a piece = one wave's unit of work)
 *   pass 2  pointops2_cell_plan_fill_launcher   key list per cell, owner cell per key slot, and per tile entry the packed
 *           rel-pos index r0 | r1 << 8 | r2 << 16 (model :186-190, clamped to [0, L)) with bit 31 set where the candidate is
 *           NOT a key of that query (same window coordinate, :34)
 * All arrays are caller-allocated device memory: cell_order, qcell, cell_perm [N]; cell_desc [4N]; cell_qstart, cell_kbase,
 * cell_pbase, parent_first [N+2]; counts [8]; cell_keys, kcell [K]; relp [P]. */
typedef struct pointops2_cell_plan {
    int n_points;            /* N */
    int n_cells;             /* host copy of counts[0] */
    int n_parents;           /* host copy of counts[4]: cells before the cut into pieces of max_queries */
    int n_pairs;             /* P = sum over cells of n_q * n_k (host copy of counts[1]) */
    int n_keyslots;          /* K = sum over cells of n_k       (host copy of counts[2]) */
    const int *counts;       /* device [8] */
    const int *parent_first; /* [N+2] first piece (cell id) of a parent; pieces of a parent are consecutive, their tiles contiguous */
    const int *cell_perm;    /* [N]   cell ids, largest tile first (first counts[0] entries) */
    const int *cell_qstart;  /* [N+2] first sorted query position of a cell */
    const int *cell_kbase;   /* [N+2] first key slot of a cell */
    const int *cell_pbase;   /* [N+2] first tile entry of a cell; entry (il, jl) is at pbase + il * n_k + jl */
    const int *cell_order;   /* [N]   query (point) ids grouped by cell, ascending inside a cell */
    const int *qcell;        /* [N]   cell of a sorted query position */
    const int *cell_keys;    /* [K]   key (point) ids per cell: dense keys ascending, then stratified candidates ascending */
    const int *kcell;        /* [K]   cell of a key slot */
    const unsigned int *relp;/* [P]   packed rel-pos index + "not a key" flag */
    int task_first;          /* the launch works on the cells cell_perm[task_first + i * task_step] only (one scene over several */
    int task_step;           /* ranks: rank r of w takes r, r + w, ... - cells are sorted by size, so the shares are balanced); */
                             /* 0 / 0 (or step 1): all cells.  A partial forward writes only its cells' rows of `out`, a partial */
                             /* backward only its queries' rows of grad_q: zero-fill them and sum over the ranks. */
    int table_rows;          /* L the packed rel-pos indices of relp were clamped to (pass 2): the attention launchers */
                             /* reject tables with any other row count (ABI version 2) */
    int max_queries;         /* the max_queries the plan was cut with in pass 1 (0 = uncut): sizes the kernels' query tiles */
    const int *task_list;    /* optional (NULL: all cells / the task_first, task_step share): the cell ids this launch works on, */
    const int *task_count;   /* device [1]: how many of them.  One scene over several ranks with cells assigned by OWNER of */
                             /* their first query (sharding.py, halo exchange): same zero-fill rules as a task_step share. */
} pointops2_cell_plan;
size_t pointops2_cell_plan_workspace_bytes(int N);
void pointops2_cell_plan_count_launcher(int N, int max_queries, const int *s_cluster, const int *s_starts, const int *l_cluster,
                                        const int *ls_starts, int *cell_order, int *qcell, int *cell_desc, int *cell_qstart,
                                        int *cell_kbase, int *cell_pbase, int *cell_perm, int *parent_first, int *counts, void *ws,
                                        size_t ws_bytes);
/* pass 1 in two halves, for a caller that builds beside the stage's sampler: `prepare` needs the two partitions only (cells = points
 * sorted by (small, large) window, their cut into pieces, the parents) and leaves its intermediate arrays in ws, which must stay untouched
 * until `sizes` - which needs the sampled points too (ls_starts) - has run on the same ws. */
void pointops2_cell_plan_prepare_launcher(int N, int max_queries, const int *s_cluster, const int *l_cluster, int *cell_order,
                                          int *parent_first, int *counts, void *ws, size_t ws_bytes);
void pointops2_cell_plan_sizes_launcher(int N, const int *s_cluster, const int *s_starts, const int *l_cluster, const int *ls_starts,
                                        const int *cell_order, int *qcell, int *cell_desc, int *cell_qstart, int *cell_kbase,
                                        int *cell_pbase, int *cell_perm, int *counts, void *ws, size_t ws_bytes);
void pointops2_cell_plan_fill_launcher(int N, const float *xyz, float window, float quant, int L, const int *s_order, const int *ls,
                                       const float *wc, const int *cell_order, const int *qcell, const int *cell_qstart,
                                       const int *cell_desc, const int *cell_kbase, const int *cell_pbase, int *cell_keys, int *kcell,
                                       unsigned int *relp);
/* The whole operator sequence of WindowAttention.forward (:183-208) on a cell plan, d = 16.  q, k, v [N,h,16]; tables [L,h,16,3];
 * out [N,h,16] fully written; pbuf [h, P] receives the softmax weights in tile order (kept for the backward); ml [N,h,2] is
 * scratch (running max / sum of rows longer than one register chunk). */
void cell_attention_forward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *q, const float *k,
                                     const float *v, const float *table_q, const float *table_k, const float *table_v, float *out,
                                     float *ml, float *pbuf);
/* Its backward.  out / pbuf = the forward's; gsbuf [h, P] scratch (receives the logit gradients in tile order); grad_q fully
 * written; grad_k, grad_v and the three table gradients are ACCUMULATED (zero-fill them).  L <= 80.
 * Both launchers record an error (pointops2_last_error) unless L == plan->table_rows. */
void cell_attention_backward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const float *q,
                                      const float *k, const float *v, const float *out, const float *table_q, const float *table_k,
                                      const float *table_v, const float *pbuf, float *gsbuf, float *grad_q, float *grad_k,
                                      float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v);

/* ---- the data-side step in front of the path ----
 * voxel keys of util/voxelize.py:46-59,79-84 (floor(coord / voxel), the FNV-style 64-bit hash of the three cells) and the
 * crop distances of util/data_util.py:188-191 (squared distance to the seed point), in the coordinates' own precision
 * (is_f64: coord / dist are double arrays, else float). */
void pointops2_voxel_keys_launcher(int N, int is_f64, const void *coord, double voxel, unsigned long long *keys);
void pointops2_crop_dist_launcher(int N, int is_f64, const void *coord, int seed, void *dist);

/* The same two entry points with q / k / v / tables STORED as bf16 (raw 16-bit patterns; BASELINE config 3's second leg).
 * Arithmetic, outputs (out, pbuf) and all gradients stay fp32: the reference's operators are fp32-only
 * (model/stratified_transformer.py:183,194,208 cast every operand with .float()), so this is an extension whose parity
 * target is "the fp32 path on the bf16-rounded operands" (identical up to summation order). */
void cell_attention_forward_bf16_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const uint16_t *q, const uint16_t *k,
                                          const uint16_t *v, const uint16_t *table_q, const uint16_t *table_k, const uint16_t *table_v,
                                          float *out, float *ml, float *pbuf);
void cell_attention_backward_bf16_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const uint16_t *q,
                                           const uint16_t *k, const uint16_t *v, const float *out, const uint16_t *table_q,
                                           const uint16_t *table_k, const uint16_t *table_v, const float *pbuf, float *gsbuf, float *grad_q,
                                           float *grad_k, float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v);

#ifdef __cplusplus
}
#endif
#endif /* POINTOPS2_HIP_H */
