/*
 * ngcf_hip.h - C ABI of libngcf_hip.so, the MI355X (gfx950) NGCF embedding-propagation engine.
 *
 * This is the drop-in boundary for ONE path of haesungpyun/seoul_tourism_recommendation_NGCF:
 * the body of `NGCF.forward` (model/NGCF.py:102-156) and `BPR.forward` (model/bprloss.py:15-22).
 * The reference is pure Python on top of PyTorch, so "what its FFI would bind" is the set of
 * tensor ops it issues on that path; each entry point below names the reference lines it replaces.
 * The Python mirror of the reference's nn.Module surface (seoul_tourism_recommendation_ngcf_amd/
 * NGCF.py, bprloss.py) calls these through ctypes with `tensor.data_ptr()` and
 * `torch.cuda.current_stream().cuda_stream`; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.
 *   - every `const float*` / `float*` / index pointer is a DEVICE pointer owned by the caller
 *     unless a comment says "host".  The library allocates only ngcf_csr_t objects.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Compute entry points
 *     are asynchronous on that stream; ngcf_csr_* builders synchronise it once (one-time set-up).
 *   - return 0 on success, non-zero on error; ngcf_last_error() gives the message (thread-local).
 *     The Python mirror raises RuntimeError (IndexError for NGCF_ERR_INDEX), like torch does
 *     at the same call sites in the reference.
 *   - row-major fp32 matrices with an explicit leading dimension `ld*` in elements.
 */
#ifndef NGCF_HIP_H
#define NGCF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGCF_OK            0
#define NGCF_ERR_ARG       1   /* bad argument (null pointer, negative size, unsupported width) */
#define NGCF_ERR_HIP       2   /* a HIP runtime call failed */
#define NGCF_ERR_INDEX     3   /* an index is out of range (IndexError in the Python mirror) */
#define NGCF_ERR_WORKSPACE 4   /* caller's workspace is too small */

typedef struct ngcf_csr ngcf_csr_t;

/* ---- library ------------------------------------------------------------------------- */
const char *ngcf_last_error(void);
/* "gfx950" - the only architecture the code objects are built for. */
const char *ngcf_target_arch(void);
/* ABI version of this header.  ngcf_version() returns the value the library was built with; the Python mirror refuses to bind
 * a library whose version differs (a stale .so would otherwise receive shifted arguments). */
#define NGCF_ABI_VERSION 7
int ngcf_version(void);

/* Tunables of the kernel dispatch (thresholds, lab switches).  The library reads its NGCF_* environment variables ONCE, in
 * ngcf_options_from_env() on first use (no launch path touches the environment); call it again to re-read them, or set single
 * options by name (the variable's name without the NGCF_ prefix, lower case: "dense_resident", "swept_lead", ...; the table
 * is NgcfOptions in csrc/common.h).  Unknown names are an error.  Not thread-safe against concurrent launches. */
int ngcf_options_from_env(void);
int ngcf_set_option(const char *name, int64_t value);
int ngcf_set_option_str(const char *name, const char *value);

/* Timing of the dominant kernel for bench.py's roofline line: while enabled, every SpMM kernel launch is
 * bracketed by a hipEvent pair on its own stream.  ngcf_prof_collect waits for them and returns the number
 * of timed launches and their summed duration (ms), then resets the recorder.  Not thread-safe. */
int ngcf_prof_enable(int on);
int ngcf_prof_collect(int64_t *n_launches, double *total_ms);

/* ---- Laplacian: COO (the layout of `lap_list[k]`, matrix.py:79-83) -> CSR -------------- */
/*
 * Replaces the per-call `self.lap_list[year_idx].to(device)` + COO SpMM set-up (NGCF.py:118,130):
 * built once per year slice.  `rows/cols` are int64[nnz] (the two rows of `_indices()`), `vals`
 * fp32[nnz] (`_values()`), all on the device.  Entries need not be coalesced; duplicates are kept
 * as separate entries, like the reference's CPU `torch.mm` which FMAs each stored entry.
 * Row-sorted input (what matrix.py emits) is converted on the device; anything else is stably
 * sorted by row on the host.  `n_rows` x `n_cols` is the shape of this slab: a full Laplacian has
 * n_rows == n_cols == N; a row slab of a row-partitioned graph has n_rows < n_cols and row ids
 * relative to the slab.  Columns are stored as int32 (n_cols < 2^31).
 * Fails with NGCF_ERR_INDEX when a row/col id is out of range.
 */
int ngcf_csr_from_coo(const int64_t *rows, const int64_t *cols, const float *vals, int64_t nnz,
                      int64_t n_rows, int64_t n_cols, ngcf_csr_t **out, void *stream);
/* Adopt device CSR arrays (rowptr int64[n_rows+1], colidx int32[nnz], vals fp32[nnz]); they are
 * borrowed and must outlive the handle. */
int ngcf_csr_from_arrays(const int64_t *rowptr, const int32_t *colidx, const float *vals,
                         int64_t n_rows, int64_t n_cols, int64_t nnz, ngcf_csr_t **out, void *stream);
/* Re-plan the row segmentation: rows with more than `seg_len` stored entries are cut into
 * segments of `seg_len` entries whose partial sums are combined in a fixed order (no atomics).
 * The constructors plan with a default that grows with the matrix: the power of two in [64, 2048]
 * that gives about 4 096 segments or more (2048 from 4 M stored entries up). */
int ngcf_csr_plan(ngcf_csr_t *csr, int32_t seg_len, void *stream);
/* SpMM kernel choice.  0 = row-wise gather kernels, d-sliced on the row groups whose gathered table is small
 * (default: right for matrices that live for one product, e.g. the per-layer node-dropout matrices); 1 = row-wise
 * kernels without d-slicing; 2 = L2-swept kernel (csrc/spmm_swept.hip) on every row group whose shape allows it
 * (tests); 3 = L2-swept kernel on the row groups where its host-side plan expects enough L2 re-use to pay, the
 * row-wise kernels on the rest - meant for long-lived matrices (the Laplacians of `lap_list`): building the plan
 * costs a host pass over the entries and as much device memory again as the CSR.  Calls whose width is not a
 * multiple of 64, that use edge dropout, or whose table spans more than 4 GiB use the row-wise kernels anyway.
 * A CSR with swept parts keeps a block of sweep counters per launching stream (up to four): products of the same CSR may run
 * concurrently on different streams (beyond four streams the extra ones share a block: results stay correct - the counters
 * only pace the sweep - but those products lose L2 re-use).  The stream registry is not thread-safe: launch a given CSR from one
 * host thread. */
int ngcf_csr_set_mode(ngcf_csr_t *csr, int mode, void *stream);
/*
 * Thinned copy of a CSR, made on the device: *dst keeps, in src's order, every stored entry e of `src` with
 * keep[map ? map[e] : e] != 0 (`keep`: device uint8, one flag per entry of the matrix the flags were drawn for; `map`: device
 * int32[src nnz] or NULL = identity).  This is the reference's per-layer `sparse_dropout` (NGCF.py:93-100,124-126: a COO tensor
 * rebuilt from `indices[:, mask]`, values not rescaled) as one stream compaction: no host round trip, no synchronisation, and
 * no allocation after the first call - pass the previous step's object back in *dst and its buffers are re-used (*dst == NULL:
 * a new object, sized for all of src's entries).  `nnz_kept`: the number of set flags if the caller knows it (it drew the
 * mask), else -1: ngcf_csr_nnz(*dst) is then an upper bound.  The copy runs on the row-wise kernels (mode 0) with src's
 * segment structure and row groups and BORROWS them: `src` must outlive `*dst`.
 * ngcf_csr_filter_pos(dst): device int32[src nnz + 1], pos[e] = kept entries before e (the new position of a kept entry).
 */
int ngcf_csr_filter(const ngcf_csr_t *src, const uint8_t *keep, const int32_t *map, int64_t nnz_kept, ngcf_csr_t **dst,
                    void *stream);
const int32_t *ngcf_csr_filter_pos(const ngcf_csr_t *dst);
/*
 * Entry map of a thinned transpose.  L^T is thinned with the flags drawn for L through map[j] = position in L of entry j of
 * L^T; after both were filtered, the next layer needs the same map between the two thinned matrices:
 * map_out[pos_t[j]] = pos_l[map[j]] for every kept j, with pos_t = ngcf_csr_filter_pos(dst_t), pos_l = that of the thinned L.
 * n_src_entries: entries of the transpose that was filtered into dst_t.
 */
int ngcf_csr_filter_remap(const ngcf_csr_t *dst_t, const uint8_t *keep, const int32_t *map, int64_t n_src_entries,
                          const int32_t *pos_l, int32_t *map_out, void *stream);
void ngcf_csr_free(ngcf_csr_t *csr);
int64_t ngcf_csr_nnz(const ngcf_csr_t *csr);
int64_t ngcf_csr_n_rows(const ngcf_csr_t *csr);
int64_t ngcf_csr_n_cols(const ngcf_csr_t *csr);
int64_t ngcf_csr_n_segments(const ngcf_csr_t *csr);
int64_t ngcf_csr_max_row_len(const ngcf_csr_t *csr);        /* stored entries of the longest row */
/* rows currently covered by L2-swept parts (0: every product of this CSR runs on the row-wise kernels: mode 0/1, a shape the
 * plan declines - too small, expected re-use below 3, table beyond 32-bit offsets - or a device that does not report 256 CUs,
 * for which one note is printed on stderr) */
int64_t ngcf_csr_swept_rows(const ngcf_csr_t *csr);
/* device pointers of the CSR arrays (for tests / the transposed view) */
const int64_t *ngcf_csr_rowptr(const ngcf_csr_t *csr);
const int32_t *ngcf_csr_colidx(const ngcf_csr_t *csr);
const float *ngcf_csr_vals(const ngcf_csr_t *csr);

/* ---- propagation ----------------------------------------------------------------------- */
/* Bytes of workspace ngcf_spmm_csr_f32 / ngcf_layer_fused_f32 need for this CSR at width d_in
 * (segment partial sums; for the fused layer also the L.E tile and the packed weights). */
int64_t ngcf_spmm_workspace_bytes(const ngcf_csr_t *csr, int d);
int64_t ngcf_layer_workspace_bytes(const ngcf_csr_t *csr, int d_in, int d_out);

/*
 * LE = L.E  (NGCF.py:130, `torch.mm(L, E)`).
 * E: [n_cols, d] with leading dimension ldE; LE: [n_rows, d] with ldLE.  fp32 FMA accumulation;
 * the summation order inside a row differs from the reference's sequential order (tolerance in
 * tests/test_parity_gpu.py).
 */
int ngcf_spmm_csr_f32(const ngcf_csr_t *csr, const float *E, int64_t ldE, int d, float *LE,
                      int64_t ldLE, void *workspace, int64_t workspace_bytes, void *stream);

/* Width to run a product at when LE has padded rows (leading dimension a multiple of 4 >= d rounded up): a width that is not
 * a multiple of 4 on a small (launch-bound) matrix is multiplied up to the next multiple of 4 when the gathered rows are
 * 16-byte aligned and padded - the extra columns land in the padding of LE.  ngcf_layer_fused_f32 applies this rule itself;
 * callers that run the SpMM on its own (the training path) ask here so that both paths produce the same bits. */
int ngcf_spmm_product_width(const ngcf_csr_t *csr, const float *E, int64_t ldE, int d);

/*
 * LE = drop(L).E with node dropout on the device (NGCF.py:93-100,124-126 semantics: every stored entry is kept
 * with probability 1-p, values are NOT rescaled, and the thinning is cumulative over layers): the entry (i, j) of L
 * survives iff hash(seeds[q], i, j) passes for every q < n_seeds (layer k passes its own and all earlier layers' seeds,
 * n_seeds <= 4, host array).  No CSR is rebuilt.  The hash is keyed by the entry's row and column in L, so any layout of
 * L is thinned the same way: pass `transposed` != 0 when `csr` holds L^T (the backward pass), and L^T loses exactly the
 * entries L lost.  (Entries stored twice at the same (i, j) share their fate.)  The mask is a counter-based hash, not
 * torch's generator: same distribution as the reference, different stream.
 */
int ngcf_spmm_csr_dropout_f32(const ngcf_csr_t *csr, const float *E, int64_t ldE, int d, float *LE, int64_t ldLE,
                              float drop_p, const uint64_t *seeds, int n_seeds, int transposed,
                              void *workspace, int64_t workspace_bytes, void *stream);

/*
 * One whole propagation layer (NGCF.py:130-146) for the rows of `csr`:
 *   LE    = L.E_gather                                              NGCF.py:130
 *   M     = (LE+E_self).W1^T + (LE*E_self).W2^T + (2*b1 + b2)       NGCF.py:131-138 (b1 twice)
 *   carry = dropout_p(leaky_relu(M, slope))                         NGCF.py:140-142
 *   norm  = carry / max(||carry||_2, 1e-12)                         NGCF.py:144
 * E_gather: [n_cols, d_in] table the neighbours are read from; E_self: [n_rows, d_in], the same
 * rows as the output (E_gather + row_start*ld for a row slab).  W1, W2: [d_out, d_in] row-major
 * (nn.Linear.weight), b1, b2: [d_out].  `carry` ([n_rows, d_out], may be NULL for the last
 * layer) feeds the next layer; `norm` is written straight into its column block of all_E
 * (pointer already offset by the block's first column, leading dimension ldn = ld of all_E),
 * which removes the reference's `torch.cat` (NGCF.py:147).  drop_p == 0 -> no dropout (eval).
 * Message dropout (NGCF.py:142), two forms: `drop_mask` != NULL ([n_rows, d_out], leading dimension ld_mask) is the
 * noise tensor nn.Dropout multiplies by (0 or 1/(1-p)), drawn by the caller - the mirror draws it from torch's CPU
 * generator exactly where the reference does, so the zero pattern is bit-identical ("reference" mode); drop_mask == NULL
 * and drop_p > 0: keep mask = counter-based hash of (drop_seed, row, column) evaluated in the epilogue ("device" mode).
 * One call, two phases on the stream: the SpMM writes LE [n_rows, d_in] into the workspace and the dense phase (everything from
 * the nn.Linear contractions to both outputs in one kernel) reads it back; what is fused is the dense half and its epilogue, not
 * the SpMM into it (DESIGN.md 4.2 says why).
 */
int ngcf_layer_fused_f32(const ngcf_csr_t *csr, const float *E_gather, int64_t ldEg,
                         const float *E_self, int64_t ldEs, int d_in,
                         const float *W1, const float *b1, const float *W2, const float *b2, int d_out,
                         float leaky_slope, float drop_p, uint64_t drop_seed,
                         const float *drop_mask, int64_t ld_mask,
                         float *carry, int64_t ldc, float *norm, int64_t ldn,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* The dense half alone (NGCF.py:131-146) on an already computed LE; same arguments as above.
 * Needs ngcf_dense_workspace_bytes(d_in, d_out) bytes of workspace (packed weights). */
int64_t ngcf_dense_workspace_bytes(int d_in, int d_out);
int ngcf_layer_dense_f32(const float *LE, int64_t ldLE, const float *E_self, int64_t ldEs,
                         int64_t n_rows, int d_in,
                         const float *W1, const float *b1, const float *W2, const float *b2, int d_out,
                         float leaky_slope, float drop_p, uint64_t drop_seed,
                         const float *drop_mask, int64_t ld_mask,
                         float *carry, int64_t ldc, float *norm, int64_t ldn,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* dst[r, 0:d] = src[r, 0:d] for r < n_rows (strided copy; writes E0 into its block of all_E,
 * NGCF.py:120-121 + 147). */
int ngcf_copy_rows_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t n_rows, int d,
                       void *stream);
/* The same with two destinations (the source rows are read once): E0 goes to its block of all_E and, when the rows of
 * all_E are not 16-byte aligned (embed_size 65 / 130 / 515, NGCF.py:39-43), to the aligned copy the first layer gathers from. */
int ngcf_copy_rows2_f32(const float *src, int64_t lds, float *dst, int64_t ldd, float *dst2, int64_t ldd2,
                        int64_t n_rows, int d, void *stream);

/* dst[idx[b], 0:d] = src[idx[b], 0:d] for b < n_idx, ids outside [0, n_rows) skipped (r04): block 0 of a retained all_E follows
 * the rows the feature injection rewrote (NGCF.py:114-115) without the whole table being copied again.  idx: int64, device. */
int ngcf_copy_rows_indexed_f32(const float *src, int64_t lds, float *dst, int64_t ldd, const int64_t *idx, int64_t n_idx,
                               int64_t n_rows, int d, void *stream);

/* HOST routine (r04): n draws of torch's CPU `bernoulli_(keep)` - what `nn.Dropout` draws on a CPU tensor, NGCF.py:93-100,142 -
 * from the generator whose state bytes (`torch.get_rng_state()`, >= 5016 bytes, legacy mt19937 layout) are `rng_state`; the bytes
 * are advanced in place exactly as torch advances its generator (two 32-bit outputs per element).  flags[i] (uint8, may be NULL) =
 * kept, noise[i] (float, may be NULL) = kept ? scale : 0, *n_kept = count.  Host pointers.  Bit-identical to torch's own serial
 * kernel (checked against it by the caller once per process), several times faster (vector loops, no lock per element). */
int ngcf_torch_cpu_bernoulli(uint8_t *rng_state, int64_t state_bytes, int64_t n, double keep, uint8_t *flags, float *noise,
                             float scale, int64_t *n_kept);

/* The draws of a whole forward in ONE call (r04; a helper thread draws the next forward's masks ahead and must not need the
 * interpreter between two draws): draw i is ngcf_torch_cpu_bernoulli(n[i], keep[i], flags[i], noise[i], scale[i]) -> n_kept[i], one
 * after the other on the same state bytes; n[i] = -1 stands for "as many as the latest earlier draw WITH flags kept" (the node
 * mask of a layer has one flag per entry the earlier layers kept, NGCF.py:126: its buffer must hold the first such draw's n). */
int ngcf_torch_cpu_bernoulli_seq(uint8_t *rng_state, int64_t state_bytes, int n_draws, const int64_t *n, const double *keep,
                                 uint8_t *const *flags, float *const *noise, const float *scale, int64_t *n_kept);

/* ---- feature injection (NGCF.py:103-115) ---------------------------------------------- */
/*
 * user_w[u_id[b], :] = user_w[u_id[b], :]*(1-r) + cat(age,sex,month,day,dow rows)[b, :]*r.
 * tables[5]: device pointers of the five [card, fw] tables in the concat order age, sex, month,
 * day, dow (NGCF.py:110); idx[5]: the five int64[B] index vectors in the same order; cards[5] their
 * cardinalities (host).  5*fw must equal d0 (the reference raises RuntimeError otherwise).
 * Duplicate u_id: the LAST occurrence in the batch wins (what the CPU index_put_ of the reference
 * does).  `scratch` is int32[n_user], all -1 on entry and on exit.
 * `status` is a device int32: set non-zero when an index is out of range (rows skipped).
 */
int ngcf_feature_inject_f32(float *user_w, int64_t ldu, int64_t n_user, int d0,
                            const float *const *tables, const int64_t *const *idx, const int64_t *cards,
                            int fw, const int64_t *u_id, int64_t B, double emb_ratio,
                            int32_t *scratch, int32_t *status, void *stream);

/* ---- dropout seeds on the device ---------------------------------------------------------- */
/* Every 64-bit dropout seed of this interface (drop_seed of the layer entry points, the `seeds` of the node-dropout products) is
 * either a value below 2^62 or a TAGGED DEVICE ADDRESS: top 16 bits 0xD5ED, low 48 bits the address of a uint64_t that holds the
 * value - read by the kernels when they run.  A captured hipGraph bakes kernel arguments in; with its seeds behind addresses every
 * replay still draws new masks.  ngcf_seeds_advance steps n such words to their next values (a splitmix64 chain, results below
 * 2^62), asynchronously on `stream`. */
#define NGCF_SEED_PTR_TAG 0xD5EDull
int ngcf_seeds_advance(uint64_t *seeds_device, int n, void *stream);

/* ---- gathers (NGCF.py:151-155) ---------------------------------------------------------- */
/* out[b, 0:d] = table[(row_off + idx[b]), 0:d], bit-exact copies.  idx must lie in [0, n_idx_rows);
 * offenders are skipped and *status (device int32) is set non-zero. */
int ngcf_gather_rows_f32(const float *table, int64_t ld, int d, const int64_t *idx, int64_t B,
                         int64_t row_off, int64_t n_idx_rows, float *out, int64_t ldo,
                         int32_t *status, void *stream);

/* The three gathers of NGCF.forward (NGCF.py:151-155: users, positive items, negative items) from one table in ONE launch:
 * out_k[b, 0:d] = table[(row_off_k + idx_k[b]), 0:d] for k = 0, 1, 2; a set with B_k == 0 is skipped (no negative items:
 * NGCF.py:153).  Same bounds rule and status word as ngcf_gather_rows_f32; the three outputs share the leading dimension. */
int ngcf_gather_rows3_f32(const float *table, int64_t ld, int d,
                          const int64_t *idx0, int64_t B0, int64_t row_off0, int64_t n_idx_rows0, float *out0,
                          const int64_t *idx1, int64_t B1, int64_t row_off1, int64_t n_idx_rows1, float *out1,
                          const int64_t *idx2, int64_t B2, int64_t row_off2, int64_t n_idx_rows2, float *out2,
                          int64_t ldo, int32_t *status, void *stream);

/* ---- BPR (bprloss.py:15-22) ------------------------------------------------------------- */
/*
 * loss = (-sum_r logsigmoid(|u_r.p_r| - |u_r.n_r|) + wd*(sum|u|^2 + sum|p|^2 + sum|n|^2)) / batch_size
 * u: [Bu, D], p: [Bp, D], n: [Bn, D] contiguous; each row count is 1 (broadcast) or R = max.
 * The squared norms run over each tensor's own rows.  `loss` is one device float.
 * workspace: ngcf_bpr_workspace_bytes(R) bytes.  Deterministic (fixed-order two-stage reduction).
 */
int64_t ngcf_bpr_workspace_bytes(int64_t R);
int ngcf_bpr_fused_f32(const float *u, int64_t Bu, const float *p, int64_t Bp, const float *n, int64_t Bn,
                       int D, float weight_decay, float batch_size, float *loss,
                       void *workspace, int64_t workspace_bytes, void *stream);

/* ---- backward pass (`loss.backward()` of experiment.py:57; driven by autograd.py) -------------- */
/* Everything `loss.backward()` runs: no library GEMM at any width.  L^T.dLE re-uses ngcf_spmm_csr_f32 on the CSR of L^T. */
/* du/dp/dn of the BPR loss (bprloss.py:15-22) times the upstream scalar *grad_out (device). */
int ngcf_bpr_backward_f32(const float *u, int64_t Bu, const float *p, int64_t Bp, const float *n, int64_t Bn, int D,
                          float weight_decay, float batch_size, const float *grad_out, float *du, float *dp,
                          float *dn, void *stream);
/* Backward of the row gathers (NGCF.py:151-155): the gradient rows g [M, d] of the gathered positions, summed per distinct row
 * of all_E in a fixed order: out[r, :] = sum of g[order[j], :] for j in [segptr[r], segptr[r+1]), in that order (`order`: the
 * gathered positions sorted by row, stable - duplicates add up in batch order; int64 device arrays).  No atomics. */
int ngcf_segment_sum_rows_f32(const float *g, int64_t ldg, int d, const int64_t *order, const int64_t *segptr, int64_t n_seg,
                              const int64_t *dst_rows, const int64_t *n_seg_dev, float *out, int64_t ldo, int64_t n_out_rows,
                              void *stream);
/* (dst_rows != NULL: the sum of segment r goes to row dst_rows[r] of `out` - the scatter into a dense, zero-filled gradient of
 * all_E with n_out_rows rows; a segment whose destination lies outside [0, n_out_rows) is skipped (r04: memory-safe whatever the
 * caller's index-check cadence); n_seg_dev != NULL: only the first *n_seg_dev segments exist (the count ngcf_rows_sort_unique left on the device: no
 * host round trip between the two launches), n_seg is then an upper bound that sizes the grid) */
/* The distinct rows among M <= 8 192 gathered positions, in one launch: idx int64[M] (rows of all_E, each < 2^50) ->
 * order int64[M] (the positions 0..M-1 sorted by row, equal rows in batch order), rows int64[<= M] (distinct, ascending),
 * segptr int64[<= M + 1] (group bounds inside `order`), n_rows int64[1].  All device arrays sized for M (segptr M + 1).
 * Feeds ngcf_segment_sum_rows_f32; replaces torch.unique + sort + cumsum (a dozen library launches) on the training step. */
int ngcf_rows_sort_unique(const int64_t *idx, int64_t M, int64_t max_row, int64_t *order, int64_t *rows, int64_t *segptr,
                          int64_t *n_rows, void *stream);
/* (max_row: the largest valid row, or -1 = unknown / unchecked: below 2^19 the sort runs on 32-bit keys.  r04: with max_row >= 0 an id
 * outside [0, max_row] - which the forward gather clamped and flagged in its status word - is sorted as ONE sentinel row
 * max_row + 1 behind the valid ones, forms the last segment of `order` and is not counted in n_rows.) */
/* Backward of normalise + dropout + LeakyReLU (NGCF.py:140-144): dM from dN (gradient of the all_E block; NULL = zero:
 * the rows no gather touched), dC (gradient of the carry from the next layer, may be NULL; not both) and the saved carry C. */
int ngcf_layer_bwd_pre_f32(const float *dN, int64_t ldn, const float *dC, int64_t ldc, const float *C, int64_t ldC,
                           int64_t n_rows, int d, float leaky_slope, float drop_p, uint64_t drop_seed,
                           const float *drop_mask, int64_t ld_mask, const int64_t *row_ids, float *dM, int64_t ldm, void *stream);
/* (row_ids: NULL, or the matrix rows the n_rows compacted rows stand for - they index the hash stream of device-mode dropout) */
/* out = init + L^T . X for a ROW-SPARSE X (the last layer's backward: dLE is non-zero on the <= 3 B gathered rows only).
 * csr_t: the CSR of L^T ([N, N]); slot: device int32[N], slot[r] = row of the compact X [R, d] that holds matrix row r, or -1
 * (zero row); init: compact [R, d] or NULL, added to the rows with slot >= 0 (the direct part of the gradient).
 * out[c, :] = (slot[c] >= 0 ? init[slot[c], :] : 0) + sum over the stored entries (c, r, v) of csr_t with slot[r] >= 0 of
 * v * X[slot[r], :], in entry order: a fixed summation order, no atomics; every row of out [N, d] is written.
 * drop_p / seeds: device-side node dropout as in ngcf_spmm_csr_dropout_f32 (csr_t is walked as the transpose).
 * workspace: ngcf_spmm_workspace_bytes(csr_t, min(d, 512)) bytes (partial sums of the cut rows). */
int ngcf_spmm_t_rows_f32(const ngcf_csr_t *csr_t, const int32_t *slot, const float *X, int64_t ldx, int d, const float *init,
                         int64_t ldi, float *out, int64_t ldo, float drop_p, const uint64_t *seeds, int n_seeds,
                         void *workspace, int64_t workspace_bytes, void *stream);
/* Input gradients of a layer's dense half in one MFMA kernel: dS = dM.W1, dP = dM.W2 (never stored), dLE = dS + dP*E,
 * dE_direct = dS + dP*LE (NGCF.py:131-136 differentiated).  dM: [n_rows, d_out], any d_out >= 1 (65 in the reference's own
 * configuration), with 16-byte aligned rows padded to a multiple of 4 floats (the padding is never used); W1, W2: [d_out, d_in] row-major (nn.Linear.weight); LE, E, dLE, dE: [n_rows, d_in]. */
int64_t ngcf_layer_bwd_input_workspace_bytes(int d_out);
int ngcf_layer_bwd_input_f32(const float *dM, int64_t ldM, int64_t n_rows, int d_out, const float *W1, const float *W2,
                             int d_in, const float *LE, int64_t ldLE, const float *E, int64_t ldE, float *dLE, int64_t ldd,
                             float *dE, int64_t lde, void *workspace, int64_t workspace_bytes, void *stream);
/* weight gradients of one layer on the fp32 matrix cores: gW1 [d_out, d_in] = dM^T . (LE + E) (the gradient of W1,
 * NGCF.py:131-133), gW2 [d_out, d_in] = dM^T . (LE * E) (W2, NGCF.py:135-136), each row-major with its own leading dimension
 * (so a caller tiling a wide layer passes sub-blocks of the full gradients); d_in, d_out <= 128 per call.  Fixed summation
 * order (per-workgroup partials in the workspace, added in workgroup order).  gb2 / gb1 (may be NULL): [d_out] column sums of dM
 * from the same pass and twice that - the gradients of W2's and W1's bias (b1 enters the layer twice, NGCF.py:131,133). */
int64_t ngcf_bwd_weight_workspace_bytes(void);
int ngcf_layer_bwd_weight_f32(const float *dM, int64_t ldM, const float *LE, int64_t ldLE, const float *E, int64_t ldE,
                              int64_t n_rows, int d_in, int d_out, float *gW1, int64_t ld1, float *gW2, int64_t ld2,
                              float *gb1, float *gb2, void *workspace, int64_t workspace_bytes, void *stream);
/* out[r, 0:d] += add[r, 0:d] */
int ngcf_add_rows_f32(float *out, int64_t ldo, const float *add, int64_t lda, int64_t n_rows, int d, void *stream);

/* ---- top-k of score rows (experiment.py:104-111, demo.py:234-235: `torch.topk(torch.mm(u, items.T), k)`) ---- */
/* out_val/out_idx [n_rows, k]: the k largest entries of every row of `scores` [n_rows, n_cols] in descending
 * order (equal values: lowest column first; NaN sorts above +inf like torch.topk).  1 <= k <= min(n_cols, 1024).
 * The score matrix is a plain GEMM and is left to the caller. */
int ngcf_topk_rows_f32(const float *scores, int64_t ld, int64_t n_rows, int64_t n_cols, int k, float *out_val,
                       int64_t *out_idx, void *stream);
/* Score-and-select in one launch (experiment.py:93,104-109; demo.py:233-235): scores[b, i] = u[b, :] . items[i, :] for every
 * item, then top-k per user row (values descending, ties lowest item first), without a library GEMM.  u: [B, D] (ldu),
 * items: [n_items, D] (ldi; e.g. all_items_emb, a row range of all_E), scratch: [B, ld_scratch >= n_items] floats of caller
 * memory that receives the score matrix (it is what `torch.mm` would have returned), out_val [B, k], out_idx [B, k] int64. */
int ngcf_recommend_topk_f32(const float *u, int64_t ldu, int64_t B, const float *items, int64_t ldi, int64_t n_items,
                            int D, int k, float *scratch, int64_t ld_scratch, float *out_val, int64_t *out_idx, void *stream);

/* ---- multi-GPU row partition (new design; host-only helper) --------------- */
/*
 * Cut rows [row_begin, row_end) into `world` contiguous ranges of roughly equal stored-entry
 * count.  rowptr is a HOST int64[n_rows+1].  bounds is a HOST int64[world+1].
 */
int ngcf_shard_plan(const int64_t *rowptr_host, int64_t row_begin, int64_t row_end, int world,
                    int64_t *bounds_host);

/*
 * The exchange step between two layers of the row-partitioned engine (the reference is
 * single-device, NGCF.py has no counterpart): recv[q*rows_per_rank + k, :] = rank q's send[k, :] for every rank q of
 * the communicator - one ncclAllGather (RCCL over xGMI) of rows_per_rank*d floats per rank, asynchronous on `stream`.
 * `nccl_comm` is an ncclComm_t passed as void* (the Python mirror hands over its process group's communicator);
 * send: [rows_per_rank, d] contiguous, recv: [n_ranks*rows_per_rank, d] contiguous, both device pointers.  With the
 * padded rank-major numbering of dist.ShardLayout the received block IS the replica the next layer gathers from.
 * The library binds to the librccl already loaded in the process (it does not link one).
 */
int ngcf_allgather_rows(void *nccl_comm, const float *send, float *recv, int64_t rows_per_rank, int d, void *stream);
/* number of ranks of the communicator (host call) */
int ngcf_comm_size(void *nccl_comm, int *n_ranks);

/*
 * CU-free exchange between the ranks of one node (r03): copy engines move the bytes, host threads do the waiting, no kernel of
 * the exchange ever occupies a CU - the L2-swept SpMM keeps the chip to itself while the previous step's rows travel.
 * Every rank owns one exchange buffer; its producers write the rows the others need into it; the others PULL:
 *   ngcf_p2p_create   allocate the buffer (current device) and map the node's shared sequence words (POSIX shm `shm_name`, the
 *                     same name on every rank, unique per job)
 *   ngcf_p2p_handle   64-byte IPC handle of the buffer; the caller all-gathers the handles (host, any transport)
 *   ngcf_p2p_connect  open the peers' buffers (handles: world x 64 bytes, rank-major)
 *   ngcf_p2p_publish  on `stream`: everything enqueued so far is finished and visible to other devices when peers see `seq`
 *   ngcf_p2p_pull     host-wait (bounded) for rank `peer`'s `seq` in `slot`, then enqueue dst <- peer buffer[src_off, +bytes)
 *                     on this rank's copy stream for that peer (device-to-device: an SDMA engine across xGMI)
 *   ngcf_p2p_ack / ngcf_p2p_wait_acks   the reverse notice: a producer overwrites a region only after all peers have read it
 *   ngcf_p2p_fence    later copies start only after what `stream` holds now (the last readers of their destinations)
 *   ngcf_p2p_join     `stream` waits (stream dependency, no kernel) for every copy enqueued since the last join
 *   ngcf_p2p_stats    host time spent inside those waits so far, number of waits, number that found their word missing
 * `seq` must grow from call to call per slot (64 slots).  A wait that exceeds timeout_ms fails with NGCF_ERR_HIP.
 * Why the waiting is done by host threads and the publishing by a non-blocking host function (r04, tools/memops_lab.hip,
 * profiles/r04_memops_lab.txt): on this runtime hipStreamWriteValue64 / hipStreamWaitValue64 on host-registered memory, 8-byte
 * copies and event-less flag kernels are all SHADER work - beside a kernel that holds every CU (the L2-swept SpMM) none of them
 * starts before it ends (2.6 ms late beside a 2.65 ms kernel; a host function: 24 us), and a pending wait-value occupies a CU
 * (a one-workgroup-per-CU kernel launched behind it takes twice as long); a BLOCKING host function on a copy stream would take
 * the host out of the path, but the runtime runs every stream's host functions on one thread: a blocked one holds the
 * publishing ones of the same process (released only by its time-out).
 * ngcf_sum_slots_f32: out = slots[0] + slots[1] + ... (fixed order): the owner's side of a reduce-scatter done with pulls.
 */
typedef struct ngcf_p2p ngcf_p2p_t;
int ngcf_p2p_create(int rank, int world, int64_t bytes, const char *shm_name, ngcf_p2p_t **out);
void ngcf_p2p_destroy(ngcf_p2p_t *p2p);
int ngcf_p2p_handle(ngcf_p2p_t *p2p, void *handle64_host);
int ngcf_p2p_connect(ngcf_p2p_t *p2p, const void *handles_host);
void *ngcf_p2p_local(ngcf_p2p_t *p2p);
int64_t ngcf_p2p_bytes(const ngcf_p2p_t *p2p);
int ngcf_p2p_publish(ngcf_p2p_t *p2p, int slot, uint64_t seq, void *stream);
int ngcf_p2p_pull(ngcf_p2p_t *p2p, int peer, int slot, uint64_t seq, int64_t src_off, void *dst, int64_t bytes, double timeout_ms);
int ngcf_p2p_ack(ngcf_p2p_t *p2p, int peer, int slot, uint64_t seq);
int ngcf_p2p_wait_acks(ngcf_p2p_t *p2p, int slot, uint64_t seq, double timeout_ms);
int ngcf_p2p_fence(ngcf_p2p_t *p2p, void *stream);
int ngcf_p2p_join(ngcf_p2p_t *p2p, void *stream);
int ngcf_p2p_stats(ngcf_p2p_t *p2p, double *blocked_ms, int64_t *waits, int64_t *waits_blocked, int reset);
int ngcf_sum_slots_f32(const float *slots, int64_t slot_stride, int n_slots, int64_t n, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NGCF_HIP_H */
