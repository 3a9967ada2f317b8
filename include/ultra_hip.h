/*
 * ultra_hip.h — C ABI of libultra_hip.so, the MI355X (gfx950) implementation of uLTRA's per-read
 * alignment core (MAM finding + MAM chaining).
 *
 * The reference has no FFI for this path (it is pure Python calling the edlib wheel), so each
 * entry point below cites the reference call site(s) whose arithmetic it replaces.  Citations are
 * relative to /root/reference/.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every input and output buffer;
 *   - all functions return 0 (ULTRA_OK) or a negative ULTRA_E_* code; ultra_last_error(ctx) gives
 *     the text; nothing throws, nothing calls exit();
 *   - sequences are compared byte-exact and case-sensitive ('N' is an ordinary symbol), as the
 *     reference does (uLTRA:43, modules/create_augmented_gene.py:556);
 *   - a ctx is bound to one GPU and is not thread-safe; different ctxs are independent;
 *   - calls are synchronous on return unless the name says _stage/_run/_fetch.
 */
#ifndef ULTRA_HIP_H
#define ULTRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ULTRA_OK 0
#define ULTRA_E_ARG (-1)         /* bad argument (null pointer, negative length, offsets out of range) */
#define ULTRA_E_HIP (-2)         /* a HIP runtime call failed */
#define ULTRA_E_CAPACITY (-3)    /* a caller-sized output is too small; the needed size is reported */
#define ULTRA_E_UNSUPPORTED (-4) /* input outside the supported envelope (more than 36 distinct bytes in a batch) */
#define ULTRA_E_NODEVICE (-5)    /* no usable gfx950 device */
#define ULTRA_E_STATE (-6)       /* _run/_fetch called without a staged batch */

typedef struct ultra_ctx ultra_ctx;

/* Hard-coded constants of the reference's hot path, made explicit (defaults = reference values). */
typedef struct ultra_cfg {
    int32_t chain_switch_n;   /* > this many MAMs -> n log n solver   (classify_read_with_mams.py:561)  = 200 */
    int32_t overlap_small;    /* overlap threshold of the quadratic solver (classify_read_with_mams.py:575) = 20 */
    int32_t overlap_large;    /* overlap threshold of the n log n solver   (classify_read_with_mams.py:563) = 5 */
    int32_t loc_slots;        /* end positions kept per pair in the first pass before the exact re-run = 32 */
    int32_t banded_traceback; /* 1 = first-location traceback in a diagonal band sized by the known distance
                                 (same results, far less work); 0 = full-height columns everywhere        = 1 */
    int32_t nam_switch_n;     /* >= this many NAMs -> n log n NAM solver (align.py:448)                      = 90 */
    int32_t launch_lanes;     /* HIP streams the independent launches of one pass are dealt over, 1..8; 0 = one per
                                 hardware queue of the process (GPU_MAX_HW_QUEUES in the environment, else 4) = 0 */
    int32_t banded_starts;    /* 1 = start positions of locations with distance <= 59 by a banded reverse pass, lane per
                                 location (same results, 24 % fewer instructions, but 0.6 ms slower per 262144-read
                                 batch than the lane-group pass it replaces: kept as an option); 0 = full height   = 0 */
} ultra_cfg;

void ultra_cfg_default(ultra_cfg *cfg);

/* Library / device management (no reference counterpart: the reference has no device). */
int ultra_ctx_create(int device, const ultra_cfg *cfg, ultra_ctx **out);
void ultra_ctx_destroy(ultra_ctx *ctx);
const char *ultra_last_error(const ultra_ctx *ctx);
int ultra_device_count(void);
/* name/CU count/clock of the ctx's device, for bench reports */
int ultra_device_info(const ultra_ctx *ctx, char *name, int name_cap, int *n_cu, int *clock_khz);
/* Page-locked host memory for a caller's input arrays (hipHostMalloc / hipHostFree).  Optional: every entry point takes
 * ordinary memory too, but the copies of a batch to the device (NAM chains + reads, ~1.4 kB per read) only run at the
 * link's rate and beside the caller's thread when they start in page-locked memory.  No reference counterpart. */
int ultra_host_alloc(size_t bytes, void **out);
void ultra_host_free(void *p);
/* Debug canaries (no reference counterpart; SURVEY.md section 5 asks for them because ROCm has no compute-sanitizer): with
 * ULTRA_CANARIES=1 in the environment when the library is loaded, every device buffer is allocated at exactly the size
 * asked for between two 256-byte guard regions.  This call synchronises and reads every live buffer's guards back:
 * *n_damaged > 0 means a kernel stored out of bounds.  self_test != 0 also overruns a scratch buffer by one byte and
 * requires the check to notice.  ULTRA_E_STATE when the mode is off.  tests/test_gpu_canaries.py runs the hot path under it. */
int ultra_debug_canaries(int64_t *n_buffers, int64_t *n_damaged, int self_test);

/*
 * Batched semi-global ("HW", infix) Myers bit-vector search with locations and first-location path
 * statistics.  Replaces, for a whole batch of (query, target) pairs,
 *     edlib.align(query, target, task="path", mode="HW", k=k)
 * as called at modules/classify_read_with_mams.py:99 (from :312 segment-in-read, :353 exact small
 * segment, :377 read-in-segment) and consumed by edlib_alignment (:98-119) / cigar_to_accuracy (:71-94).
 *
 * arena[arena_len]       all sequences of the batch, raw bytes
 * pair p                 query = arena[q_off[p] .. +q_len[p]), target = arena[t_off[p] .. +t_len[p]),
 *                        k[p] = maximum distance (k < 0: unbounded)
 * out ed[p]              best distance, or -1 when it exceeds k[p]  (edlib 'editDistance')
 * out n_loc[p]           number of end positions attaining it
 * out loc_off[n_pairs+1] CSR offsets into loc_start/loc_end (filled by the library)
 * out loc_start/loc_end  inclusive 0-based target coordinates, ascending in end; start = smallest
 *                        start whose global alignment to the query costs ed  (edlib 'locations')
 * out first_ins[p]       number of inserted query bases ('I' columns) on the traceback path of the
 *                        FIRST location (tie rule: up, then left, then diagonal). With
 *                        L = end-start+1: accuracy = (L - ed + I) / (L + I)   (cigar_to_accuracy)
 * loc_cap                capacity of loc_start/loc_end in entries. If too small the call returns
 *                        ULTRA_E_CAPACITY and *loc_need holds the required number of entries.
 * ops/ops_off            optional (may be NULL): per-pair edit script of the first location, written
 *                        BACKWARDS (last column first) into ops[ops_off[p] .. ops_off[p+1]) as codes
 *                        0 '=', 1 'I', 2 'D', 3 'X'; out n_ops[p] = number of codes written.
 */
int ultra_myers_hw_batch(ultra_ctx *ctx,
                         const uint8_t *arena, size_t arena_len,
                         int32_t n_pairs,
                         const int64_t *q_off, const int32_t *q_len,
                         const int64_t *t_off, const int32_t *t_len,
                         const int32_t *k,
                         int32_t *ed, int32_t *n_loc, int64_t *loc_off,
                         int32_t *loc_start, int32_t *loc_end, int64_t loc_cap, int64_t *loc_need,
                         int32_t *first_ins,
                         uint8_t *ops, const int64_t *ops_off, int32_t *n_ops);

/* The same work split in three so that a benchmark can time the device part alone with inputs
 * resident in HBM: _stage copies and prepares (host -> device), _run launches every kernel of the
 * batch and waits, _fetch copies results back (same output contract as above, without the
 * optional edit scripts). */
int ultra_myers_stage(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                      const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                      const int32_t *k);
int ultra_myers_run(ultra_ctx *ctx);
int ultra_myers_fetch(ultra_ctx *ctx, int32_t *ed, int32_t *n_loc, int64_t *loc_off,
                      int32_t *loc_start, int32_t *loc_end, int64_t loc_cap, int64_t *loc_need,
                      int32_t *first_ins);
/* Forward-pass cell count (sum of q_len*t_len) and device milliseconds of the last _run, split by
 * stage: [0] forward (distance + end positions, incl. the location scan and the exact re-run), [1] reverse
 * (start positions), [2] traceback, [3] everything (first launch to last completion), [4] the forward
 * kernels alone (the concurrent myers_mega<FWD, bucket> launches, first start to last end).  Measured with
 * HIP events on the streams the kernels are launched on. */
int ultra_myers_last_timing(const ultra_ctx *ctx, double *cells, float ms[5]);
/* Cell accounting of the staged batch.  edlib.align(..., k) as the reference calls it (classify_read_with_mams.py:312:
 * k = 0.4 * min(len(read), len(seq))) returns -1 for a query more than k bases longer than its target after a few
 * columns - its band never reaches the last row.  The library answers those pairs (ed = -1, no locations) when the
 * batch is staged; no kernel sees them.  *cells_computed = sum of q_len * t_len over the pairs the forward pass runs
 * (what ultra_myers_last_timing reports and every GCUPS figure is taken over), *cells_skipped = the same sum over the
 * pairs answered without a launch, *pairs_skipped their number. */
int ultra_myers_cell_counts(const ultra_ctx *ctx, double *cells_computed, double *cells_skipped, int64_t *pairs_skipped);
/* Per register bucket of the last _run (a bucket = one forward kernel, myers_mega<FWD, bucket>, covering
 * the size classes up to W 64-row words per lane; bucket 3 = multi-lane classes for queries > 512):
 * G[i] = bucket id, W[i] = its widest class, forward-pass cells of its pairs and device milliseconds
 * of its forward launch.  Fills up to cap entries, *n_classes = number of non-empty buckets. */
int ultra_myers_class_timing(const ultra_ctx *ctx, int32_t cap, int32_t *G, int32_t *W, double *cells, float *fwd_ms,
                             int32_t *n_classes);

/*
 * Batched MAM colinear chaining.  Replaces
 *     colinear_solver.read_coverage_mam_score(mams, overlap_threshold=20)   (modules/colinear_solver.py:276-389)
 *     colinear_solver.n_logn_read_coverage_mams(mams, overlap_threshold=5)  (modules/colinear_solver.py:394-493,
 *         with modules/range_query_max_search_tree.py:19-157)
 * and the solver switch at modules/classify_read_with_mams.py:560-575.
 *
 * Instances are given in CSR form: instance i owns MAMs inst_off[i] .. inst_off[i+1]), already
 * sorted by y (stable), index j = position inside the instance.
 * solver: 0 = pick by cfg.chain_switch_n (reference behaviour), 1 = force quadratic, 2 = force n log n.
 * out chain_idx   caller-sized like the inputs; instance i's chain (ascending j) is written to
 *                 chain_idx[inst_off[i] .. inst_off[i] + chain_len[i])
 * out value[i]    optimal objective, FP64, bit-identical to the reference's arithmetic; -1 for an empty instance
 * overlap thresholds < 0 select the cfg defaults.
 */
int ultra_chain_mams_batch(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                           const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d,
                           const double *val, int32_t solver, int32_t overlap_small, int32_t overlap_large,
                           int32_t *chain_idx, int32_t *chain_len, double *value);

/*
 * ultra_chain_mams_batch for callers that cannot allocate device memory.  NO COUNTERPART IN THE REFERENCE: it exists for
 * tight serving loops with a fixed memory budget and harnesses that forbid hipMalloc (the call itself is synchronous -
 * host copies and a stream synchronisation - so it is not meant to be recorded into a HIP graph).  Same arguments, same results, bit for
 * bit, as ultra_chain_mams_batch above; but every byte of device memory it needs beyond the caller's host arrays
 * (staging copies of the MAM arrays, launch lists, result vectors, the n log n solver's tree arrays) lives in a fixed
 * 8 MiB array inside the library image - no hipMalloc / hipFree, no growable context buffers.
 *   - limit: ultra_chain_noalloc_capacity() bytes; a batch needs about 28 B per MAM + 32 B per instance (+ 28 B per
 *     MAM of an instance that takes the n log n solver); beyond that the call returns ULTRA_E_CAPACITY and touches
 *     nothing - split the batch, or use ultra_chain_mams_batch;
 *   - every GPU the library runs on holds its own copy of the array until the process ends;
 *   - all calls on one GPU share that GPU's array (whatever the context): the library serialises them (a lock per GPU held
 *     for the duration of the call); calls from different PROCESSES have their own copies of the array.
 */
int ultra_chain_mams_batch_noalloc(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                                   const int32_t *x, const int32_t *y, const int32_t *c, const int32_t *d,
                                   const double *val, int32_t solver, int32_t overlap_small, int32_t overlap_large,
                                   int32_t *chain_idx, int32_t *chain_len, double *value);
int64_t ultra_chain_noalloc_capacity(void);

/*
 * Batched NAM (seed) colinear chaining - the step before the MAM stage.  Replaces
 *     colinear_solver.read_coverage(mems, max_intron)    (modules/colinear_solver.py:169-272, n < 90)
 *     colinear_solver.n_logn_read_coverage(mems)         (modules/colinear_solver.py:85-167,  n >= 90)
 * and the solver switch at modules/align.py:448-453.  The objective is the number of read bases covered,
 * all arithmetic is integral.
 *
 * Instances in CSR form as above; the NAMs of an instance sorted the way align.py hands them over (by y,
 * help_functions / get_mems_from_input order), j = position inside the instance.  y = reference end of a NAM
 * (only read by the quadratic solver's intron test, y[j] - y[j'] < max_intron[i]), [c, d] = inclusive read
 * interval, 0 <= c <= d < 2^30.  max_intron[i]: limit of instance i (align.py:446); y / max_intron may be NULL
 * when solver == 2.
 * solver: 0 = pick by cfg.nam_switch_n (reference behaviour), 1 = force quadratic, 2 = force n log n.
 * The reference returns EVERY co-optimal chain (colinear_solver.py:159-165, :245-270), so the outputs are the
 * DP vectors themselves, n + 1 entries per instance at index inst_off[i] + i:
 * out C      C[0] = 0, C[j+1] = best coverage of a chain ending in NAM j
 * out trace  trace[0] = -1, trace[j+1] = 1 + predecessor NAM, 0 = chain starts here
 * out value  max(C) of the instance (the reference's second return value)
 * The co-optimal chains are the back-walks from every j with C[j] == value, in ascending j
 * (ultra_amd/colinear_solver.py:all_solutions does exactly that).
 */
int ultra_chain_nams_batch(ultra_ctx *ctx, int32_t n_inst, const int64_t *inst_off,
                           const int32_t *y, const int32_t *c, const int32_t *d,
                           const int64_t *max_intron, int32_t solver,
                           int64_t *C, int32_t *trace, int64_t *value);

/*
 * Batched base-level alignment - the step right behind the MAM stage.  Replaces
 *     help_functions.parasail_alignment(s1, s2, match_score=2, mismatch_penalty=-2, opening_penalty=3, gap_ext=1)
 *         -> parasail.sg_trace_scan_16 / _32                    (modules/help_functions.py:204-226)
 * as align.get_exact_alignment calls it per surviving location (modules/align.py:338-354): semi-global alignment of
 * query (the read) against target (the reconstructed exon sequence), end gaps free in both sequences at both ends,
 * affine gaps (a gap of length L costs gap_open + (L - 1) * gap_extend, penalties given as positive numbers),
 * substitution matrix matrix_create("ACGT", match, mismatch): A, C, G, T match case-insensitively, two different ones
 * score `mismatch`, a pair with any other character (N, IUPAC) scores 0 (the matrix's wildcard row / column).
 * out score[p]                 best score (int32 arithmetic: what the 16-bit scan kernel gives unless it saturates and
 *                              what the 32-bit rerun gives then)
 * out end_query / end_ref      0-based end cell (on the last row or the last column)
 * ops / ops_off / n_ops        optional (all three or none): edit script over the WHOLE of both sequences - leading and
 *                              trailing free gaps included, as help_functions.cigar_to_seq applies it (:217) - in
 *                              FORWARD order into ops[ops_off[p] ..), codes 0 '=', 1 'I' (query base against a gap),
 *                              2 'D', 3 'X'; every slot must hold q_len + t_len codes
 * parasail itself is absent from the build image: the score is mathematically defined; the choice among co-optimal
 * end cells and paths is this library's own (last row before last column, strictly better replaces; diagonal
 * before 'I' before 'D'; extending a gap wins a tie against opening one) and is PARITY UNPINNED against parasail.
 */
int ultra_sg_affine_batch(ultra_ctx *ctx, const uint8_t *arena, size_t arena_len, int32_t n_pairs,
                          const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                          int32_t match, int32_t mismatch, int32_t gap_open, int32_t gap_extend,
                          int32_t *score, int32_t *end_query, int32_t *end_ref,
                          uint8_t *ops, const int64_t *ops_off, int32_t *n_ops);

/*
 * The whole MAM stage of a batch of (read, location) instances in one call, results identical to
 *     classify_read_with_mams.main(solution, ..., read_seq, warning_log_file, min_acc)
 *         -> (non_covered_regions, value, mam_solution)          (modules/classify_read_with_mams.py:406-594)
 * for every instance: all edlib calls (ultra_myers_hw_batch above), add_segment_to_mam (:303-402) on their
 * results, phase A / first-last-exon reduction / phase B with its de-duplication (:440-523), the
 * outermost-partial filter (:530-547), the stable sort by y (:553-555), the solver switch (:560-575)
 * and the chainers (ultra_chain_mams_batch above) run on the GPU without intermediate host copies.
 * What stays on the host is the dictionary-driven candidate enumeration (:180-300, :441-523 loop
 * order), whose result is the input below.
 *
 * Candidates of instance i are cand[inst_cand_off[i] .. inst_cand_off[i+1]) in the reference's
 * emission order: full segments (kind 0, by start), full flanks (1, by start), then for every
 * partially hit segment the pair (2 = "_partial_segment_start" variant, 3 = "_partial_segment_end"
 * variant; the library applies the reference's if/elif), then partial flanks (4 start, 5 end).
 *   cand_e_start/e_stop     coordinates of the FULL segment or flank (x, y of its MAMs)
 *   cand_pair               pair index of the segment-in-read search (edlib call :312 / :353), -1 if none
 *                           (segment narrower than 5 bp); its k must be (2*min(m,n))/5, or 0 for widths 5..8
 *   cand_pair_contained     pair index of the read-in-segment search (:377), -1 if e_stop-e_start < 0.8*len(read)
 *   cand_seq_len            length of the sequence handed to add_segment_to_mam (partial variants: the cut piece)
 *   cand_seq_id             equal for two candidates of one instance iff their sequences are equal strings
 *                           (the reference's segm_already_tried sets, :479-523)
 *   inst_first_part_stop / inst_last_part_start   values of :194-221 (2^32 / 0 when no part was hit)
 */
typedef struct ultra_mam_input {
    const uint8_t *arena; size_t arena_len;
    int32_t n_pairs;
    const int64_t *q_off; const int32_t *q_len; const int64_t *t_off; const int32_t *t_len; const int32_t *k;
    int32_t n_cand;
    const uint8_t *cand_kind;
    const int32_t *cand_e_start, *cand_e_stop, *cand_pair, *cand_pair_contained, *cand_seq_len, *cand_seq_id;
    int32_t n_inst;
    const int64_t *inst_cand_off;       /* [n_inst + 1] */
    const int32_t *inst_read_len;
    const int64_t *inst_first_part_stop, *inst_last_part_start;
    const int32_t *cand_chr;            /* optional (may be NULL): chromosome id of every candidate's MAMs (ref_chr_id of
                                           :17); only needed when ultra_mam_fetch is asked for chain_chr */
} ultra_mam_input;

/*
 * Caller-sized outputs.  Per instance: value (-1 for an empty MAM instance, :577), n_mam (size of the
 * filtered MAM instance), chain_off (CSR, n_inst + 1).  Per chain element (the reference's mam_solution,
 * in chain order): x, y, c, d, val, min_segment_length, the candidate index it came from (the host
 * rebuilds mam_id = str(segm_id) + label from it) and j = its index in the instance's sorted MAM list.  mam_* (optional, all NULL or all set, with mam_off):
 * every instance's complete MAM list sorted by y, for parity tests on MAM sets.
 */
typedef struct ultra_mam_output {
    double *value; int32_t *n_mam; int64_t *chain_off;
    int64_t chain_cap;
    int32_t *chain_x, *chain_y, *chain_c, *chain_d; double *chain_val; int32_t *chain_msl, *chain_cand, *chain_j;
    int64_t mam_cap; int64_t *mam_off;
    int32_t *mam_x, *mam_y, *mam_c, *mam_d; double *mam_val; int32_t *mam_msl, *mam_cand;
    /* optional (may be NULL), per chain element like chain_x: the candidate kind (0..5, the label of its mam_id) and
     * the chromosome id of the MAM (ref_chr_id) - with chain_x / chain_y (= the candidate's e_start / e_stop) this is
     * everything a host needs to rebuild the reference's mam tuples without the candidate arrays */
    uint8_t *chain_kind; int32_t *chain_chr;
} ultra_mam_output;

int ultra_mam_batch(ultra_ctx *ctx, const ultra_mam_input *in, double min_acc, ultra_mam_output *out);
/* The same in three steps (host -> HBM once, then any number of device-only runs, then results back).
 * After _run, ultra_mam_sizes gives the sizes _fetch needs (sum of chain lengths, sum of n_mam). */
int ultra_mam_stage(ultra_ctx *ctx, const ultra_mam_input *in);
int ultra_mam_run(ultra_ctx *ctx, double min_acc);
/* ultra_mam_run in two halves, for a caller that pipelines consecutive batches (the reference overlaps reads by running
 * nr_cores workers, modules/pc.py:78-83; here the overlap is between a run's tail and the next batch's planner).
 * _run_begin returns when the whole run has been ENQUEUED - the host has waited for the forward pass only; the start pass,
 * the traceback, the MAM build and the chainers are still in flight - and _run_end waits for the device and closes the run.
 * Between the two nothing of this context may be touched; ultra_mam_run = _run_begin + _run_end. */
int ultra_mam_run_begin(ultra_ctx *ctx, double min_acc);
int ultra_mam_run_end(ultra_ctx *ctx);
int ultra_mam_sizes(const ultra_ctx *ctx, int64_t *n_chain_total, int64_t *n_mam_total);
int ultra_mam_fetch(ultra_ctx *ctx, ultra_mam_output *out);
/* Device milliseconds of the last _run beyond the Myers passes (ultra_myers_last_timing):
 * [0] MAM materialisation, [1] chaining, [2] chain compaction. */
int ultra_mam_last_timing(const ultra_ctx *ctx, float ms[3]);

/*
 * Packed index + candidate enumeration on the host.  No GPU work: these calls build,
 * natively and for a whole batch, the ultra_mam_input that ultra_mam_stage takes.  They replace the dictionary part
 * of classify_read_with_mams.main (modules/classify_read_with_mams.py:406-523): get_unique_exon_and_flank_locations
 * (:180-253), get_unique_segment_and_flank_choordinates (:256-300), the candidate loops and the edlib call shapes
 * (:303-377, :441-523), over flat copies of the five index dictionaries a worker unpickles (modules/align.py:53-110).
 *
 * Index: every key is the triple (chromosome id, start, stop) the reference packs with array('L', ...).tobytes().
 *   parts_to_segments        part_key[3*n_parts], CSR part_seg_off -> part_seg triples, in the stored order
 *   segment_to_gene          seg_gene_key[3*n_seg_gene], CSR seg_gene_off -> dense gene numbers 0..n_genes-1
 *   gene_to_small_segments   CSR gene_small_off[n_genes+1] -> gene_small triples
 *   ref_segment_sequences / ref_flank_sequences   keys + offsets (n+1) into pool
 * The index is copied; the caller's arrays may go away after the call.
 */
typedef struct ultra_index ultra_index;
typedef struct ultra_index_input {
    int64_t n_parts; const int64_t *part_key, *part_seg_off, *part_seg;
    int64_t n_seg_gene; const int64_t *seg_gene_key, *seg_gene_off; const int32_t *seg_gene;
    int64_t n_genes; const int64_t *gene_small_off, *gene_small;
    int64_t n_seg_seq; const int64_t *seg_seq_key, *seg_seq_off;
    int64_t n_flank_seq; const int64_t *flank_seq_key, *flank_seq_off;
    const uint8_t *pool; int64_t pool_len;
} ultra_index_input;
int ultra_index_create(const ultra_index_input *in, ultra_index **out);
void ultra_index_destroy(ultra_index *idx);

/*
 * One batch of (NAM chain, read) instances: instance i owns NAMs inst_nam_off[i] .. inst_nam_off[i+1]) in chain order
 * (nam_part = the (chromosome, start, stop) of the NAM's exon_part_id "chr^start^stop", nam_x / nam_y its reference
 * interval) and the read bytes reads[read_off[i] .. read_off[i+1]).
 * The plan owns the arrays behind ultra_plan_input(); hand that pointer to ultra_mam_stage / ultra_mam_batch.
 * ultra_plan_cands: per candidate the chromosome id of its MAMs and the chromosome field of its segment identifier
 * (mam_id = str(pack('LLL', ident_chr, e_start, e_stop)) + label, or "flank_<start>_<stop>" + label), and the
 * forward-pass cell count of the batch.  A candidate whose segment has no gene / no sequence in the index is the
 * reference's KeyError: ULTRA_E_ARG, text from ultra_plan_last_error() (thread-local).
 * Where the reference's order depends on the iteration order of a Python set - hit segments of one instance that
 * start at the same coordinate, classify_read_with_mams.py:180-253 - both planners reproduce that order (CPython's
 * tuple hash and set probing restated in csrc/pyset.h).  ultra_pyset_order exposes the restatement so that it can be
 * checked against an interpreter: `triples` = n pairwise distinct (chr, start, stop) in insertion order, order[k] =
 * index of the k-th tuple a Python set of them iterates.  Host only, no GPU.
 */
typedef struct ultra_plan ultra_plan;
typedef struct ultra_plan_request {
    int32_t n_inst;
    const int64_t *inst_nam_off;
    const int64_t *nam_part, *nam_x, *nam_y;
    const uint8_t *reads; const int64_t *read_off;
} ultra_plan_request;
int ultra_plan_create(const ultra_index *idx, const ultra_plan_request *rq, ultra_plan **out);
const ultra_mam_input *ultra_plan_input(const ultra_plan *plan);
int ultra_plan_cands(const ultra_plan *plan, const int32_t **chr, const int32_t **ident_chr, double *cells);
void ultra_plan_destroy(ultra_plan *plan);
const char *ultra_plan_last_error(void);
int ultra_pyset_order(const int64_t *triples, int32_t n, int32_t *order);

/*
 * The same enumeration ON THE DEVICE.  ultra_dindex_create flattens the host index once more and uploads it (segment
 * records in (start, stop, chromosome) order, one hash table for parts and flanks, the raw sequence pool: a few
 * hundred MB for a human annotation, resident in HBM).  ultra_mam_stage_nams then takes the place of
 * ultra_plan_create + ultra_mam_stage for a batch: the NAM chains and the reads go up, candidates and (query, target,
 * k) pairs are produced in device memory - over the resident pool, no per-batch copy of segment sequences - and the
 * context is left staged for ultra_mam_run / ultra_mam_fetch.  Results are identical to the host-planned batch.
 * Requirements beyond ultra_plan_create's: coordinates < 2^31, a part's segments on the part's chromosome, at most
 * 2048 raw hits / list entries, 64 genes and 16 flanks per instance (ULTRA_E_UNSUPPORTED otherwise: plan that batch
 * on the host).  ultra_mam_fetch's chain_cand then indexes candidate arrays that only exist on the device: ask for
 * chain_kind / chain_chr instead.
 */
typedef struct ultra_dindex ultra_dindex;
int ultra_dindex_create(int device, const ultra_index *idx, ultra_dindex **out);
void ultra_dindex_destroy(ultra_dindex *dix);
const char *ultra_dindex_last_error(void);
int ultra_mam_stage_nams(ultra_ctx *ctx, const ultra_dindex *dix, const ultra_plan_request *rq);
/* ultra_mam_stage_nams in its two halves, for callers that keep a batch's request in HBM (and for bench.py, whose timed
 * step starts with its inputs resident in HBM): _upload_nams copies the request's arrays into the context's device
 * buffers (synchronous; the host arrays may go away afterwards), _plan_resident runs the enumeration of
 * classify_read_with_mams.py:180-300, :441-523 over them, sorts the pairs and leaves the context staged.  The uploaded
 * request survives ultra_mam_run / _fetch and can be planned again; ultra_myers_stage / ultra_mam_stage replace it. */
int ultra_mam_upload_nams(ultra_ctx *ctx, const ultra_plan_request *rq);
int ultra_mam_plan_resident(ultra_ctx *ctx, const ultra_dindex *dix);
/* _plan_resident with its kernels ordered (on the device) behind the start pass of the run that `front`, another context
 * of the same GPU, has in flight (ultra_mam_run_begin, or ultra_mam_run on another thread: the call then first waits until
 * that run has enqueued its start pass).  The forward and start passes fill the chip, the rest of a run is latency-bound:
 * the planner of batch i + 1 rides under the tail of batch i.  With no run in flight on `front` (or front == NULL) it is
 * ultra_mam_plan_resident.  Results are those of ultra_mam_plan_resident. */
int ultra_mam_plan_resident_behind(ultra_ctx *ctx, const ultra_dindex *dix, ultra_ctx *front);
/* device copies of the staged batch's candidate arrays, for tests that compare the two planners: any pointer may be
 * NULL; capacities are the counts ultra_mam_counts reports */
int ultra_mam_counts(const ultra_ctx *ctx, int32_t *n_inst, int32_t *n_cand, int32_t *n_pairs);
int ultra_mam_fetch_plan(ultra_ctx *ctx, int64_t *inst_cand_off, int32_t *inst_read_len, int64_t *inst_fps, int64_t *inst_lps,
                         uint8_t *cand_kind, int32_t *cand_e_start, int32_t *cand_e_stop, int32_t *cand_pair, int32_t *cand_pair_contained,
                         int32_t *cand_seq_len, int32_t *cand_seq_id, int32_t *cand_chr,
                         int64_t *q_off, int32_t *q_len, int64_t *t_off, int32_t *t_len, int32_t *k);

/*
 * Batch pipeline of one GPU - the device-side replacement of the reference's worker fan-out (modules/pc.py:78-83:
 * nr_cores align_single processes on 1000-read batches).  `depth` slots (1..4), each with a context, a host thread
 * and pinned result buffers, share one resident device index; a submitted batch walks stage (H2D, device plan, sort)
 * -> run (kernels; one slot at a time) -> fetch (D2H into pinned memory) on its slot's thread, so the copies and the
 * planning of one batch ride beside the kernels of another.  A batch whose instances exceed the device planner's
 * envelope is planned on the host (ultra_plan_create) inside the slot; results do not depend on the route.
 *   ultra_pipe_submit   enqueues a batch; the request's arrays must stay valid until the batch has been waited for;
 *                       ULTRA_E_STATE when no slot is idle (a slot is busy from submit until release)
 *   ultra_pipe_wait     blocks until the OLDEST submitted batch is done and lends out its results: `view` points
 *                       into pinned memory of the slot (chain_* incl. chain_kind / chain_chr, value, n_mam, chain_off)
 *   ultra_pipe_release  gives the slot back; the view is invalid afterwards
 *   ultra_pipe_timing   of the batch currently lent out: host milliseconds [0] stage [1] waiting for the GPU
 *                       [2] run [3] fetch, forward cells, and the device milliseconds of ultra_myers_last_timing /
 *                       ultra_mam_last_timing
 * One caller thread per pipe.  To keep two batches in flight while a third is being consumed use depth 3.
 */
typedef struct ultra_pipe ultra_pipe;
int ultra_pipe_create(int device, const ultra_cfg *cfg, const ultra_index *idx, int32_t depth, ultra_pipe **out);
void ultra_pipe_destroy(ultra_pipe *pipe);
const char *ultra_pipe_last_error(const ultra_pipe *pipe);
int ultra_pipe_submit(ultra_pipe *pipe, const ultra_plan_request *rq, double min_acc);
int ultra_pipe_pending(const ultra_pipe *pipe);
int ultra_pipe_wait(ultra_pipe *pipe, int32_t *n_inst, int64_t *n_chain, ultra_mam_output *view);
int ultra_pipe_release(ultra_pipe *pipe);
int ultra_pipe_timing(const ultra_pipe *pipe, float ms[4], double *cells, float myers_ms[5], float mam_ms[3]);

/*
 * The per-read steps in FRONT of the MAM stage for a whole batch of reads, natively (worker threads on the host, one
 * NAM-chaining launch and one device-planned MAM stage on the GPU).  Replaces, per read of the batch,
 *     get_mems_from_input                  modules/align.py:357-388   hit lines -> NAMs per chromosome sorted by y
 *     annotate_guaranteed_optimal_bound    modules/align.py:112-188   instances cut at the intron limit + coverage bound
 *     the chaining loop of align_single    modules/align.py:440-475   instances by descending bound, break rule (:443),
 *                                                                     solver switch (:448), every co-optimal chain,
 *                                                                     sort by (-score, span), dropoff / max_loc (:472)
 *     help_functions.remove_read_polyA_ends (:41-73, the sequence; to_len = 1 as align.py:431 calls it)
 *     help_functions.reverse_complement     (:93-97) for locations on the reverse strand
 * and then classify_read_with_mams.main for every kept (read, location) (ultra_mam_stage_nams / _run / _fetch above).
 *
 * Input: the raw reads, and the hit lines of every read as byte ranges of one text block in the format namfinder
 * writes ("<chr^start^stop> <ref_pos> <read_pos> <len>" per line, whitespace separated, 1-based positions;
 * modules/seed_wrapper.py:139-179 streams these lines): fw_* = the read's forward record, rc_* = its "Reverse" record.
 * max_intron_chr is indexed by chromosome id (-1 = no entry: the reference's KeyError).  What the reference answers
 * with a Python exception (a malformed hit line, an unknown chromosome) is ULTRA_E_ARG with the read's index in the
 * text of ultra_front_last_error.
 *
 * Output (ultra_front_view: pointers into memory the ultra_front owns, valid until its next call or destruction):
 *   read_mod / read_mod_off      the poly-A-trimmed reads
 *   loc_off[n_reads + 1]         the locations kept per read, in the reference's order; per location its chromosome,
 *                                strand, NAM-chaining score, NAM chain (sol_*: CSR by sol_off; part = 3 values per NAM)
 *                                and the read as the MAM stage saw it (loc_reads: reverse-complemented where loc_rc)
 *   last_limit[n_reads]          the intron limit of the last instance the chaining loop worked on (:446; -1 = none),
 *                                which the reference's later filter (:505) compares every location of the read with
 *   mam                          the MAM stage's results, one instance per location (chain_kind / chain_chr included)
 *   ms[6]                        host milliseconds: parse + bounds, NAM chaining, selection, stage, run, fetch
 * ultra_front_prepare stops after the first phase (no GPU): instances in processing order (read_inst_off, their NAMs as
 * inst_nam_off / nam_y / nam_c / nam_d / inst_intron, and four values per instance - chromosome, strand, bound, read -
 * in inst_chr_rc_bound) for parity tests on machines without a GPU.
 */
typedef struct ultra_front ultra_front;
typedef struct ultra_front_input {
    int32_t n_reads;
    const uint8_t *reads; const int64_t *read_off;
    const uint8_t *seed_text;
    const int64_t *fw_begin, *fw_end, *rc_begin, *rc_end;
    int32_t n_chr; const int64_t *max_intron_chr;
    int64_t max_global_intron;     /* args.max_intron */
    double dropoff;                /* args.dropoff */
    int32_t max_loc;               /* args.max_loc */
    int32_t polyA_threshold;       /* args.reduce_read_ployA */
    double min_acc;                /* args.min_acc */
    int32_t n_threads;             /* 0 = min(16, hardware threads) */
} ultra_front_input;
typedef struct ultra_front_view {
    int64_t n_inst; const int64_t *read_inst_off;
    int64_t n_nam; const int64_t *inst_nam_off, *inst_intron; const int32_t *nam_y, *nam_c, *nam_d;
    const int64_t *inst_chr_rc_bound;
    const uint8_t *read_mod; const int64_t *read_mod_off;
    int64_t n_loc; const int64_t *loc_off, *last_limit;
    const int32_t *loc_chr; const uint8_t *loc_rc; const int64_t *loc_score;
    const int64_t *sol_off, *sol_part, *sol_x, *sol_y; const int32_t *sol_c, *sol_d, *sol_j; /* sol_j: the seed's index in its instance (Mem.j) */
    const uint8_t *loc_reads; const int64_t *loc_read_off;
    int64_t n_quadratic, n_nlogn;  /* instances that took the quadratic / the n log n NAM solver (align.py:450, :453) */
    ultra_mam_output mam;
    float ms[6];
} ultra_front_view;
int ultra_front_create(ultra_front **out);
void ultra_front_destroy(ultra_front *f);
const char *ultra_front_last_error(const ultra_front *f);
int ultra_front_prepare(ultra_front *f, const ultra_front_input *in, ultra_front_view *view);
/* ix: the host index, for batches beyond the device planner's envelope (may be NULL: such a batch is ULTRA_E_UNSUPPORTED);
 * with_mam_stage = 0 stops after the location selection (dix may then be NULL) */
int ultra_front_run(ultra_front *f, ultra_ctx *ctx, const ultra_dindex *dix, const ultra_index *ix, const ultra_front_input *in,
                    int32_t with_mam_stage, ultra_front_view *view);

#ifdef __cplusplus
}
#endif
#endif /* ULTRA_HIP_H */
