/*
 * mga_dev.h -- internal C interface between the host pipeline (C) and the HIP side.
 * Everything here takes DEVICE pointers unless a parameter is prefixed h_.  Not part of the public ABI.
 */
#ifndef MGA_DEV_H
#define MGA_DEV_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/minigraph_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime plumbing (dev_rt.hip) ---- */
int  mga_dev_init(void);                       /* 0 ok; <0 no usable GPU (message via mga_last_error) */
void mga_set_error(const char *fmt, ...);
void *mga_dmalloc(size_t bytes);               /* NULL on failure */
void mga_dfree(void *p);
int  mga_h2d(void *d, const void *h, size_t bytes);
int  mga_d2h(void *h, const void *d, size_t bytes);
int  mga_dmemset(void *d, int v, size_t bytes);
int  mga_dsync(void);
void *mga_hmalloc_pinned(size_t bytes);        /* pinned host memory for fast PCIe copies */
void mga_hfree_pinned(void *p);
double mga_wtime(void);
int  mga_host_pin(void *p, size_t bytes);      /* hipHostRegister with a registry: mga_free() and every realloc / free inside the library unregister first */
void mga_host_unpin(void *p);

/* grow-only device buffer */
typedef struct { void *p; size_t cap; } mga_dbuf_t;
int  mga_dbuf_reserve(mga_dbuf_t *b, size_t bytes); /* contents are NOT preserved on growth */
void mga_dbuf_free(mga_dbuf_t *b);

/* ---- stream context: one HIP stream + the per-stream scratch of the WFA kernels.  Every launcher below takes one;
 * two contexts let the mapping pipeline overlap the GPU work of one chunk with the host work of another. ---- */
typedef struct mga_sctx_s {
	void *stream;              /* hipStream_t */
	mga_dbuf_t wfa_ws[10];     /* per-tier WFA workspaces (register / HBM tiers) */
	mga_dbuf_t wfa_tbuf[8];    /* traceback regions of the windowed tiers, one buffer per pass of the ladder (k_wfa_w.hip) */
	mga_dbuf_t wfa_cnt;        /* work-queue counters, one 64-byte line per tier */
	mga_dbuf_t scan_tmp;       /* tile sums of mga_dev_scan_i32_to_i64 */
	mga_dbuf_t txt_cnt, txt_off, txt_vwb, txt_el; /* text kernel scratch (k_text.hip) */
	mga_dbuf_t wfa_list[2], wfa_key, wfa_ctl, wfa_fb; /* tier scheduler (k_wfa_sched.hip): work lists of a sweep / of the problems a sweep left open, sort keys, counters, problems for the chained fallback */
	int wfa_own[16];           /* a sweep's initial list lengths (source of an async upload) */
	void *tier_stream[16];      /* WFA tiers run concurrently on their own streams (long-tailed wide problems next to the small ones) */
	void *ev_ready, *ev_done[16];
	void *ev_sync;             /* event behind mga_ssync() */
	void *stage;               /* pinned staging for small device-to-host read-backs, delivered by mga_ssync() */
	mga_dbuf_t gc_split;       /* three-kernel form of graph chaining: counters, state list, jobs, walk vertices, state pool of the chunk in flight */
	mga_dbuf_t gc_arena[2];    /* per-wave scratch arenas of k_gchain: 1 MiB x resident waves, and the large tier for the reads that outgrow that */
	int wfa_uncapped;          /* set while the ladder runs the chained fallback's sub-problems: no 1e8-cell cap, unbounded last tier */
	mga_dbuf_t fb_prob, fb_res; /* sub-problems of the chained fallback and their results */
	mga_dbuf_t sk_planes; const void *sk_planes_src; /* the packed (bit-plane) form of a read buffer (mga_dev_pack2) and the buffer it was made from: sketch launches on that buffer read it */
	const int32_t *lc_order;   /* launch order of the NEXT mga_dev_lchain call on this context (device array of n read numbers; consumed by the call) */
} mga_sctx_t;
static inline void mga_dev_lchain_order(mga_sctx_t *sc, const int32_t *d_order) { sc->lc_order = d_order; }
void *mga_wfa_stream(mga_sctx_t *sc, int slot);       /* stream of WFA tier `slot` (the context's own stream unless MGA_WFA_CONCURRENT=1) */
int mga_wfa_tiers_serial(void);
int mga_wfa_fork(mga_sctx_t *sc);                     /* tier streams wait for everything queued on sc->stream so far */
int mga_wfa_join(mga_sctx_t *sc);                     /* sc->stream waits for every tier stream */
mga_sctx_t *mga_sctx_create(void);
void mga_sctx_destroy(mga_sctx_t *sc);
mga_sctx_t *mga_sctx_default(void);            /* lazily created, used by the stage-level API */
int  mga_dev_bind_thread(void);                /* hipSetDevice() for threads other than the one that called mga_dev_init */
int  mga_h2d_s(mga_sctx_t *sc, void *d, const void *h, size_t bytes);   /* async on sc->stream */
int  mga_d2h_s(mga_sctx_t *sc, void *h, const void *d, size_t bytes); /* <= 256 KB: any host memory, valid after mga_ssync(); larger: h must be pinned */
int  mga_dmemset_s(mga_sctx_t *sc, void *d, int v, size_t bytes);
int  mga_ssync(mga_sctx_t *sc);
void mga_sctx_abort(mga_sctx_t *sc);           /* error path: drain the stream, drop the staged read-backs (their destinations die with the caller's frame) */
typedef struct { void *p; size_t cap; } mga_hbuf_t;                     /* grow-only PINNED host buffer */
int  mga_hbuf_reserve(mga_hbuf_t *b, size_t bytes);
void mga_hbuf_free(mga_hbuf_t *b);

/* per-kernel HIP-event timing on the launch stream (bench.py reads it through mga_prof_get) */
enum { MGA_K_SKETCH = 0, MGA_K_SEED_COUNT, MGA_K_SEED_FILL, MGA_K_LCHAIN, MGA_K_WFA0 /* +tier: 0-2 single-wave register tiers (band 64,128,192), 3-6 multi-wave register tiers (256..2048), 7-8 HBM tiers */, MGA_K_SCAN = MGA_K_WFA0 + 9, MGA_K_TEXT, MGA_K_GCHAIN, MGA_K_PLAN,
	   MGA_K_WFAW0 /* +0..5: windowed tiers of 16 (4 problems per wave), 32 (2), 64, 128, 192, 256 diagonals (k_wfa_w.hip) */, MGA_K_WFATB = MGA_K_WFAW0 + 6 /* their traceback */, MGA_K_GCHAIN2 /* graph chaining, three-kernel form: part 2 (a wavefront per bridge) */, MGA_K_GCHAIN3 /* part 3; part 1 counts as MGA_K_GCHAIN */, MGA_K_GAF /* whole GAF lines (k_gaf.hip) */, MGA_K_N };
#define MGA_WFW_N 6         /* windowed tiers */
#define MGA_WFA_N_TIER 9    /* register + HBM tiers (k_wfa_r.hip 0-6, k_wfa.hip 7-8) */
#define MGA_WFA_N_SLOT 11   /* rungs of the ladder: W0-W5, R4-R6, H0-H1 (MGA_WFA_LADDER=old: R0-R6, H0-H1) */
#define MGA_WFA_MAX_TIER 16 /* array size of the per-rung resources */
void mga_prof_enable(int on);
void mga_prof_begin(void *stream, int kid);
void mga_prof_end(void *stream, int kid);
void mga_prof_collect(void);
void mga_prof_get(double *ms, int64_t *launches, int reset); /* arrays of MGA_K_N */


/* exclusive prefix sum of n int32 counts into n+1 int64 offsets, on device */
/* same over PIECES of sequences: item i = int32[4] {sequence, first base, end base, 0}; counts / offsets are per item (k must be odd) */
int mga_dev_sketch_items(mga_sctx_t *sc, int n_items, const int32_t *d_items, const char *d_seq, const int64_t *d_off, const uint32_t *d_rid, int w, int k,
						 int32_t *d_cnt, const int64_t *d_mz_off, mg128_t *d_mz);
int mga_dev_scan_i32_to_i64(mga_sctx_t *sc, const int32_t *d_cnt, int64_t n, int64_t *d_off);

/* ---- sketch (k_sketch.hip) ---- */
/* the read buffer as bit planes (3 x 64 bits per 64 bytes: low bit, high bit, is-ACGT): the sketch launches on d_seq that follow on this context read it instead of the bytes */
int mga_dev_pack2(mga_sctx_t *sc, const char *d_seq, int64_t n_bytes);
/* pass 1 (d_mz == NULL): d_cnt[i] = number of minimizers of sequence i.
 * pass 2: writes minimizers of sequence i at d_mz + d_mz_off[i]. */
int mga_dev_sketch(mga_sctx_t *sc, int n, const char *d_seq, const int64_t *d_off, const uint32_t *d_rid, int w, int k,
				   int32_t *d_cnt, const int64_t *d_mz_off, mg128_t *d_mz);

/* ---- device replica of the minimizer index (k_seed.hip) ---- */
typedef struct {
	uint64_t n_slots;        /* power of two = 1<<bits */
	int32_t bits, n_seg;
	mg128_t *d_tab;          /* n_slots x {key | MGA_IDX_LIST, value}; key == MGA_IDX_EMPTY: free slot.
	                            value: the single y, or off<<32 | n into d_pos when MGA_IDX_LIST is set */
	uint64_t *d_pos;         /* position lists, each ascending */
	int64_t n_pos;
	int32_t *d_seg_len;      /* n_seg segment lengths */
	char *d_gseq;            /* forward segment sequences back to back (text kernel: target bases of ds:Z) */
	int64_t *d_gseq_off;     /* n_seg + 1 offsets into d_gseq */
	/* graph replica for chaining on the device (k_gchain.hip): arcs in the host's order (gfa_arc_t[]), per-vertex arc index, reverse complements */
	void *d_arc;
	uint64_t *d_arc_idx;
	char *d_gseq_rc;         /* reverse complement of every segment at the same offsets as d_gseq */
	/* names and stable-sequence coordinates for the GAF lines written on the device (k_gaf.hip) */
	void *d_gaf_seg, *d_gaf_sseq;
	char *d_gaf_names;
} mga_didx_t;

/* collect_matches (map-algo.c:58-91), pass 1: probe every minimizer.  Flat per-minimizer outputs
 * d_occ[m] (occurrence count) and d_val[m] (slot value); per read d_na[i] (anchors), d_nmini[i]
 * (kept minimizers) and d_rep_len[i]. */
int mga_dev_seed_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, const int32_t *d_mz_cnt /* NULL: mz_off is exact */, int max_occ,
					   int32_t *d_occ, uint64_t *d_val, int32_t *d_na, int32_t *d_nmini, int32_t *d_rep_len);
/* collect_seed_hits (map-algo.c:152-192), pass 2: expand hits into anchors at d_a + d_a_off[i], write mini_pos at
 * d_mini + d_mini_off[i], then sort each read's anchors by x with the reference's exact permutation.
 * d_tmp: scratch of the same size as d_a. */
int mga_dev_seed_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, const int32_t *d_mz_cnt, int max_occ,
					  const int32_t *d_occ, const uint64_t *d_val, const int64_t *d_a_off, mg128_t *d_a,
					  const int64_t *d_mini_off, int32_t *d_mini, mg128_t *d_tmp);

/* long queries (MG_M_RMQ: contigs of megabases, the host chains): the same two passes with one thread per minimizer of the whole
 * batch and NO sort.  d_mz_off must be exact and contiguous.  count: probes, per-minimizer scans (d_off_a / d_off_m, n_mz + 1 int64 each),
 * per-read d_a_off / d_mini_off (n + 1) and d_rep_len; scratch d_tk, d_kf (n_mz int32), d_rep_key, d_rep_max (n_mz uint64).
 * fill: anchors in hit order at d_a + d_off_a[m], mini_pos at d_mini + d_off_m[m]. */
int mga_dev_seed_long_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, int64_t n_mz, int max_occ,
							int32_t *d_occ, uint64_t *d_val, int32_t *d_tk, int32_t *d_kf, int64_t *d_off_a, int64_t *d_off_m, uint64_t *d_rep_key, uint64_t *d_rep_max,
							int64_t *d_a_off, int64_t *d_mini_off, int32_t *d_rep_len);
int mga_dev_seed_long_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, int64_t n_mz, int max_occ,
						   const int32_t *d_occ, const uint64_t *d_val, const int64_t *d_off_a, const int64_t *d_off_m, mg128_t *d_a, int32_t *d_mini);

/* ---- linear chaining (k_lchain.hip) ---- */
/* per read i with anchors d_a[a_off[i]..a_off[i+1]) (x-sorted): chains u[] (score<<32|cnt) at d_u + a_off[i],
 * compacted anchors at d_b + a_off[i]; d_nu[i], d_nb[i] = their counts.  d_ws: mga_dev_lchain_ws_bytes(total) bytes. */
/* long-join rescue on the device (k_lchain.hip): parameters of the mg_lchain_rmq call at map-algo.c:407-417 */
typedef struct {
	int32_t enabled, max_dist, max_dist_inner, bw, max_skip, cap, min_cnt, min_sc;
	float chn_pen_gap, chn_pen_skip;
	int32_t rescue_size;
	float rescue_ratio;
	int32_t frag_len, frag_min_gap; /* -F (max_frag_len > 0, no max_gap_ref): the DP's reference gap is max(frag_len - qlen, frag_min_gap) per read (map-algo.c:383-386) */
} mga_rescue_par_t;
/* resc/d_q_off/d_flag may be NULL (no rescue).  d_flag[i]: 0 first-pass chains, 1 rescued on the device, 2 rescue due but left to the host */
int mga_dev_lchain(mga_sctx_t *sc, int n, const mg128_t *d_a, const int64_t *d_a_off, const mga_lchain_par_t *par, const mga_rescue_par_t *resc,
				   const int64_t *d_q_off, uint64_t *d_u, mg128_t *d_b, int32_t *d_nu, int32_t *d_nb, int32_t *d_flag, void *d_ws, size_t ws_bytes, int64_t total_anchors);
size_t mga_dev_lchain_ws_bytes(int64_t total_anchors);
int mga_dev_sort128x(mga_sctx_t *sc, int n, mg128_t *d_a, const int64_t *d_a_off, mg128_t *d_tmp, int32_t *d_stk); /* (stage test of the kernels' klib sort; d_tmp: as many elements as d_a, d_stk: 3 int32 per element + 8 per array) */

/* ---- WFA (k_wfa.hip) ---- */
typedef struct { int64_t t_off, q_off; int32_t tl, ql; } mga_wfa_prob_t;
typedef struct { int32_t score, n_cigar; int64_t cig_off; int32_t status, pad; int64_t n_iter; } mga_wfa_res_t;
enum { MGA_WFA_OK = 0, MGA_WFA_PENDING = 1, MGA_WFA_RETRY_TIER = 2, MGA_WFA_POOL_FULL = 3, MGA_WFA_MAX_ITER = 4,
	   MGA_WFA_TB = 5 /* forward pass of a windowed tier done: score set, cig_off = address of the traceback region, pad = last state | phase << 4 | window << 8 */ };
/* where a WFA kernel appends the problems that outgrew its tier (device pointers); err counts the other failures; fb_list/fb_cnt collect
 * the problems that hit the reference's 1e8-cell cap (miniwfa.c:827): they are re-done by the chained fallback (k_wfa_sched.hip) */
typedef struct { int32_t *list; int *cnt; int *err; int32_t *fb_list; int *fb_cnt; int32_t cap; } mga_wfa_retry_t; /* (cap: room of `list`; *cnt runs past it when it overflows) */

/* All tier launchers: the work list is d_list[first .. min(*d_n, cap)) -- the count is read ON THE DEVICE when the kernel starts, so that a rung can take what
 * the rungs below it appended in the same sweep without a host round trip; cap sizes the launch; slot picks the 64-byte work-queue counter of the launch
 * (zeroed by the caller), stream the HIP stream (NULL: the context's).
 * solves problems of the list in HBM-resident capacity tier 0..1 (4096 / 32768 diagonals); cigars are
 * appended to d_pool (capacity pool_cap ops, *d_pool_used bumped atomically); sequences must be padded by >= 8 readable bytes */
int mga_dev_wfa(mga_sctx_t *sc, const int *d_n, int cap, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
				mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt);
/* register-resident tiers (k_wfa_r.hip): tier 0-2 one wave per problem (64, 128, 192 diagonals), 3-6 two to sixteen waves (256, 512, 1024, 2048) */
int mga_dev_wfa_reg(mga_sctx_t *sc, const int *d_n, int cap, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
					mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt);
/* windowed tiers (k_wfa_w.hip): forward pass of tier wt (0..5) over d_list; traceback bytes go to d_tb + item * mga_dev_wfa_win_tb_stride(wt); `slot` picks the
 * work-queue counter.  mga_dev_wfa_traceback() then turns every MGA_WFA_TB result of the SAME list into score + CIGAR in the pool (MGA_WFA_OK); it runs right
 * behind the rung's forward pass, so that ONE region buffer (the largest rung's) serves the whole sweep. */
int64_t mga_dev_wfa_win_tb_stride(int wt);
int mga_dev_wfa_win(mga_sctx_t *sc, const int *d_n, int cap, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
					mga_wfa_res_t *d_res, char *d_tb, int wt, int slot, mga_wfa_retry_t rt,
					uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int *d_err /* the packed rungs (128 / 192 / 256 diagonals) walk their own alignments: results final, CIGARs in the pool */);
#define MGA_WFA_FUSE_SLACK (3LL * 8192 * 1024) /* CIGAR pool: the abandoned tails of the blocks k_wfa_fwp's wavefronts take (three rungs, <= 8192 wavefronts, 1024 operators) */
int mga_dev_wfa_traceback(mga_sctx_t *sc, void *stream /* NULL: the context's */, const int *d_n, int cap, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq, mga_wfa_res_t *d_res,
						  uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int *d_err, int wt /* the rung's window tier: only results of THAT window are walked */);
int32_t mga_wfw_window(int32_t W, int32_t tl, int32_t ql, int32_t *lo);
/* tier 0..8: register tiers (one wave, then 2-16 waves per problem), then the HBM-resident tiers */
int mga_wfa_first_tier(int32_t tl, int32_t ql); /* cheapest tier likely to fit, from the sequence lengths */
int mga_dev_wfa_tier(mga_sctx_t *sc, const int *d_n, int cap, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
					 mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt);
/* the whole ladder (k_wfa_sched.hip): every problem in its first tier, the ones that outgrow it one tier up, until none is left;
 * d_res[i] / d_pool hold the results; *cells (optional) = total wavefront cells */
int mga_dev_wfa_solve(mga_sctx_t *sc, int n, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
					  mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int64_t *cells,
					  void (*bulk_done)(void*), void *bulk_arg); /* bulk_done (optional): called once the tiers of <= 512 diagonals finished their first pass */

/* the whole minimizer index of the graph on the device (k_index.hip): sketch of the n_seg segments in d_seq/d_off, sort, table, position lists;
 * fills ix->d_tab/d_pos/n_slots/bits/n_pos; the host gets the counters and the occurrence histogram (malloc'ed) */
int mga_dev_index_build(mga_sctx_t *sc, int n_seg, const char *d_seq, const int64_t *d_off, int w, int k, mga_didx_t *ix,
						int64_t *h_n_keys, int64_t *h_n_mz, int64_t **h_occ_hist, int64_t *h_max_occ);

/* ---- alignment text on the device (k_text.hip): stitched CIGAR statistics + cg:Z / ds:Z strings of a chain ---- */
typedef struct { int32_t op, val; } mga_cigitem_t; /* op >= 0: ready operator (op, len = val); op == -1: WFA problem #val of this read's pool */
typedef struct {
	int64_t item_beg, item_end;   /* plan items of the chain */
	int64_t prob_base;            /* global index of problem 0 of the pool the items refer to */
	int64_t q_base;               /* offset of the read in the resident read buffer */
	int64_t vert_beg;             /* oriented vertices of the walk */
	int32_t vert_cnt;
	int32_t qs, qe, ss, ee, ps, pe;
	int32_t rev_sign;             /* the line is printed on the reverse strand (format.c:123,183) */
} mga_txt_chain_t;
typedef struct { int64_t txt_off; int32_t cg_len, ds_len, n_cigar, mlen, blen, aplen, status, pad; } mga_txt_res_t; /* text = pool + txt_off: cg then ds */
int mga_dev_text_tables(const unsigned char *comp, const unsigned char *nt4); /* once: IUPAC complement + nt4 code tables to constant memory */
/* n_el_max: upper bound on the operators of all chains before merging (plan items + operators of all WFA CIGARs) */
int mga_dev_text(mga_sctx_t *sc, int n_chain, const mga_txt_chain_t *d_chain, const mga_cigitem_t *d_item, int64_t n_vert, const uint32_t *d_vert,
				 const mga_didx_t *ix, const char *d_reads, int64_t n_el_max, const int32_t *d_ncig, const int64_t *d_cigoff, const uint32_t *d_ord,
				 mga_txt_res_t *d_res, char *d_pool, int64_t pool_cap, unsigned long long *d_pool_used);

/* ---- whole GAF lines on the device (k_gaf.hip): mg_write_gaf, format.c:121-250 ---- */
typedef struct {
	int32_t read;             /* read of the chunk: its name (d_qnames + d_qname_off[read]) */
	int32_t chain;            /* the text kernel's chain: walk, cg:Z / ds:Z, mlen / blen; -1: the line of an unmapped read (MG_M_SHOW_UNMAP) */
	int32_t qlen, qs, qe, plen, ps, pe;
	int32_t mapq, n_anchor, score, subsc;
	int32_t primary;          /* tp:A:P (id == parent) or tp:A:S */
	float div;                /* dv:f:, printed when 0 <= div <= 1 */
} mga_gaf_line_t;
int mga_dev_gaf_names_upload(const gfa_t *g, mga_didx_t *ix);
int mga_dev_gaf_div(mga_sctx_t *sc, int n, const float *d_div, char *d_out); /* (stage test) dv:f: text of n values, 8 bytes each */
/* d_out == NULL: d_len[i] = bytes of line i; else: line i at d_out + d_off[i].  Lines are the caller's: printed chains of the chunk in read order */
int mga_dev_gaf(mga_sctx_t *sc, const mga_didx_t *ix, int n_lines, const mga_gaf_line_t *d_line, const char *d_qnames, const int64_t *d_qname_off, uint64_t flag,
				const mga_txt_chain_t *d_chain, const uint32_t *d_vert, const mga_txt_res_t *d_tres, const char *d_tpool, int32_t *d_len, const int64_t *d_off, char *d_out);

/* ---- graph chaining on the device (k_gchain.hip, gc_core.h): from the chains of k_lchain to filtered graph chains ---- */
typedef struct { int32_t n_gc, n_lc, n_a, status; int64_t gc_off, lc_off, a_off; } mga_gc_hdr_t; /* per read: records at gc_pool + gc_off, lc_pool + lc_off, anchors at a_pool + a_off */
#define MGA_GC_E_POOL 3   /* status: the chunk's record pools were too small: grow and re-run the listed reads */
#define MGA_GC_HOST   4   /* status: not chained here (k_lchain left the read's long-join rescue to the host tree): the host chains it */
int mga_dev_graph_upload(mga_sctx_t *sc, const gfa_t *g, const unsigned char *comp, mga_didx_t *ix); /* arcs, arc index, reverse complements -> HBM */
size_t mga_gc_rec_bytes(void);
size_t mga_dev_gchain_arena_bytes(int tier);
int mga_dev_gchain_waves(int tier);
int mga_dev_gchain(mga_sctx_t *sc, const mga_didx_t *ix, const mg_mapopt_t *opt, int k, float pen_gap, int n, const int32_t *d_list, int tier,
				   const int64_t *d_a_off, const int32_t *d_nu, const int32_t *d_nb, const uint64_t *d_u, const mg128_t *d_b,
				   const int64_t *d_mini_off, const int32_t *d_mini, const int64_t *d_q_off, const char *d_seq, const uint32_t *d_hash, const int32_t *d_rflag,
				   mga_gc_hdr_t *d_hdr, void *d_gc_pool, int64_t gc_cap, mg_llchain_t *d_lc_pool, int64_t lc_cap, mg128_t *d_a_pool, int64_t a_cap,
				   unsigned long long *d_ctl, int32_t *d_retry);
/* flat records of one read -> a malloc'ed mg_gchains_t (div and MAPQ computed here, on the host's libm) */
mg_gchains_t *mga_gchains_from_flat(int32_t n_gc, const void *gc_recs, int32_t n_lc, const mg_llchain_t *lc, int32_t n_a, const mg128_t *a,
									int32_t rep_len, int32_t qlen, int32_t n_mz, int32_t min_gc_score);
/* the same routine on a host thread (gc_core.h compiled for the host): -x asm, CPU parity tests.  a[] is modified. */
mg_gchains_t *mga_gchain_host_read(const mg_idx_t *gi, const int32_t *seg_len, const mg_mapopt_t *opt, float pen_gap, int32_t qlen, uint32_t hash,
								   int32_t n_u, const uint64_t *u, mg128_t *a, int32_t n_a, int32_t n_mini, const int32_t *mini_pos, const char *qseq,
								   int32_t rep_len, int32_t n_mz, int32_t *n_gwfa, int32_t *n_shortk);

/* ---- the gap list on the device (k_plan.hip): what align.c:mga_plan_cigar makes on the host, for chains that never left the device ---- */
typedef struct { /* one graph chain as k_gchain leaves it in the record pool (gc_core.h:gc_rec_t; k_gchain.hip asserts the layouts agree) */
	int32_t off, cnt, n_anchor, score;
	int32_t qs, qe, plen, ps, pe, blen, mlen;
	int32_t n_mini, q_span;
	int32_t id, parent, subsc, n_sub, flt;
	uint32_t hash;
} mga_gc_rec_t;
typedef struct { int64_t lc0; int32_t n_lc, x0, x1, pad; } mga_plan_src_t; /* target of a problem: base x0+1 of vertex lc_pool[lc0] .. base x1 of vertex lc_pool[lc0 + n_lc] */
/* pass 1: d_cnt = int32[5][n] scratch; d_off = int64[5][n+1] exclusive scans of printed chains, plan items, problems, walk vertices, target bytes;
 * d_tot = 8 x uint64: the five totals, [5] query bases of all problems, [6] reads whose target bytes do not fit 31 bits */
int mga_dev_plan_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, int print_2nd, const mga_gc_hdr_t *d_hdr, const void *d_gc_pool, const mg_llchain_t *d_lc_pool,
					   const mg128_t *d_a_pool, int32_t *d_cnt, int64_t *d_off, unsigned long long *d_tot);
/* pass 2: items, problems (+ their targets in d_tseq), printed chains (rev_sign from d_rev, one flag per printed chain) and walk vertices at the scanned offsets */
int mga_dev_plan_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, int print_2nd, const mga_gc_hdr_t *d_hdr, const void *d_gc_pool, const mg_llchain_t *d_lc_pool,
					  const mg128_t *d_a_pool, const int64_t *d_q_off, const int64_t *d_off, const int32_t *d_rev, int64_t n_prob,
					  mga_cigitem_t *d_item, mga_wfa_prob_t *d_prob, mga_plan_src_t *d_src, mga_txt_chain_t *d_chain, uint32_t *d_vert, char *d_tseq);

/* forward pass of the RMQ chainer (mg_lchain_rmq, lchain.c:252-357) over runs of a CHUNK's x-sorted anchors (k_rmq.hip; each read's slice sorted on its own): d_runs = array of
 * { int64 beg, end, base } (positions in d_a; base = first anchor of the run's read), taken in the order d_order lists them; d_status[r] = 0, or 1 / 2 when the host has to redo
 * the run (tied priorities / inner window beyond the kernel's sort); p[] comes out relative to `base`.  Scratch: d_t (zeroed by the caller), d_ys (int32), d_pri (doubles) */
typedef struct { int64_t beg, end, base; } mga_rq_run_t;
int mga_dev_rmq_fwd(mga_sctx_t *sc, int64_t n_total, const mg128_t *d_a, int n_order, const void *d_runs, const int32_t *d_order, int max_dist, int max_dist_inner, int bw,
					int max_skip, int cap, float pen_gap, float pen_skip, int32_t *d_f, int64_t *d_p, int32_t *d_v, int32_t *d_t, double *d_pri, int32_t *d_ys,
					int32_t *d_status, int *d_counter);

/* targets of HOST-listed gaps spliced on the device: problem j's target = the walk d_vert[src[j].lc0 .. + n_lc] from base x0 + 1 to base x1, written at d_tseq + prob[j].t_off */
int mga_dev_plan_target_verts(mga_sctx_t *sc, const mga_didx_t *ix, int64_t n_prob, const mga_wfa_prob_t *d_prob, const mga_plan_src_t *d_src, const uint32_t *d_vert, char *d_tseq);

/* CIGARs of all problems copied into problem order: d_ncig[i] operators at d_ord + d_off[i] (d_off has n+1 entries); *h_total = d_off[n] */
int mga_dev_wfa_gather(mga_sctx_t *sc, int n, const mga_wfa_res_t *d_res, const uint32_t *d_pool, int32_t *d_ncig, int64_t *d_off, uint32_t *d_ord,
					   int64_t ord_cap, int64_t *h_total);

#ifdef __cplusplus
}
#endif
#endif
