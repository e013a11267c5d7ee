/*
 * mapper.c -- the batched mapping pipeline: mg_map_batch() and the reference-compatible per-read
 * wrappers mg_map() / mg_map_frag().
 *
 * This is the MI355X replacement for kt_for(n_threads, worker_for, ...) -> mg_map_frag() at the
 * reference's gmap.c:99 / map-algo.c:340-495.  One call maps a whole mini-batch:
 *
 *   GPU   reads -> HBM, k_sketch (count + write), k_seed_count / k_seed_fill, k_lchain
 *   host  per read, on n_threads threads (map-algo.c:407-474): long-join rescue (RMQ chainer), lchain
 *         records + clean-up, graph chaining with shortest-k / GWFA bridging, parent/filter/MAPQ, and
 *         the gap list for base alignment
 *   GPU   k_wfa over every gap of the batch (three capacity tiers)
 *   host  CIGAR stitching, ds:Z
 *
 * The two host halves are exported separately (mga_batch_chain / mga_batch_finish) so that the
 * host logic is testable on its own; mg_map_batch() is the only caller that matters in production.
 */
#include <stdio.h>
#include <math.h>
#include <assert.h>
#include "hchain.h"
#include "align.h"
#include "mapper.h"

/* MGA_DEBUG_PIPE: per-stage host CPU time (thread clocks), summed over worker threads */
#include <time.h>
enum { C_LCCOPY, C_LCRESCUE, C_LCPREP, C_GCDP, C_GCGEN, C_GCPOST, C_PLAN, C_APPLY, C_DS, C_GAF, C_EXPORT, C_PIPE, C_SYNC, C_COMMIT, C_READER, C_FASCAN, C_WRITER, C_N };
static const char *g_cname[C_N] = { "lchain_copy", "lchain_rescue", "lchain_gen", "gchain_dp", "gchain_gen", "gchain_post", "plan_cigar", "apply_cigar", "gen_ds", "gaf", "export", "pipe_thread(all)", "of_which_ssync", "commit", "reader_thread", "fasta_scan_fill", "writer+collector" };
static volatile int64_t g_cpu_ns[C_N];
static int g_cpu_on = 0;
static inline int64_t cpu_now(void) { struct timespec ts; if (!g_cpu_on) return 0; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec; }
void mga_cpu_note(int which, int64_t ns) { if (g_cpu_on) __sync_fetch_and_add(&g_cpu_ns[which], ns); } /* (other files: 11 pipe, 12 ssync, 13 commit, 14 reader, 15 fasta scan/fill, 16 writer) */
int64_t mga_cpu_now(void) { return cpu_now(); }
#define CPU_ADD(which, t0) do { if (g_cpu_on) { int64_t t1_ = cpu_now(); __sync_fetch_and_add(&g_cpu_ns[which], t1_ - (t0)); (t0) = t1_; } } while (0)

/* Grow-only scratch that is recycled across chunks instead of being freed: the per-thread planning pools and the GAF
 * pieces are tens of MB per chunk, and handing them back to malloc means mmap/munmap and a page fault per 4 KB on
 * every chunk ([measured] ~0.2 s of host CPU per 100k reads). */
#include <pthread.h>
static pthread_mutex_t g_cache_mtx = PTHREAD_MUTEX_INITIALIZER;
#define TP_CACHE_MAX 512
static mga_tpool_t g_tp_cache[TP_CACHE_MAX];
static int g_n_tp_cache = 0;
static void tpool_get(mga_tpool_t *tp)
{
	pthread_mutex_lock(&g_cache_mtx);
	if (g_n_tp_cache > 0) *tp = g_tp_cache[--g_n_tp_cache]; else memset(tp, 0, sizeof *tp);
	pthread_mutex_unlock(&g_cache_mtx);
	tp->n_t = tp->n_prob = tp->n_item = tp->n_chain = tp->n_vert = 0, tp->wfa_t_bases = tp->wfa_q_bases = 0, tp->want_src = 0;
}
static void tpool_put(mga_tpool_t *tp)
{
	pthread_mutex_lock(&g_cache_mtx);
	if (g_n_tp_cache < TP_CACHE_MAX) { g_tp_cache[g_n_tp_cache++] = *tp; pthread_mutex_unlock(&g_cache_mtx); return; }
	pthread_mutex_unlock(&g_cache_mtx);
	free(tp->tseq); free(tp->prob); free(tp->item); free(tp->chain); free(tp->vert); free(tp->src);
}
#define STR_CACHE_MAX 512
static struct { char *s; size_t m; } g_str_cache[STR_CACHE_MAX];
static int g_n_str_cache = 0;
static char *strbuf_get(size_t want, size_t *cap) /* a cached buffer (possibly grown) of at least `want` bytes */
{
	char *p = 0;
	size_t m = 0;
	int i, best = -1;
	pthread_mutex_lock(&g_cache_mtx);
	for (i = g_n_str_cache - 1; i >= 0; --i) { if (g_str_cache[i].m >= want) { best = i; break; } if (best < 0 || g_str_cache[i].m > g_str_cache[best].m) best = i; }
	if (best >= 0) { p = g_str_cache[best].s, m = g_str_cache[best].m; g_str_cache[best] = g_str_cache[--g_n_str_cache]; }
	pthread_mutex_unlock(&g_cache_mtx);
	if (m < want) { p = (char*)realloc(p, want); m = want; }
	*cap = m;
	return p;
}
static void strbuf_put(char *p, size_t m)
{
	if (p == 0) return;
	pthread_mutex_lock(&g_cache_mtx);
	if (g_n_str_cache < STR_CACHE_MAX) { g_str_cache[g_n_str_cache].s = p, g_str_cache[g_n_str_cache++].m = m; p = 0; }
	pthread_mutex_unlock(&g_cache_mtx);
	free(p);
}


/* ------------------------------------------------------------------------------------------------ */

typedef struct {
	int32_t tid;          /* pool that holds this read's plan */
	int32_t n_gc;
	int64_t *item_off;    /* n_gc + 1 offsets into the pool's item[] */
	int64_t *chain_id;    /* text mode: per chain, its index in the pool's chain[] (-1: not printed) */
} read_plan_t;

struct mga_batch_s {
	const mg_idx_t *gi;
	mg_mapopt_t opt;
	int n, n_threads;
	const int *qlens;
	const char **seqs, **qnames;
	const int64_t *q_off;   /* offset of read i in the device read buffer */
	float pen_gap, pen_skip;
	mg_gchains_t **gcs;
	read_plan_t *plan;
	mga_tpool_t *tp;
	int64_t *tp_prob_base, *tp_t_base, *tp_item_base, *tp_chain_base, *tp_vert_base;
	int want_text;          /* the caller only wants GAF bytes: cg:Z / ds:Z come from the device (k_text.hip) */
	int want_src;           /* ... and the targets of the gaps are spliced on the device from descriptors (align.h: mga_tpool_t::want_src) */
	const mga_txt_res_t *txt_res; /* text-mode results, global chain order */
	const char *txt_pool;
	const char *gaf_dev; int64_t gaf_dev_len; /* the chunk's GAF lines as the device wrote them (k_gaf.hip), read order, back to back; NULL: the host formats */
	/* stage-1 inputs, valid during mga_batch_chain() only */
	const int32_t *n_mz, *rep_len, *mini_pos, *nu, *nb;
	const int64_t *mini_off, *a_off;
	const uint64_t *u;
	const mg128_t *a;
	int a_is_raw;
	struct rq_read_s *rq;   /* a_is_raw: per-read state of the phased RMQ chaining (see rq_* below) */
	/* round 5: the forward passes of the RMQ chainer on the device (k_rmq.hip), one read at a time on the chunk's stream; rq_harr = pinned arrays p | f | v | t of ALL the
	 * chunk's anchors (the device writes f, p, v straight into them), rq_tot = their length in anchors */
	int (*rq_dev_fwd)(void *ctx, int n_reads, const int64_t *r_abs0, const int64_t *r_n, const mg128_t *const *r_a, int32_t *const *r_f, int64_t *const *r_p, int32_t *const *r_v,
					  int n_runs, const int64_t *runs /* beg, end, base triples, chunk-level */, int n_order, const int32_t *order, int bw, int32_t *status);
	void *rq_dev_ctx; char *rq_harr; int64_t rq_tot;
	pthread_mutex_t rq_dev_mtx;
	int rq_dev_err; char rq_dev_errmsg[256]; /* a device call of the hook failed (not: a run handed back): the chunk fails with this message -- no silent host path */
	int32_t *seg_len;       /* segment lengths as a flat array (the graph view of gc_core.h on the host) */
	/* graph chains made on the device (k_gchain.hip): per-read headers + record pools; status != 0 marks the reads the host still chains */
	const mga_gc_hdr_t *gc_hdr; const char *gc_pool; const mg_llchain_t *gc_lc; const mg128_t *gc_a; size_t gc_rec;
	const int32_t *rescue_flag; /* per read: what the chaining kernel did about the long-join rescue (NULL: decide here) */
	/* gap list made on the device (k_plan.hip) for the device-chained reads: first printed chain of read i in the device's chain table; the host
	 * fills one strand flag per printed chain (dp_rev); the pools above then only hold the plans of the reads chained HERE, appended after the device's */
	const int64_t *dp_chain_off; int32_t *dp_rev; int64_t dp_n_chain;
	/* stage-2 inputs */
	mga_cigsrc_t src;
	int err;
};

mga_batch_t *mga_batch_init(const mg_idx_t *gi, const mg_mapopt_t *opt, int n, const int *qlens, const char **seqs, const char **qnames,
							const int64_t *q_off, int n_threads)
{
	mga_batch_t *b = MGA_CALLOC(mga_batch_t, 1);
	float tmp;
	b->gi = gi, b->opt = *opt, b->n = n, b->qlens = qlens, b->seqs = seqs, b->qnames = qnames, b->q_off = q_off;
	b->n_threads = n_threads > 0 ? n_threads : 1;
	tmp = expf(-opt->div * gi->k); /* map-algo.c:388-390 */
	b->pen_gap = opt->chn_pen_gap * tmp, b->pen_skip = opt->chn_pen_skip * tmp;
	b->gcs = MGA_CALLOC(mg_gchains_t*, n > 0 ? n : 1);
	b->plan = MGA_CALLOC(read_plan_t, n > 0 ? n : 1);
	b->tp = MGA_CALLOC(mga_tpool_t, b->n_threads);
	{ int t_; for (t_ = 0; t_ < b->n_threads; ++t_) tpool_get(&b->tp[t_]); }
	b->tp_prob_base = MGA_CALLOC(int64_t, b->n_threads + 1);
	b->tp_t_base = MGA_CALLOC(int64_t, b->n_threads + 1);
	b->tp_item_base = MGA_CALLOC(int64_t, b->n_threads + 1);
	b->tp_chain_base = MGA_CALLOC(int64_t, b->n_threads + 1);
	b->tp_vert_base = MGA_CALLOC(int64_t, b->n_threads + 1);
	return b;
}

void mga_batch_wfa_export_src(const mga_batch_t *b, mga_wfa_prob_t *prob, char *tseq, mga_plan_src_t *src, int64_t vert_base);
void mga_batch_set_want_src(mga_batch_t *b, int on) { int t; b->want_src = on; for (t = 0; t < b->n_threads; ++t) b->tp[t].want_src = on; }

void mga_batch_set_device_chains(mga_batch_t *b, const mga_gc_hdr_t *hdr, const void *gc_pool, const mg_llchain_t *lc_pool, const mg128_t *a_pool)
{
	b->gc_hdr = hdr, b->gc_pool = (const char*)gc_pool, b->gc_lc = lc_pool, b->gc_a = a_pool, b->gc_rec = mga_gc_rec_bytes();
}

void mga_batch_set_device_plan(mga_batch_t *b, const int64_t *chain_off, int32_t *rev, int64_t n_chain) { b->dp_chain_off = chain_off, b->dp_rev = rev, b->dp_n_chain = n_chain; }

void mga_batch_set_rq_device(mga_batch_t *b, int (*fwd)(void*, int, const int64_t*, const int64_t*, const mg128_t *const*, int32_t *const*, int64_t *const*, int32_t *const*, int, const int64_t*, int, const int32_t*, int, int32_t*),
							 void *ctx, char *harr, int64_t tot)
{
	b->rq_dev_fwd = fwd, b->rq_dev_ctx = ctx, b->rq_harr = harr, b->rq_tot = tot, b->rq_dev_err = 0;
	pthread_mutex_init(&b->rq_dev_mtx, 0);
}

void mga_batch_lchain_par(const mg_idx_t *gi, const mg_mapopt_t *opt, int qlen_max, mga_lchain_par_t *par) /* map-algo.c:377-403 for long reads */
{
	float tmp = expf(-opt->div * gi->k);
	int gap_ref;
	(void)qlen_max;
	if (opt->max_gap_ref > 0) gap_ref = opt->max_gap_ref;
	else if (opt->max_frag_len > 0) gap_ref = opt->max_gap; /* the per-read value max(max_frag_len - qlen, max_gap) is applied inside k_lchain (mga_rescue_par_t::frag_len) */
	else gap_ref = opt->max_gap;
	par->max_dist_x = gap_ref, par->max_dist_y = opt->max_gap;
	par->bw = opt->bw, par->max_skip = opt->max_lc_skip, par->max_iter = opt->max_lc_iter;
	par->min_cnt = opt->min_lc_cnt, par->min_sc = opt->min_lc_score;
	par->chn_pen_gap = opt->chn_pen_gap * tmp, par->chn_pen_skip = opt->chn_pen_skip * tmp;
}

/* ---- MG_M_RMQ (-x asm): the RMQ chainer in phases -------------------------------------------------------------------------
 * A batch under -x asm is a handful of contigs with up to 10^7 anchors each, and one chaining pass over such a contig is seconds of
 * sequential tree work.  The forward pass restarts from empty trees wherever the target (segment, strand) changes
 * (rmq.c: mga_lchain_rmq_fwd), so runs of whole groups are independent: phase 1 sorts every read's anchors (hit order from the
 * device) and cuts them into such runs, phase 2 runs the forward pass of all runs of all reads on all threads, phase 3 backtracks
 * per read and decides about the long-join rescue (map-algo.c:407-417), phases 4-5 repeat 2-3 with bw_long for the reads that
 * need it.  chain_worker then picks the chains up. */
extern __thread int mga_ksort_threads;
typedef struct rq_read_s {
	mg128_t *a; int64_t n;           /* anchors being chained: the read's slice of the batch (pass 1) or `out` of pass 1 (pass 2) */
	int32_t *f, *v, *t; int64_t *p;
	int64_t *cut; int32_t n_cut;     /* run boundaries: cut[0] = 0 .. cut[n_cut] = n */
	mg128_t *out; uint64_t *u; int n_lc, pass2;
} rq_read_t;
typedef struct { int32_t read, k; } rq_task_t;

#define RQ_RUN_MIN 16384 /* anchors per work item, at least */
#define RQ_RUN_MIN_DEV 1024 /* ... of the device pass: a wavefront per run, thousands of them */
#define RQ_RUN_MAX_DEV (1 << 20) /* a run beyond this is the host's: one wavefront walks a run anchor by anchor */

static void rq_cut(rq_read_t *r, int64_t run_min)
{
	int64_t i, last = 0;
	int32_t m = 16;
	r->cut = MGA_MALLOC(int64_t, m + 1), r->n_cut = 0;
	r->cut[0] = 0;
	for (i = 1; i < r->n; ++i)
		if (r->a[i].x >> 32 != r->a[i - 1].x >> 32 && i - last >= run_min) {
			if (r->n_cut + 1 == m) { m <<= 1; r->cut = MGA_REALLOC(int64_t, r->cut, m + 1); }
			r->cut[++r->n_cut] = i, last = i;
		}
	r->cut[++r->n_cut] = r->n;
}

static void rq_arrays(mga_batch_t *b, rq_read_t *r, int64_t i)
{
	if (b->rq_harr) { /* the device pass: slices of the chunk's pinned arrays p | f | v | t (pass 2 has fewer anchors than pass 1: the same slices) */
		const int64_t T = (b->rq_tot + 1) & ~1LL, o = b->a_off[i] - b->a_off[0];
		r->p = (int64_t*)b->rq_harr + o, r->f = (int32_t*)(b->rq_harr + 8 * T) + o, r->v = (int32_t*)(b->rq_harr + 12 * T) + o, r->t = (int32_t*)(b->rq_harr + 16 * T) + o;
		/* the pinned block is neither zeroed by its allocation nor between chunks (whose p | f | v | t cuts differ), and every HOST forward pass -- the DP of an ultra-long -x lr
		 * read, the barrier form, a run the device hands back -- wants its skip marks cleared (lchain.c:166, :270): cheap next to any of them, so always */
		memset(r->t, 0, (size_t)r->n * 4);
		return;
	}
	r->p = MGA_MALLOC(int64_t, r->n); r->f = MGA_MALLOC(int32_t, r->n); r->v = MGA_MALLOC(int32_t, r->n); r->t = MGA_CALLOC(int32_t, r->n);
}

static void rq_prepare_worker_(void *data, int64_t i, int tid)
{
	mga_batch_t *b = (mga_batch_t*)data;
	rq_read_t *r = &b->rq[i];
	int64_t tc = cpu_now();
	(void)tid;
	r->n = b->a_off[i + 1] - b->a_off[i];
	if (b->qlens[i] == 0 || (b->opt.max_qlen > 0 && b->qlens[i] > b->opt.max_qlen)) r->n = 0; /* chain_worker returns early for these */
	if (r->n <= 0) return;
	r->a = (mg128_t*)(b->a + b->a_off[i]); /* every read owns its slice of the staging buffer */
	if ((b->a_is_raw == 2 || b->a_is_raw == 3) && r->n > 1) mga_ksort_128x(r->n, r->a); /* hit order from the device: radix_sort_128x (map-algo.c:189) */
	if (b->a_is_raw == 3 || b->a_is_raw == 5) { /* (5: already sorted -- the CPU tests' oracle anchors) */ r->cut = MGA_MALLOC(int64_t, 2); r->cut[0] = 0, r->cut[1] = r->n, r->n_cut = 1; } /* an ultra-long -x lr read: its first pass is the DP, one work item */
	else rq_cut(r, b->rq_dev_fwd ? RQ_RUN_MIN_DEV : RQ_RUN_MIN);
	rq_arrays(b, r, i);
	CPU_ADD(C_LCCOPY, tc);
}

static void rq_fwd_worker(void *data, int64_t j, int tid)
{
	mga_batch_t *b = ((mga_batch_t**)data)[0];
	const rq_task_t *task = &((const rq_task_t*)((void**)data)[1])[j];
	rq_read_t *r = &b->rq[task->read];
	const mg_mapopt_t *opt = &b->opt;
	int64_t tc = cpu_now();
	(void)tid;
	if ((b->a_is_raw == 3 || b->a_is_raw == 5) && !r->pass2) { /* mg_lchain_dp (map-algo.c:393-403) on this host thread: hchain.c */
		mga_lchain_par_t par;
		mga_batch_lchain_par(b->gi, opt, 0, &par);
		if (opt->max_gap_ref <= 0 && opt->max_frag_len > 0) { const int g = opt->max_frag_len - b->qlens[task->read]; par.max_dist_x = g > opt->max_gap ? g : opt->max_gap; } /* -F: map-algo.c:383-386 */
		mga_lchain_dp_fwd(par.max_dist_x, par.max_dist_y, par.bw, par.max_skip, par.max_iter, par.chn_pen_gap, par.chn_pen_skip, r->n, r->a, r->f, r->p, r->v, r->t);
	} else
	mga_lchain_rmq_fwd(opt->max_gap, opt->max_gap_pre, r->pass2 ? opt->bw_long : opt->bw, opt->max_lc_skip, opt->rmq_size_cap, b->pen_gap, b->pen_skip,
					   r->cut[task->k], r->cut[task->k + 1], r->a, r->f, r->p, r->v, r->t);
	CPU_ADD(r->pass2 ? C_LCRESCUE : C_LCCOPY, tc);
}

/* runs of the RMQ chainer's forward pass: [0] taken by the device, [1] redone on the host because two candidates tied on the priority (the AVL shape decides: k_rmq.hip),
 * [2] ... because the inner window held more candidates than the kernel sorts, [3] kept on the host for their length, [4] device call failed */
static int64_t g_rq_dev_stat[8];
void mga_rq_dev_stats(int64_t *out, int reset) { int k; for (k = 0; k < 8; ++k) { out[k] = __atomic_load_n(&g_rq_dev_stat[k], __ATOMIC_RELAXED); if (reset) __atomic_store_n(&g_rq_dev_stat[k], 0, __ATOMIC_RELAXED); } }

typedef struct { int64_t len; int32_t k; } rq_ord_t;
static int rq_ord_cmp(const void *x, const void *y) { const rq_ord_t *p = (const rq_ord_t*)x, *q = (const rq_ord_t*)y; return p->len > q->len ? -1 : p->len < q->len ? 1 : (p->k > q->k) - (p->k < q->k); }

static int rq_use_dev(const mga_batch_t *b, const rq_read_t *r) { return b->rq_dev_fwd != 0 && !((b->a_is_raw == 3 || b->a_is_raw == 5) && !r->pass2); }

/* the forward pass of ALL runs of some reads' current pass (the same pass for all of them): on the device, a wavefront per run, ONE launch for all the reads (k_rmq.hip:
 * a launch lasts as long as its longest run, and a read's ~2 000 runs fill a quarter of the wave slots); what the device gives back -- a run with tied priorities, an inner
 * window beyond the kernel's sort -- and what is too long for one wavefront goes through the host's exact tree (rmq.c), run by run, on this thread */
static void rq_dev_worker(mga_batch_t *b, int n_reads, const int32_t *reads)
{
	const mg_mapopt_t *opt = &b->opt;
	const int pass2 = b->rq[reads[0]].pass2;
	int64_t n_runs = 0, k, q, *runs, *abs0, *rn, cnt[5] = { 0, 0, 0, 0, 0 }, tc = cpu_now();
	const mg128_t **ra; int32_t **rf, **rv; int64_t **rp;
	rq_ord_t *ord;
	int32_t *order, *status, n_dev = 0;
	int rc = 0, x;
	for (x = 0; x < n_reads; ++x) n_runs += b->rq[reads[x]].n_cut;
	runs = MGA_MALLOC(int64_t, 3 * n_runs + 3); ord = MGA_MALLOC(rq_ord_t, n_runs + 1); order = MGA_MALLOC(int32_t, n_runs + 1); status = MGA_MALLOC(int32_t, n_runs + 1);
	abs0 = MGA_MALLOC(int64_t, n_reads); rn = MGA_MALLOC(int64_t, n_reads); ra = (const mg128_t**)MGA_MALLOC(void*, n_reads); rf = (int32_t**)MGA_MALLOC(void*, n_reads);
	rv = (int32_t**)MGA_MALLOC(void*, n_reads); rp = (int64_t**)MGA_MALLOC(void*, n_reads);
	for (x = 0, q = 0; x < n_reads; ++x) {
		rq_read_t *r = &b->rq[reads[x]];
		abs0[x] = b->a_off[reads[x]] - b->a_off[0], rn[x] = r->n, ra[x] = r->a, rf[x] = r->f, rv[x] = r->v, rp[x] = r->p;
		for (k = 0; k < r->n_cut; ++k, ++q) {
			runs[3 * q] = abs0[x] + r->cut[k], runs[3 * q + 1] = abs0[x] + r->cut[k + 1], runs[3 * q + 2] = abs0[x];
			status[q] = 3;
			if (r->cut[k + 1] - r->cut[k] <= RQ_RUN_MAX_DEV) ord[n_dev].len = r->cut[k + 1] - r->cut[k], ord[n_dev++].k = (int32_t)q;
		}
	}
	qsort(ord, (size_t)n_dev, sizeof *ord, rq_ord_cmp); /* longest first: a launch lasts as long as its longest run */
	for (k = 0; k < n_dev; ++k) order[k] = ord[k].k, status[ord[k].k] = 4;
	if (n_dev > 0) {
		const double tw0 = mga_wtime();
		pthread_mutex_lock(&b->rq_dev_mtx);
		const double tw1 = mga_wtime();
		rc = b->rq_dev_fwd(b->rq_dev_ctx, n_reads, abs0, rn, ra, rf, rp, rv, (int)n_runs, runs, n_dev, order, pass2 ? opt->bw_long : opt->bw, status);
		if (rc < 0 && !b->rq_dev_err) { b->rq_dev_err = 1; snprintf(b->rq_dev_errmsg, sizeof b->rq_dev_errmsg, "%s", mga_last_error()); } /* (the message is this thread's) */
		pthread_mutex_unlock(&b->rq_dev_mtx);
		if (g_cpu_on) fprintf(stderr, "[rq] pass %d of %d reads: %d runs (longest %ld anchors) through the device in %.1f ms (+ %.1f ms waiting for it)\n", pass2 + 1, n_reads, n_dev, (long)ord[0].len, (mga_wtime() - tw1) * 1e3, (tw1 - tw0) * 1e3);
		if (rc < 0) for (k = 0; k < n_dev; ++k) status[order[k]] = 4; /* (the task graph runs to its end on the host's tree; mga_batch_chain() then FAILS the chunk with the device's message) */
	}
	for (x = 0, q = 0; x < n_reads; ++x) {
		rq_read_t *r = &b->rq[reads[x]];
		for (k = 0; k < r->n_cut; ++k, ++q) {
			if (status[q] == 0) { ++cnt[0]; continue; }
			++cnt[status[q] <= 4 ? status[q] : 4];
			memset(r->t + r->cut[k], 0, (size_t)(r->cut[k + 1] - r->cut[k]) * 4); /* the marks of a run stay inside it */
			mga_lchain_rmq_fwd(opt->max_gap, opt->max_gap_pre, pass2 ? opt->bw_long : opt->bw, opt->max_lc_skip, opt->rmq_size_cap, b->pen_gap, b->pen_skip,
							   r->cut[k], r->cut[k + 1], r->a, r->f, r->p, r->v, r->t);
		}
	}
	for (k = 0; k < 5; ++k) if (cnt[k]) __atomic_fetch_add(&g_rq_dev_stat[k], cnt[k], __ATOMIC_RELAXED);
	free(runs); free(ord); free(order); free(status); free(abs0); free(rn); free((void*)ra); free(rf); free(rv); free(rp);
	CPU_ADD(pass2 ? C_LCRESCUE : C_LCCOPY, tc);
}

static void rq_finish_worker_(void *data, int64_t i, int tid)
{
	mga_batch_t *b = (mga_batch_t*)data;
	rq_read_t *r = &b->rq[i];
	const mg_mapopt_t *opt = &b->opt;
	int64_t tc = cpu_now();
	(void)tid;
	if (r->n <= 0 || r->f == 0) return;
	if (!r->pass2) {
		r->out = mga_lchain_rmq_finish2(opt->bw, opt->min_lc_cnt, opt->min_lc_score, r->n, r->a, r->f, r->p, r->v, r->t, &r->n_lc, &r->u, b->rq_harr != 0);
		r->f = r->v = r->t = 0, r->p = 0; free(r->cut); r->cut = 0;
		if (opt->bw_long > opt->bw && (opt->flag & (MG_M_SPLICE | MG_M_SR)) == 0 && r->n_lc > 1) { /* map-algo.c:407-417 */
			const int32_t qlen = b->qlens[i], st = (int32_t)r->out[0].y, en = (int32_t)r->out[(int32_t)r->u[0] - 1].y;
			if (qlen - (en - st) > opt->rmq_rescue_size || qlen - (en - st) > qlen * opt->rmq_rescue_ratio) {
				int32_t k;
				int64_t n_a = 0;
				for (k = 0; k < r->n_lc; ++k) n_a += (int32_t)r->u[k];
				free(r->u); r->u = 0;
				mga_ksort_128x(n_a, r->out);
				r->a = r->out, r->n = n_a, r->pass2 = 1;
				if (b->rq_dev_fwd && b->a_is_raw == 2) { /* the re-chained anchors go back into the read's slice of the (pinned) staging buffer: the device pass uploads from there, and the
				                                          * chunk-level positions of the second pass are the first pass's (n_a <= the slice's length) */
					mg128_t *slice = (mg128_t*)(b->a + b->a_off[i]);
					memcpy(slice, r->out, (size_t)n_a * sizeof(mg128_t));
					free(r->out); r->out = 0, r->a = slice;
				}
				rq_cut(r, b->rq_dev_fwd ? RQ_RUN_MIN_DEV : RQ_RUN_MIN); rq_arrays(b, r, i);
			}
		}
		CPU_ADD(C_LCCOPY, tc);
	} else {
		mg128_t *a2 = mga_lchain_rmq_finish2(opt->bw_long, opt->min_lc_cnt, opt->min_lc_score, r->n, r->a, r->f, r->p, r->v, r->t, &r->n_lc, &r->u, b->rq_harr != 0);
		r->f = r->v = r->t = 0, r->p = 0; free(r->cut); r->cut = 0;
		free(r->out); r->out = a2;
		CPU_ADD(C_LCRESCUE, tc);
	}
}

/* the big sorts of a contig fan out over the pool (ksortx.c: mga_ksort_threads is per thread) for the duration of the task that sorts, and never beyond it */
static void rq_prepare_worker(void *data, int64_t i, int tid) { mga_ksort_threads = ((mga_batch_t*)data)->n_threads; rq_prepare_worker_(data, i, tid); mga_ksort_threads = 1; }
static void rq_finish_worker(void *data, int64_t i, int tid) { mga_ksort_threads = ((mga_batch_t*)data)->n_threads; rq_finish_worker_(data, i, tid); mga_ksort_threads = 1; }

static void rq_run_fwd(mga_batch_t *b, int pass2)
{
	int64_t n_task = 0, j = 0;
	int i, k;
	rq_task_t *task;
	void *arg[2];
	for (i = 0; i < b->n; ++i) if (b->rq[i].f && b->rq[i].pass2 == pass2) n_task += b->rq[i].n_cut;
	if (n_task == 0) return;
	task = MGA_MALLOC(rq_task_t, n_task);
	for (i = 0; i < b->n; ++i)
		if (b->rq[i].f && b->rq[i].pass2 == pass2)
			for (k = 0; k < b->rq[i].n_cut; ++k) task[j].read = i, task[j++].k = k;
	arg[0] = b, arg[1] = task;
	mga_parallel_for(b->n_threads, n_task, rq_fwd_worker, arg);
	free(task);
}

/* The phases as a task graph (round 4).  With barriers between the phases a batch of 8 contigs keeps 8 of 16 threads busy while it sorts and backtracks (one thread per read,
 * ~1 s per 50 Mbp contig and pass) and every read waits for the slowest: [measured, 10 x 50 Mbp contigs vs a 500 Mbp graph] 65 CPU-s of RMQ chaining in 7.9 s of wall time.
 * The dependencies are per READ -- sort(i) -> forward runs (i, k) -> backtrack(i) -> forward runs with bw_long (i, k) -> backtrack(i) -- so the threads take tasks from one queue:
 * finished forward passes hand their read's backtrack to the FRONT of the queue, a read's forward runs are queued as one block behind the others', and read i backtracks while the
 * forward runs of read i+1 are still being taken.  Same calls on the same data per read: the bytes cannot change.  MGA_RQ_BARRIERS=1: the phases with barriers (A/B). */
typedef struct {
	mga_batch_t *b;
	pthread_mutex_t mtx; pthread_cond_t cv;
	rq_task_t *lo; int64_t lo_head, lo_tail, lo_cap;  /* forward runs (FIFO) */
	int32_t *hi; int32_t n_hi;                         /* reads whose sort (>= 0: read, < 0: ~read = backtrack) is due: taken first, last in first out */
	int32_t *pending;                                  /* forward runs of a read not finished yet */
	int n_open;                                        /* reads with work left */
} rq_sched_t;

static void rq_push_fwd(rq_sched_t *S, int read) /* (locked) */
{
	const rq_read_t *r = &S->b->rq[read];
	int k;
	if (S->lo_tail + r->n_cut > S->lo_cap) { S->lo_cap = (S->lo_tail + r->n_cut) * 2; S->lo = MGA_REALLOC(rq_task_t, S->lo, S->lo_cap); }
	if (rq_use_dev(S->b, r)) { /* ONE task: all runs of the read through the device (k = -1) */
		S->lo[S->lo_tail].read = read, S->lo[S->lo_tail++].k = -1;
		S->pending[read] = 1;
		return;
	}
	for (k = 0; k < r->n_cut; ++k) S->lo[S->lo_tail].read = read, S->lo[S->lo_tail++].k = k;
	S->pending[read] = r->n_cut;
}

static void rq_sched_worker(void *data, int64_t j_, int tid)
{
	rq_sched_t *S = (rq_sched_t*)data;
	mga_batch_t *b = S->b;
	(void)j_;
	pthread_mutex_lock(&S->mtx);
	for (;;) {
		while (S->n_hi == 0 && S->lo_head == S->lo_tail && S->n_open > 0) pthread_cond_wait(&S->cv, &S->mtx);
		if (S->n_hi > 0) {
			const int32_t h = S->hi[--S->n_hi], read = h >= 0 ? h : ~h;
			rq_read_t *r = &b->rq[read];
			pthread_mutex_unlock(&S->mtx);
			if (h >= 0) rq_prepare_worker(b, read, tid); else rq_finish_worker(b, read, tid);
			pthread_mutex_lock(&S->mtx);
			if (r->n > 0 && r->f != 0) rq_push_fwd(S, read); /* sorted and cut, or a second pass with bw_long is due */
			else --S->n_open;
			pthread_cond_broadcast(&S->cv);
		} else if (S->lo_head < S->lo_tail) {
			const rq_task_t t = S->lo[S->lo_head++];
			void *arg[2];
			if (t.k < 0) { /* a device task: every other read whose forward pass of the SAME pass is waiting in the queue goes into the same launch */
				int32_t *rd = MGA_MALLOC(int32_t, S->lo_tail - S->lo_head + 1), n_rd = 0;
				int64_t j, w_ = S->lo_head;
				int x;
				rd[n_rd++] = t.read;
				for (j = S->lo_head; j < S->lo_tail; ++j) {
					if (S->lo[j].k < 0 && b->rq[S->lo[j].read].pass2 == b->rq[t.read].pass2) rd[n_rd++] = S->lo[j].read;
					else S->lo[w_++] = S->lo[j];
				}
				S->lo_tail = w_;
				pthread_mutex_unlock(&S->mtx);
				rq_dev_worker(b, n_rd, rd);
				pthread_mutex_lock(&S->mtx);
				for (x = 0; x < n_rd; ++x) if (--S->pending[rd[x]] == 0) S->hi[S->n_hi++] = ~rd[x];
				pthread_cond_broadcast(&S->cv);
				free(rd);
				continue;
			}
			pthread_mutex_unlock(&S->mtx);
			arg[0] = b, arg[1] = (void*)&t;
			rq_fwd_worker(arg, 0, tid);
			pthread_mutex_lock(&S->mtx);
			if (--S->pending[t.read] == 0) { S->hi[S->n_hi++] = ~t.read; pthread_cond_broadcast(&S->cv); }
		} else break; /* n_open == 0 */
	}
	pthread_mutex_unlock(&S->mtx);
}

static void rq_chain_all(mga_batch_t *b)
{
	const int prof = getenv("MGA_DEBUG_PIPE") && atoi(getenv("MGA_DEBUG_PIPE")) > 0; /* wall time: where a -x asm job's host time goes (long queries) */
	const int64_t n_a = b->n > 0 ? b->a_off[b->n] - b->a_off[0] : 0;
	double t[6];
	b->rq = MGA_CALLOC(rq_read_t, b->n > 0 ? b->n : 1);
	t[0] = mga_wtime();
	if (getenv("MGA_RQ_BARRIERS") && atoi(getenv("MGA_RQ_BARRIERS")) > 0) {
		b->rq_dev_fwd = 0; /* (the A/B form with barriers between the phases is the host's) */
		mga_parallel_for(b->n_threads, b->n, rq_prepare_worker, b);
		t[1] = mga_wtime();
		rq_run_fwd(b, 0);
		t[2] = mga_wtime();
		mga_parallel_for(b->n_threads, b->n, rq_finish_worker, b);
		t[3] = mga_wtime();
		rq_run_fwd(b, 1);
		t[4] = mga_wtime();
		mga_parallel_for(b->n_threads, b->n, rq_finish_worker, b);
		t[5] = mga_wtime();
		if (prof) fprintf(stderr, "[rq] %d queries, %ld anchors, %d threads: sort %.3f s, forward pass %.3f, backtrack + rescue decision %.3f, forward pass (bw_long) %.3f, backtrack %.3f\n", b->n, (long)n_a,
						  b->n_threads, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
		return;
	}
	if (b->n > 0) {
		rq_sched_t S;
		int i;
		memset(&S, 0, sizeof S);
		S.b = b, S.n_open = b->n;
		pthread_mutex_init(&S.mtx, 0); pthread_cond_init(&S.cv, 0);
		S.lo_cap = 1024, S.lo = MGA_MALLOC(rq_task_t, S.lo_cap);
		S.hi = MGA_MALLOC(int32_t, b->n), S.pending = MGA_CALLOC(int32_t, b->n);
		for (i = b->n - 1; i >= 0; --i) S.hi[S.n_hi++] = i; /* (taken from the top: read 0 first) */
		mga_parallel_for(b->n_threads, b->n_threads, rq_sched_worker, &S);
		pthread_mutex_destroy(&S.mtx); pthread_cond_destroy(&S.cv);
		free(S.lo); free(S.hi); free(S.pending);
	}
	if (prof) fprintf(stderr, "[rq] %d queries, %ld anchors, %d threads: sort, forward passes and backtracks of both passes as one task graph: %.3f s\n", b->n, (long)n_a, b->n_threads, mga_wtime() - t[0]);
}

uint32_t mga_read_hash(const char *qname, int qlen, int seed) /* map-algo.c:362-364 */
{
	uint32_t hash = qname ? mga_hash_str(qname) : 0;
	hash ^= mga_hash_u32((uint32_t)qlen) + mga_hash_u32((uint32_t)seed);
	return mga_hash_u32(hash);
}

static void chain_worker(void *data, int64_t i, int tid)
{
	mga_batch_t *b = (mga_batch_t*)data;
	const mg_mapopt_t *opt = &b->opt;
	const mg_idx_t *gi = b->gi;
	const int qlen = b->qlens[i];
	const char *seq = b->seqs[i], *qname = b->qnames ? b->qnames[i] : 0;
	uint32_t hash;
	int32_t n_lc = 0, k;
	int64_t n_a = 0;
	uint64_t *u = 0;
	mg128_t *a = 0;
	mg_gchains_t *gcs;

	int64_t tc = cpu_now();
	b->gcs[i] = 0;
	if (qlen == 0) return; /* map-algo.c:359-360 */
	if (opt->max_qlen > 0 && qlen > opt->max_qlen) return;
	if (b->gc_hdr && b->gc_hdr[i].status == 0) { /* chained on the device: flat records -> the reference's object (div and MAPQ through the host's libm) */
		const mga_gc_hdr_t *h = &b->gc_hdr[i];
		gcs = mga_gchains_from_flat(h->n_gc, b->gc_pool + (size_t)h->gc_off * b->gc_rec, h->n_lc, b->gc_lc + h->lc_off, b->gc_a ? h->n_a : 0, b->gc_a ? b->gc_a + h->a_off : 0, /* (no anchors on the host when the gap list was made on the device) */
									b->rep_len[i], qlen, b->n_mz[i], opt->min_gc_score);
		b->gcs[i] = gcs;
		CPU_ADD(C_GCPOST, tc);
		goto plan;
	}
	hash = mga_read_hash(qname, qlen, opt->seed);

	if (b->a_is_raw) { /* MG_M_RMQ: the RMQ chainer is the primary chainer (map-algo.c:397-399); both of its passes ran in mga_batch_chain's phases */
		a = b->rq[i].out, u = b->rq[i].u, n_lc = b->rq[i].n_lc;
		b->rq[i].out = 0, b->rq[i].u = 0;
	} else { /* chains of the GPU DP */
		n_lc = b->nu[i], n_a = b->nb[i];
		if (n_lc > 0) {
			u = MGA_MALLOC(uint64_t, n_lc); memcpy(u, b->u + b->a_off[i], (size_t)n_lc * 8);
			a = MGA_MALLOC(mg128_t, n_a); memcpy(a, b->a + b->a_off[i], (size_t)n_a * 16);
		}
	}
	CPU_ADD(C_LCCOPY, tc);
	/* long-join rescue (map-algo.c:407-417) */
	if (b->a_is_raw) { /* done in the phases */
	} else if (b->rescue_flag) { /* the kernel evaluated the condition: 1 = done there, 2 = due but deferred to the host (priority tie) */
		if (b->rescue_flag[i] == 2) goto do_rescue;
	} else if (opt->bw_long > opt->bw && (opt->flag & (MG_M_SPLICE | MG_M_SR)) == 0 && n_lc > 1) {
		int32_t st = (int32_t)a[0].y, en = (int32_t)a[(int32_t)u[0] - 1].y;
		if (qlen - (en - st) > opt->rmq_rescue_size || qlen - (en - st) > qlen * opt->rmq_rescue_ratio) {
do_rescue:;
			mg128_t *a2;
			for (k = 0, n_a = 0; k < n_lc; ++k) n_a += (int32_t)u[k];
			free(u); u = 0;
			mga_ksort_128x(n_a, a);
			a2 = mga_lchain_rmq(opt->max_gap, opt->max_gap_pre, opt->bw_long, opt->max_lc_skip, opt->rmq_size_cap, opt->min_lc_cnt, opt->min_lc_score,
								b->pen_gap, b->pen_skip, n_a, a, &n_lc, &u);
			free(a); a = a2;
		}
	}
	CPU_ADD(C_LCRESCUE, tc);
	{ /* chain records, clean-up, graph chaining, bridging, ordering and filters: gc_core.h on this host thread (map-algo.c:422-474) */
		const int32_t *mini = b->mini_pos + b->mini_off[i];
		const int32_t n_mini = (int32_t)(b->mini_off[i + 1] - b->mini_off[i]);
		int64_t n_anchor = 0;
		for (k = 0; k < n_lc; ++k) n_anchor += (int32_t)u[k];
		gcs = mga_gchain_host_read(gi, b->seg_len, opt, b->pen_gap, qlen, hash, n_lc, u, a, (int32_t)n_anchor, n_mini, mini, seq, b->rep_len[i], b->n_mz[i], 0, 0);
	}
	free(u); u = 0;
	free(a);
	CPU_ADD(C_GCGEN, tc);
	b->gcs[i] = gcs;
plan:
	if ((opt->flag & MG_M_CIGAR) && b->dp_chain_off && b->gc_hdr && b->gc_hdr[i].status == 0) { /* the gaps were listed on the device (k_plan.hip): only the strand of each printed line is decided here */
		read_plan_t *pl = &b->plan[i];
		int64_t c = b->dp_chain_off[i];
		int rev_sign = 0;
		pl->tid = -1, pl->n_gc = gcs->n_gc;
		pl->chain_id = MGA_MALLOC(int64_t, gcs->n_gc > 0 ? gcs->n_gc : 1);
		for (k = 0; k < gcs->n_gc; ++k) {
			const mg_gchain_t *gc = &gcs->gc[k];
			pl->chain_id[k] = -1;
			if ((gc->id != gc->parent && !(opt->flag & MG_M_PRINT_2ND)) || gc->cnt == 0) continue;
			if (mga_gaf_chain_rev(gi->g, gcs, gc, opt->flag)) rev_sign = 1;
			b->dp_rev[c] = rev_sign;
			pl->chain_id[k] = c++;
		}
		assert(c == b->dp_chain_off[i + 1]);
		CPU_ADD(C_PLAN, tc);
	} else if (opt->flag & MG_M_CIGAR) { /* list the gaps of every chain for the WFA kernel */
		read_plan_t *pl = &b->plan[i];
		mga_tpool_t *tp = &b->tp[tid];
		pl->tid = tid, pl->n_gc = gcs->n_gc;
		int rev_sign = 0; /* carried from chain to chain like mg_write_gaf does (format.c:123) */
		pl->item_off = MGA_MALLOC(int64_t, gcs->n_gc + 1);
		if (b->want_text) pl->chain_id = MGA_MALLOC(int64_t, gcs->n_gc > 0 ? gcs->n_gc : 1);
		for (k = 0; k < gcs->n_gc; ++k) {
			const mg_gchain_t *gc = &gcs->gc[k];
			pl->item_off[k] = tp->n_item;
			if (b->want_text) {
				pl->chain_id[k] = -1;
				if ((gc->id != gc->parent && !(opt->flag & MG_M_PRINT_2ND)) || gc->cnt == 0) continue; /* not printed (format.c:135-136): no alignment needed */
			}
			const int64_t vert_beg = tp->n_vert;
			if (b->want_text) { int32_t j; for (j = 0; j < gc->cnt; ++j) { MGA_GROW(uint32_t, tp->vert, tp->n_vert, tp->m_vert); tp->vert[tp->n_vert++] = gcs->lc[gc->off + j].v; } } /* the chain's walk: what the text kernel prints, and what the device splices the gaps' targets from */
			mga_plan_cigar(gi->g, gi->es, gcs, k, b->q_off ? b->q_off[i] : 0, tp, vert_beg);
			if (b->want_text) { /* what the text kernel needs to know about this chain */
				mga_txt_chain_t *c;
				const int32_t off_a0 = gcs->lc[gc->off].off;
				if (mga_gaf_chain_rev(gi->g, gcs, gc, opt->flag)) rev_sign = 1;
				MGA_GROW(mga_txt_chain_t, tp->chain, tp->n_chain, tp->m_chain);
				c = &tp->chain[tp->n_chain];
				c->item_beg = pl->item_off[k], c->item_end = tp->n_item, c->prob_base = 0, c->q_base = b->q_off ? b->q_off[i] : 0;
				c->vert_beg = vert_beg, c->vert_cnt = gc->cnt;
				c->qs = gc->qs, c->qe = gc->qe, c->ps = gc->ps, c->pe = gc->pe;
				c->ss = (int32_t)gcs->a[off_a0].x + 1 - (int32_t)(gcs->a[off_a0].y >> 32 & 0xff); /* galign.c:128-129 */
				c->ee = (int32_t)gcs->a[off_a0 + gc->n_anchor - 1].x + 1;
				c->rev_sign = rev_sign;
				pl->chain_id[k] = tp->n_chain++;
			}
		}
		pl->item_off[gcs->n_gc] = tp->n_item;
		CPU_ADD(C_PLAN, tc);
	}
}

int mga_batch_chain(mga_batch_t *b, const int32_t *n_mz, const int32_t *rep_len, const int32_t *mini_pos, const int64_t *mini_off,
					const int32_t *nu, const int32_t *nb, const uint64_t *u, const mg128_t *a, const int64_t *a_off, int a_is_raw, const int32_t *rescue_flag)
{
	int t;
	if (b->seg_len == 0) { uint32_t s_; b->seg_len = MGA_MALLOC(int32_t, b->gi->g->n_seg + 1); for (s_ = 0; s_ < b->gi->g->n_seg; ++s_) b->seg_len[s_] = b->gi->g->seg[s_].len; }
	b->rescue_flag = rescue_flag;
	b->n_mz = n_mz, b->rep_len = rep_len, b->mini_pos = mini_pos, b->mini_off = mini_off;
	b->nu = nu, b->nb = nb, b->u = u, b->a = a, b->a_off = a_off, b->a_is_raw = a_is_raw;
	if (a_is_raw) rq_chain_all(b);
	mga_parallel_for(b->n_threads, b->n, chain_worker, b);
	if (b->rq) { free(b->rq); b->rq = 0; }
	for (t = 0; t < b->n_threads; ++t) {
		b->tp_prob_base[t + 1] = b->tp_prob_base[t] + b->tp[t].n_prob;
		b->tp_t_base[t + 1] = b->tp_t_base[t] + b->tp[t].n_t;
		b->tp_item_base[t + 1] = b->tp_item_base[t] + b->tp[t].n_item;
		b->tp_chain_base[t + 1] = b->tp_chain_base[t] + b->tp[t].n_chain;
		b->tp_vert_base[t + 1] = b->tp_vert_base[t] + b->tp[t].n_vert;
	}
	if (b->rq_dev_err) { mga_set_error("RMQ forward pass on the device failed: %s", b->rq_dev_errmsg); return -1; } /* fail loudly: a device error must not pass for a slow job */
	return 0;
}

int64_t mga_batch_n_wfa(const mga_batch_t *b) { return b->tp_prob_base[b->n_threads]; }
int64_t mga_batch_wfa_target_bytes(const mga_batch_t *b) { return b->tp_t_base[b->n_threads]; }

typedef struct { const mga_batch_t *b; mga_wfa_prob_t *prob; char *tseq; mga_plan_src_t *src; int64_t vert_base; } export_t;
static void export_worker(void *data, int64_t t, int tid)
{
	export_t *e = (export_t*)data;
	const mga_batch_t *b = e->b;
	const mga_tpool_t *tp = &b->tp[t];
	int64_t j;
	int64_t tc = cpu_now();
	(void)tid;
	if (!tp->want_src) memcpy(e->tseq + b->tp_t_base[t], tp->tseq, (size_t)tp->n_t);
	for (j = 0; j < tp->n_prob; ++j) {
		e->prob[b->tp_prob_base[t] + j] = tp->prob[j];
		e->prob[b->tp_prob_base[t] + j].t_off += b->tp_t_base[t];
		if (tp->want_src) { e->src[b->tp_prob_base[t] + j] = tp->src[j]; e->src[b->tp_prob_base[t] + j].lc0 += b->tp_vert_base[t] + e->vert_base; }
	}
	CPU_ADD(C_EXPORT, tc);
}

void mga_batch_wfa_export(const mga_batch_t *b, mga_wfa_prob_t *prob, char *tseq) { mga_batch_wfa_export_src(b, prob, tseq, 0, 0); }

/* (want_src: src[] receives one descriptor per problem, its first walk vertex as an index into the flattened vertex array + vert_base; tseq is not written) */
void mga_batch_wfa_export_src(const mga_batch_t *b, mga_wfa_prob_t *prob, char *tseq, mga_plan_src_t *src, int64_t vert_base)
{
	export_t e;
	e.b = b, e.prob = prob, e.tseq = tseq, e.src = src, e.vert_base = vert_base;
	mga_parallel_for(b->n_threads, b->n_threads, export_worker, &e);
}

/* text mode: plan items, printed chains and their vertices of all pools, flattened in pool order with global offsets */
int64_t mga_batch_n_items(const mga_batch_t *b) { return b->tp_item_base[b->n_threads]; }
int64_t mga_batch_n_chains(const mga_batch_t *b) { return b->tp_chain_base[b->n_threads]; }
int64_t mga_batch_n_verts(const mga_batch_t *b) { return b->tp_vert_base[b->n_threads]; }
typedef struct { const mga_batch_t *b; mga_cigitem_t *item; mga_txt_chain_t *chain; uint32_t *vert; } texport_t;
static void text_export_worker(void *data, int64_t t, int tid)
{
	texport_t *e = (texport_t*)data;
	const mga_batch_t *b = e->b;
	const mga_tpool_t *tp = &b->tp[t];
	int64_t j;
	(void)tid;
	memcpy(e->item + b->tp_item_base[t], tp->item, (size_t)tp->n_item * sizeof(mga_cigitem_t));
	memcpy(e->vert + b->tp_vert_base[t], tp->vert, (size_t)tp->n_vert * 4);
	for (j = 0; j < tp->n_chain; ++j) {
		mga_txt_chain_t *c = &e->chain[b->tp_chain_base[t] + j];
		*c = tp->chain[j];
		c->item_beg += b->tp_item_base[t], c->item_end += b->tp_item_base[t], c->vert_beg += b->tp_vert_base[t], c->prob_base = b->tp_prob_base[t];
	}
}
void mga_batch_text_export(const mga_batch_t *b, mga_cigitem_t *item, mga_txt_chain_t *chain, uint32_t *vert)
{
	texport_t e;
	e.b = b, e.item = item, e.chain = chain, e.vert = vert;
	mga_parallel_for(b->n_threads, b->n_threads, text_export_worker, &e);
}

static void finish_worker(void *data, int64_t i, int tid)
{
	mga_batch_t *b = (mga_batch_t*)data;
	mg_gchains_t *gcs = b->gcs[i];
	read_plan_t *pl = &b->plan[i];
	int32_t k;
	int64_t tc = cpu_now();
	(void)tid;
	if (gcs == 0 || pl->item_off == 0) return;
	for (k = 0; k < gcs->n_gc; ++k) {
		const mga_tpool_t *tp = &b->tp[pl->tid];
		int r = mga_apply_cigar(gcs, k, tp->item + pl->item_off[k], pl->item_off[k + 1] - pl->item_off[k], b->tp_prob_base[pl->tid], &b->src);
		if (r < 0) { b->err = r; return; }
	}
	CPU_ADD(C_APPLY, tc);
	mga_gen_ds(b->gi->es, b->seqs[i], gcs);
	CPU_ADD(C_DS, tc);
}

static int batch_finish(mga_batch_t *b);

int mga_batch_finish(mga_batch_t *b, const mga_wfa_res_t *res, const uint32_t *pool)
{
	memset(&b->src, 0, sizeof b->src);
	b->src.res = res, b->src.pool = pool;
	return batch_finish(b);
}

/* same with the CIGARs gathered in problem order by the device (k_wfa_sched.hip): ncig[j] operators at ord + off[j] */
int mga_batch_finish_ordered(mga_batch_t *b, const int32_t *ncig, const int64_t *off, const uint32_t *ord)
{
	memset(&b->src, 0, sizeof b->src);
	b->src.ncig = ncig, b->src.off = off, b->src.ord = ord;
	return batch_finish(b);
}

static int batch_finish(mga_batch_t *b)
{
	if (!(b->opt.flag & MG_M_CIGAR)) return 0;
	b->err = 0;
	mga_parallel_for(b->n_threads, b->n, finish_worker, b);
	if (b->err == -1) mga_set_error("a gap came back from the WFA ladder without an alignment (status != OK): the ladder's last tier or its chained fallback (k_wfa_sched.hip: wfs_fallback, miniwfa.c:776-834) gave up on it");
	else if (b->err < 0) mga_set_error("stitched CIGAR is inconsistent with the chain coordinates");
	return b->err;
}

mg_gchains_t **mga_batch_take_results(mga_batch_t *b) { mg_gchains_t **r = b->gcs; b->gcs = 0; return r; }

void mga_batch_stats(const mga_batch_t *b, mga_stats_t *st)
{
	int t;
	for (t = 0; t < b->n_threads; ++t) st->wfa_t_bases += b->tp[t].wfa_t_bases, st->wfa_q_bases += b->tp[t].wfa_q_bases;
	st->n_wfa += b->tp_prob_base[b->n_threads];
}

void mga_batch_destroy(mga_batch_t *b)
{
	int i;
	if (b == 0) return;
	for (i = 0; i < b->n; ++i) { free(b->plan[i].item_off); free(b->plan[i].chain_id); }
	for (i = 0; i < b->n_threads; ++i) tpool_put(&b->tp[i]);
	if (b->gcs) { for (i = 0; i < b->n; ++i) mg_gchain_free(b->gcs[i]); free(b->gcs); }
	free(b->seg_len);
	free(b->plan); free(b->tp); free(b->tp_prob_base); free(b->tp_t_base); free(b->tp_item_base); free(b->tp_chain_base); free(b->tp_vert_base);
	free(b);
}

/* GAF text of one chunk: T pieces in read order; in text mode cg:Z / ds:Z are copied from the device's output.
 * Round 5: when the chunk is the NEXT one the job's output is waiting for (the usual case: chunks finish nearly in order), its lines are written STRAIGHT to their place in
 * the output buffer -- a measuring pass (the same formatter with empty payloads + the payload lengths the device reported) gives every piece its offset, the writing pass
 * formats into windows of the output (gaf.c: MGA_KS_WINDOW).  [measured, round 4] the text went device -> pinned staging -> piece -> output: two host copies of 1.1 GB per
 * 125 000 reads; now one.  A chunk that finishes ahead of its predecessors takes the old way (pieces, copied when its turn comes). */
typedef struct { int (*try_reserve)(void *ctx, int64_t bytes, char **dst); void (*commit)(void *ctx, int64_t bytes); void *ctx; } gaf_sink_t;
typedef struct { mga_batch_t *b; int n, T, mode; kstring_t *part; int64_t *bytes, *off; char *dst; } gafw_t; /* mode 0: into the pieces; 1: measure; 2: into windows of dst */

static void gaf_worker(void *data, int64_t t, int tid)
{
	gafw_t *w = (gafw_t*)data;
	mga_batch_t *bt = w->b;
	const int n = w->n;
	int64_t b = (int64_t)n * t / w->T, e = (int64_t)n * (t + 1) / w->T, i, payload = 0;
	kstring_t win, *out = &w->part[t];
	mga_chain_text_t *txt = 0;
	int32_t m_txt = 0;
	int64_t tc = cpu_now();
	(void)tid;
	if (w->mode == 2) { win.s = w->dst + w->off[t], win.l = 0, win.m = MGA_KS_WINDOW; out = &win; mga_gaf_window_limit((size_t)w->bytes[t]); }
	else if (w->mode == 0) { /* one allocation per piece: a base-aligned read prints about one byte per base (cg + ds), an unaligned one ~120 bytes */
		size_t est = 4096;
		for (i = b; i < e; ++i) est += (bt->opt.flag & MG_M_CIGAR) ? (size_t)bt->qlens[i] + 512 : 512;
		if (est < 0xfffffff0u && est > out->m) { size_t cap; char *p = strbuf_get(est, &cap); if (cap > 0xfffffff0u) cap = 0xfffffff0u; free(out->s); out->s = p, out->m = (unsigned)cap, out->l = 0; }
	} else out->l = 0; /* (measure: the piece's own buffer serves as scratch for the lines without their payloads) */
	for (i = b; i < e; ++i) {
		int32_t ql = bt->qlens[i], k;
		const mg_gchains_t *gcs = bt->gcs[i];
		const read_plan_t *pl = &bt->plan[i];
		const mga_chain_text_t *tx = 0;
		if (bt->txt_res && gcs && pl->chain_id) {
			if (gcs->n_gc > m_txt) { m_txt = gcs->n_gc + 8; txt = MGA_REALLOC(mga_chain_text_t, txt, m_txt); }
			for (k = 0; k < gcs->n_gc; ++k) {
				txt[k].cg = txt[k].ds = 0;
				if (pl->chain_id[k] >= 0) {
					const mga_txt_res_t *r = &bt->txt_res[pl->tid < 0 ? pl->chain_id[k] : bt->dp_n_chain + bt->tp_chain_base[pl->tid] + pl->chain_id[k]]; /* device-planned chains first, then the pools' */
					txt[k].cg = bt->txt_pool + r->txt_off, txt[k].ds = txt[k].cg + r->cg_len;
					txt[k].cg_len = r->cg_len, txt[k].ds_len = r->ds_len, txt[k].mlen = r->mlen, txt[k].blen = r->blen;
					if (w->mode == 1) { /* printed or not is the formatter's decision: count what it would copy by letting it copy nothing */
						if (!((gcs->gc[k].id != gcs->gc[k].parent && !(bt->opt.flag & MG_M_PRINT_2ND)) || gcs->gc[k].cnt == 0)) payload += (int64_t)r->cg_len + r->ds_len;
						txt[k].cg_len = txt[k].ds_len = 0;
					}
				}
			}
			tx = txt;
		}
		if (w->mode == 1 && out->l > (1u << 20)) { payload += out->l; out->l = 0; } /* (the scratch does not have to hold the whole piece) */
		mga_write_gaf_append(out, bt->gi->g, gcs, 1, &ql, bt->qnames ? bt->qnames[i] : "*", bt->opt.flag, tx);
		if (w->mode != 1) { mg_gchain_free(bt->gcs[i]); bt->gcs[i] = 0; }
	}
	if (w->mode == 1) { w->bytes[t] = payload + out->l; out->l = 0; }
	else if (w->mode == 2) {
		if ((int64_t)win.l != w->bytes[t]) { fprintf(stderr, "[E::%s] GAF piece %ld: %u bytes written where %ld were measured\n", __func__, (long)t, win.l, (long)w->bytes[t]); abort(); } /* the two passes run the same formatter: cannot happen, and must not go unnoticed */
	}
	free(txt);
	CPU_ADD(C_GAF, tc);
}

/* ------------------------------------------------------------------------------------------------
 * device orchestration
 *
 * A batch is cut into chunks of MGA_CHUNK reads (default 16384).  MGA_PIPE pipeline threads (default 4), each with
 * its own HIP stream context, device buffers and pinned staging buffers, pull chunks from a shared counter and run
 * the stage sequence above on them; one token per GPU phase staggers them so that the GPU work of one chunk
 * overlaps the host work of the others.
 * ---------------------------------------------------------------------------------------------- */
#include <pthread.h>

struct gpu_token_s;
typedef struct { /* one pipeline context: HIP stream + grow-only device and pinned buffers, reused from chunk to chunk */
	mga_sctx_t *sc;
	struct gpu_token_s *tok_front, *tok_wfa; /* the GPU phase tokens of the stream this context belongs to (NULL: the process-wide pair -- mg_tbuf_t contexts) */
	union {
		struct {
			mga_dbuf_t seq, qoff, cnt, mzoff, mz, occ, val, na, nmini, rep, aoff, minioff, a, tmp, mini, u, b, nu, nb, ws;
			mga_dbuf_t tseq, prob, res, pool, used, ncig, cigoff, ord, rflag, item, chain, vert, txtres, txtpool;
			mga_dbuf_t sk_item, sk_cnt, sk_off, sd_tk, sd_kf, sd_offa, sd_offm, sd_rkey, sd_rmax; /* long-query path (MG_M_RMQ): sketch pieces, per-minimizer scans */
			mga_dbuf_t hash, gchdr, gcpool, lcpool, apool, gcctl, gcretry; /* graph chaining on the device (k_gchain.hip) */
			mga_dbuf_t plcnt, ploff, pltot, plsrc, plrev; /* gap list on the device (k_plan.hip) */
			mga_dbuf_t lcord; /* k_lchain's launch order: reads by anchor count, most first */
			mga_dbuf_t rq_a, rq_f, rq_p, rq_v, rq_t, rq_pri, rq_ys, rq_cut, rq_ord, rq_stat, rq_cnt; /* forward pass of the RMQ chainer on the device (k_rmq.hip), one read at a time */
			mga_dbuf_t g_line, g_qn, g_qoff, g_len, g_off, g_out; /* whole GAF lines on the device (k_gaf.hip): line records, read names, line lengths / offsets, the text */
		};
		mga_dbuf_t dall[73];
	};
	union {
		struct { mga_hbuf_t h_b, h_u, h_mini, h_tseq, h_prob, h_pool, h_seq, h_ncig, h_cigoff, h_item, h_chain, h_vert, h_txtres, h_txtpool, h_gchdr, h_gcpool, h_lcpool, h_apool, h_plrev, h_ploff, h_lcord, h_rq, h_rqs, h_plsrc, h_gline, h_gqn, h_gout; }; /* pinned staging */
		mga_hbuf_t hall[27];
	};
} pipe_ctx_t;
_Static_assert(sizeof(((pipe_ctx_t*)0)->dall) == 73 * sizeof(mga_dbuf_t) && sizeof(((pipe_ctx_t*)0)->hall) == 27 * sizeof(mga_hbuf_t), "pipe_ctx_t: buffer lists out of sync");

#define MGA_MAX_PIPE 8

static double g_job_t0;
static int g_dbg_pipe = -1;
#define PIPE_LOG(what, c, tb) do { if (g_dbg_pipe > 0) fprintf(stderr, "[pipe] chunk %d %-10s %8.1f .. %8.1f ms\n", (c), (what), ((tb) - g_job_t0) * 1e3, (mga_wtime() - g_job_t0) * 1e3); } while (0)

/* Two pipeline threads that start together would run every stage in lockstep (both on the GPU, then both on the host).
 * One token per GPU phase staggers them: while one chunk fills gaps on the GPU, the other one's host stages run, and the
 * cheap front phase (sketch/seed/chain) of one chunk fills the tail of another chunk's WFA launches. */
typedef struct gpu_token_s { pthread_mutex_t m; pthread_cond_t c; int avail; } gpu_token_t;
static gpu_token_t g_gpu_front = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 1 }, g_gpu_wfa = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 1 };
static void token_acquire(gpu_token_t *t) { pthread_mutex_lock(&t->m); while (t->avail <= 0) pthread_cond_wait(&t->c, &t->m); --t->avail; pthread_mutex_unlock(&t->m); }
static void token_release(gpu_token_t *t) { pthread_mutex_lock(&t->m); ++t->avail; pthread_cond_signal(&t->c); pthread_mutex_unlock(&t->m); }

static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s && *s ? atoi(s) : dflt; }
/* reads of at least this many bases are "ultra-long" -x lr reads (chunks of their own, first chaining pass on host threads): read ONCE per process (the cut and
 * the chunk's placement must agree, whatever happens to the environment in between) */
static int lr_long_bases(void) { static int v = -1; if (v < 0) v = env_int("MGA_LONG_READ", 262144); return v; }

#define CK(x) do { if ((x) < 0) { rc = -1; goto done; } } while (0)

/* mga_batch_t's device hook: the forward pass of the RMQ chainer over the runs `order[0 .. n_order)` of one read's x-sorted anchors (k_rmq.hip), on this chunk's stream.
 * Called by ONE host thread at a time (mga_batch_t::rq_dev_mtx) while the chunk's pipeline thread waits inside mga_batch_chain(); f, p, v arrive in the caller's arrays
 * (pinned: the chunk's h_rq), status[run] says which runs the host has to redo. */
typedef struct { pipe_ctx_t *P; const mg_mapopt_t *opt; float pen_gap, pen_skip; int64_t n_total; } rq_dev_ctx_t;
static int rq_dev_fwd_hook(void *ctx_, int n_reads, const int64_t *r_abs0, const int64_t *r_n, const mg128_t *const *r_a, int32_t *const *r_f, int64_t *const *r_p, int32_t *const *r_v,
						   int n_runs, const int64_t *runs, int n_order, const int32_t *order, int bw, int32_t *status)
{
	rq_dev_ctx_t *C = (rq_dev_ctx_t*)ctx_;
	pipe_ctx_t *P = C->P;
	mga_sctx_t *sc = P->sc;
	const mg_mapopt_t *opt = C->opt;
	const int64_t n = C->n_total; /* the chunk's anchors: the device arrays are chunk-level, a read's slice sits at its offset */
	char *hs;
	int x;
	if (mga_dev_bind_thread() < 0) return -1;
	if (mga_dbuf_reserve(&P->rq_a, (size_t)n * 16 + 64) < 0 || mga_dbuf_reserve(&P->rq_f, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&P->rq_p, (size_t)n * 8 + 64) < 0 ||
		mga_dbuf_reserve(&P->rq_v, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&P->rq_t, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&P->rq_pri, (size_t)n * 8 + 64) < 0 ||
		mga_dbuf_reserve(&P->rq_ys, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&P->rq_cut, (size_t)n_runs * 24 + 64) < 0 || mga_dbuf_reserve(&P->rq_ord, (size_t)n_runs * 4 + 64) < 0 ||
		mga_dbuf_reserve(&P->rq_stat, (size_t)n_runs * 4 + 64) < 0 || mga_dbuf_reserve(&P->rq_cnt, 64) < 0 || mga_hbuf_reserve(&P->h_rqs, (size_t)n_runs * 32 + 64) < 0) return -1;
	hs = (char*)P->h_rqs.p; /* runs | order | status through pinned staging: copies from pageable memory wait inside the runtime */
	memcpy(hs, runs, (size_t)n_runs * 24); memcpy(hs + (size_t)n_runs * 24, order, (size_t)n_order * 4); memcpy(hs + (size_t)n_runs * 28, status, (size_t)n_runs * 4);
	for (x = 0; x < n_reads; ++x)
		if (mga_h2d_s(sc, (mg128_t*)P->rq_a.p + r_abs0[x], r_a[x], (size_t)r_n[x] * 16) < 0 || mga_dmemset_s(sc, (int32_t*)P->rq_t.p + r_abs0[x], 0, (size_t)r_n[x] * 4) < 0) return -1;
	if (mga_h2d_s(sc, P->rq_cut.p, hs, (size_t)n_runs * 24) < 0 || mga_h2d_s(sc, P->rq_ord.p, hs + (size_t)n_runs * 24, (size_t)n_order * 4) < 0 ||
		mga_h2d_s(sc, P->rq_stat.p, hs + (size_t)n_runs * 28, (size_t)n_runs * 4) < 0) return -1;
	if (mga_dev_rmq_fwd(sc, n, (const mg128_t*)P->rq_a.p, n_order, P->rq_cut.p, (const int32_t*)P->rq_ord.p, opt->max_gap, opt->max_gap_pre, bw, opt->max_lc_skip, opt->rmq_size_cap,
						C->pen_gap, C->pen_skip, (int32_t*)P->rq_f.p, (int64_t*)P->rq_p.p, (int32_t*)P->rq_v.p, (int32_t*)P->rq_t.p, (double*)P->rq_pri.p, (int32_t*)P->rq_ys.p,
						(int32_t*)P->rq_stat.p, (int*)P->rq_cnt.p) < 0) return -1;
	for (x = 0; x < n_reads; ++x)
		if (mga_d2h_s(sc, r_f[x], (int32_t*)P->rq_f.p + r_abs0[x], (size_t)r_n[x] * 4) < 0 || mga_d2h_s(sc, r_p[x], (int64_t*)P->rq_p.p + r_abs0[x], (size_t)r_n[x] * 8) < 0 ||
			mga_d2h_s(sc, r_v[x], (int32_t*)P->rq_v.p + r_abs0[x], (size_t)r_n[x] * 4) < 0) return -1;
	if (mga_d2h_s(sc, hs + (size_t)n_runs * 28, P->rq_stat.p, (size_t)n_runs * 4) < 0 || mga_ssync(sc) < 0) return -1;
	memcpy(status, hs + (size_t)n_runs * 28, (size_t)n_runs * 4);
	return 0;
}

static void release_token_cb(void *a) { gpu_token_t **held = (gpu_token_t**)a; if (*held) { token_release(*held); *held = 0; } }

/* ---- GAF lines on the device (k_gaf.hip) ----
 * One record per line the reference's writer prints for the chunk (format.c:121-250), in print order: the printed chains of every read (in text mode exactly the chains with a
 * text-kernel chain: chain_id >= 0), or the one line of an unmapped read under MG_M_SHOW_UNMAP.  Returns the number of lines; -1: a read of the chunk has chains but no plan
 * (it was not chained here) -- the host then formats the chunk as before. */
static int64_t gaf_lines_build(const mga_batch_t *b, int n, mga_gaf_line_t *line)
{
	int64_t nl = 0;
	int i, k;
	for (i = 0; i < n; ++i) {
		const mg_gchains_t *gcs = b->gcs[i];
		const read_plan_t *pl = &b->plan[i];
		if (gcs == 0 || gcs->n_gc == 0) {
			if (b->opt.flag & MG_M_SHOW_UNMAP) { mga_gaf_line_t *l = &line[nl++]; memset(l, 0, sizeof *l); l->read = i, l->chain = -1, l->qlen = b->qlens[i]; }
			continue;
		}
		if (pl->chain_id == 0) return -1;
		for (k = 0; k < gcs->n_gc; ++k) {
			const mg_gchain_t *p = &gcs->gc[k];
			mga_gaf_line_t *l;
			if (pl->chain_id[k] < 0) continue; /* not printed (format.c:135-136) */
			l = &line[nl++];
			l->read = i, l->qlen = b->qlens[i];
			l->chain = (int32_t)(pl->tid < 0 ? pl->chain_id[k] : b->dp_n_chain + b->tp_chain_base[pl->tid] + pl->chain_id[k]); /* device-planned chains first, then the pools' */
			l->qs = p->qs, l->qe = p->qe, l->plen = p->plen, l->ps = p->ps, l->pe = p->pe;
			l->mapq = (int32_t)p->mapq, l->n_anchor = p->n_anchor, l->score = p->score, l->subsc = p->subsc;
			l->primary = p->id == p->parent, l->div = p->div;
		}
	}
	return nl;
}

typedef struct { char *dst; const char *src; int64_t bytes; int T; } pcopy_t;
static void pcopy_worker(void *data, int64_t t, int tid) { pcopy_t *c = (pcopy_t*)data; const int64_t b = c->bytes * t / c->T, e = c->bytes * (t + 1) / c->T; (void)tid; if (e > b) memcpy(c->dst + b, c->src + b, (size_t)(e - b)); }

static int map_chunk(pipe_ctx_t *P, const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs_out,
					 const mg_mapopt_t *opt, int n_threads, const char *d_seq_res, const int64_t *q_off_res, int seqs_pinned, mga_stats_t *st, kstring_t *gaf_part, int lr_long, const gaf_sink_t *sink)
{
	struct mg_idx_bucket_s *B = gi->B;
	mga_sctx_t *sc = P->sc;
	int rc = 0, i;
	int64_t tot = 0, n_mz, n_a, n_mini, n_prob = 0, n_tb = 0, pool_cap;
	int64_t *q_off = MGA_MALLOC(int64_t, n + 2), *h_mzoff = 0, *h_aoff = 0, *h_minioff = 0;
	int32_t *h_nmz = 0, *h_rep = 0, *h_nu = 0, *h_nb = 0, *h_rflag = 0;
	mga_gc_hdr_t *h_gchdr_p = 0;
	int64_t gc_cap = 0, lc_cap = 0, ga_cap = 0;
	int dev_gc = 0, dev_plan = 0, need_a = 1;
	unsigned long long ptot[8] = { 0 }; /* device-made gap list: printed chains, plan items, problems, walk vertices, target bytes, query bases, overflow flag */
	int64_t *h_ploff = 0;               /* its first printed chain per read */
	const int want_text = gaf_part != 0 && (opt->flag & MG_M_CIGAR) && B->dev.d_gseq != 0 && !env_int("MGA_HOST_TEXT", 0); /* only GAF bytes are wanted: cg/ds come from the device */
	mga_batch_t *b = 0;
	mga_lchain_par_t par;
	const int is_rmq = !!(opt->flag & MG_M_RMQ);
	int chunk_long = 0; /* -x lr, a chunk of ultra-long reads (batch_cut): the long-query path as well, first chaining pass on host threads (hchain.c: mga_lchain_dp_fwd) */
	if (!is_rmq && n > 0 && (gi->k & 1) && !env_int("MGA_NO_LONGQ", 0)) { chunk_long = lr_long > 0; for (i = 0; i < n && chunk_long; ++i) if (qlens[i] < lr_long) chunk_long = 0; }
	const int long_q = (is_rmq || chunk_long) && (gi->k & 1) && !env_int("MGA_NO_LONGQ", 0); /* few, very long queries -- intra-query parallel sketch and seed expansion, anchors sorted by the host chainer */
	const char *d_seq;
	double t0, t1;
	gpu_token_t *held = 0; /* GPU phase token currently owned */
#define GPU_ACQUIRE(m) do { token_acquire(m); held = (m); } while (0)
#define GPU_RELEASE() do { if (held) { token_release(held); held = 0; } } while (0)

	if (opt->flag & (MG_M_SR | MG_M_HEAP_SORT | MG_M_SPLICE | MG_M_NO_DIAG)) {
		mga_set_error("mg_map_batch: short-read / splice / -D modes are outside the accelerated long-read path"); rc = -1; goto done;
	}
	/* ---- reads -> HBM, back to back, 64 readable bytes of padding at the end (8-byte compares in k_wfa);
	 *      skipped when the caller keeps the batch resident (d_seq_res + absolute offsets q_off_res) ---- */
	t0 = mga_wtime();
	if (d_seq_res) {
		memcpy(q_off, q_off_res, (size_t)(n + 1) * 8);
		tot = q_off[n] - q_off[0];
		GPU_ACQUIRE(P->tok_front ? P->tok_front : &g_gpu_front);
		d_seq = d_seq_res;
	} else {
		const char *h_seq;
		for (i = 0; i < n; ++i) { q_off[i] = tot; tot += qlens[i]; }
		q_off[n] = tot;
		if (seqs_pinned) h_seq = seqs[0]; /* the reader parsed the batch into pinned memory, reads back to back: no staging copy */
		else { /* staged before the GPU phase token is taken: the copy is host work */
			char *h;
			CK(mga_hbuf_reserve(&P->h_seq, (size_t)tot + 64));
			h = (char*)P->h_seq.p;
			for (i = 0; i < n; ++i) memcpy(h + q_off[i], seqs[i], (size_t)qlens[i]);
			memset(h + tot, 0, 64);
			h_seq = h;
		}
		GPU_ACQUIRE(P->tok_front ? P->tok_front : &g_gpu_front);
		CK(mga_dbuf_reserve(&P->seq, (size_t)tot + 64));
		CK(mga_h2d_s(sc, P->seq.p, h_seq, (size_t)tot + 64));
		d_seq = (const char*)P->seq.p;
	}
	CK(mga_dbuf_reserve(&P->qoff, (size_t)(n + 1) * 8));
	CK(mga_h2d_s(sc, P->qoff.p, q_off, (size_t)(n + 1) * 8));
	/* ---- sketch: ONE pass into per-read slots of qlen/2 + 64 minimizers (the density is 2/(w+1), ~3x less); a read that would
	 *      overflow its slots (never seen) sends the chunk through count + scan + write ---- */
	CK(mga_dbuf_reserve(&P->cnt, (size_t)n * 4 + 4)); CK(mga_dbuf_reserve(&P->mzoff, (size_t)(n + 1) * 8));
	h_mzoff = MGA_MALLOC(int64_t, n + 1);
	h_nmz = MGA_MALLOC(int32_t, n);
	if (long_q) { /* contigs of megabases: pieces of 64 kb are the work items (k_sketch.hip), count + scan + write, exact contiguous offsets */
		const int32_t PIECE = 1 << 16;
		int64_t n_items = 0, it = 0, *h_itoff;
		int32_t *h_item;
		for (i = 0; i < n; ++i) n_items += qlens[i] > 0 ? (qlens[i] + PIECE - 1) / PIECE : 1;
		h_item = MGA_MALLOC(int32_t, n_items * 4);
		h_itoff = MGA_MALLOC(int64_t, n_items + 1);
		for (i = 0; i < n; ++i) {
			int32_t beg = 0;
			do { h_item[it * 4] = i, h_item[it * 4 + 1] = beg, h_item[it * 4 + 2] = beg + PIECE < qlens[i] ? beg + PIECE : qlens[i], h_item[it * 4 + 3] = 0; ++it, beg += PIECE; } while (beg < qlens[i]);
		}
		rc = mga_dbuf_reserve(&P->sk_item, (size_t)n_items * 16) < 0 || mga_dbuf_reserve(&P->sk_cnt, (size_t)n_items * 4 + 4) < 0 || mga_dbuf_reserve(&P->sk_off, (size_t)(n_items + 1) * 8) < 0
			|| mga_h2d_s(sc, P->sk_item.p, h_item, (size_t)n_items * 16) < 0
			|| mga_dev_sketch_items(sc, (int)n_items, (const int32_t*)P->sk_item.p, d_seq, (const int64_t*)P->qoff.p, 0, gi->w, gi->k, (int32_t*)P->sk_cnt.p, 0, 0) < 0
			|| mga_dev_scan_i32_to_i64(sc, (const int32_t*)P->sk_cnt.p, n_items, (int64_t*)P->sk_off.p) < 0
			|| mga_ssync(sc) < 0 || mga_d2h(h_itoff, P->sk_off.p, (size_t)(n_items + 1) * 8) < 0 ? -1 : 0; /* (h_item must stay alive until the copy has been consumed) */
		if (rc == 0) {
			for (i = 0, it = 0; i < n; ++i) { h_mzoff[i] = h_itoff[it]; it += qlens[i] > 0 ? (qlens[i] + PIECE - 1) / PIECE : 1; }
			h_mzoff[n] = n_mz = h_itoff[n_items];
			for (i = 0; i < n; ++i) {
				if (h_mzoff[i + 1] - h_mzoff[i] > 0x7fffffff) { mga_set_error("read %d has more than 2^31 minimizers", i); rc = -1; break; }
				h_nmz[i] = (int32_t)(h_mzoff[i + 1] - h_mzoff[i]);
			}
		}
		free(h_item); free(h_itoff);
		if (rc < 0) goto done;
		CK(mga_h2d_s(sc, P->mzoff.p, h_mzoff, (size_t)(n + 1) * 8)); CK(mga_h2d_s(sc, P->cnt.p, h_nmz, (size_t)n * 4));
		CK(mga_dbuf_reserve(&P->mz, (size_t)n_mz * 16 + 16));
		CK(mga_dev_sketch_items(sc, (int)n_items, (const int32_t*)P->sk_item.p, d_seq, (const int64_t*)P->qoff.p, 0, gi->w, gi->k, 0, (const int64_t*)P->sk_off.p, (mg128_t*)P->mz.p));
		CK(mga_ssync(sc)); /* h_mzoff / h_nmz uploads done */
	} else {
		int overflow = 0;
		for (i = 0, n_mz = 0; i < n; ++i) { h_mzoff[i] = n_mz; n_mz += qlens[i] / 2 + 64; }
		h_mzoff[n] = n_mz; /* capacity of the chunk */
		CK(mga_h2d_s(sc, P->mzoff.p, h_mzoff, (size_t)(n + 1) * 8));
		CK(mga_dbuf_reserve(&P->mz, (size_t)n_mz * 16 + 16));
		if (env_int("MGA_SKETCH_2BIT", 0)) CK(mga_dev_pack2(sc, d_seq, q_off[n])); /* A/B: the reads as bit planes for the sketch (k_sketch.hip; DESIGN.md has the measurement) */
		CK(mga_dev_sketch(sc, n, d_seq, (const int64_t*)P->qoff.p, 0, gi->w, gi->k, (int32_t*)P->cnt.p, (const int64_t*)P->mzoff.p, (mg128_t*)P->mz.p));
		CK(mga_d2h_s(sc, h_nmz, P->cnt.p, (size_t)n * 4)); CK(mga_ssync(sc));
		for (i = 0; i < n; ++i) if (h_nmz[i] > qlens[i] / 2 + 64) overflow = 1;
		if (overflow) {
			CK(mga_dev_scan_i32_to_i64(sc, (const int32_t*)P->cnt.p, n, (int64_t*)P->mzoff.p));
			CK(mga_d2h_s(sc, h_mzoff, P->mzoff.p, (size_t)(n + 1) * 8)); CK(mga_ssync(sc));
			n_mz = h_mzoff[n];
			CK(mga_dbuf_reserve(&P->mz, (size_t)n_mz * 16 + 16));
			CK(mga_dev_sketch(sc, n, d_seq, (const int64_t*)P->qoff.p, 0, gi->w, gi->k, 0, (const int64_t*)P->mzoff.p, (mg128_t*)P->mz.p));
		}
	}
	/* ---- seeds ---- */
	CK(mga_dbuf_reserve(&P->occ, (size_t)n_mz * 4 + 4)); CK(mga_dbuf_reserve(&P->val, (size_t)n_mz * 8 + 8));
	CK(mga_dbuf_reserve(&P->na, (size_t)n * 4 + 4)); CK(mga_dbuf_reserve(&P->nmini, (size_t)n * 4 + 4)); CK(mga_dbuf_reserve(&P->rep, (size_t)n * 4 + 4));
	CK(mga_dbuf_reserve(&P->aoff, (size_t)(n + 1) * 8)); CK(mga_dbuf_reserve(&P->minioff, (size_t)(n + 1) * 8));
	if (long_q) {
		CK(mga_dbuf_reserve(&P->sd_tk, (size_t)n_mz * 4 + 4)); CK(mga_dbuf_reserve(&P->sd_kf, (size_t)n_mz * 4 + 4));
		CK(mga_dbuf_reserve(&P->sd_offa, (size_t)(n_mz + 1) * 8)); CK(mga_dbuf_reserve(&P->sd_offm, (size_t)(n_mz + 1) * 8));
		CK(mga_dbuf_reserve(&P->sd_rkey, (size_t)n_mz * 8 + 8)); CK(mga_dbuf_reserve(&P->sd_rmax, (size_t)n_mz * 8 + 8));
		CK(mga_dev_seed_long_count(sc, &B->dev, n, (const mg128_t*)P->mz.p, (const int64_t*)P->mzoff.p, n_mz, opt->occ_max1, (int32_t*)P->occ.p, (uint64_t*)P->val.p,
								   (int32_t*)P->sd_tk.p, (int32_t*)P->sd_kf.p, (int64_t*)P->sd_offa.p, (int64_t*)P->sd_offm.p, (uint64_t*)P->sd_rkey.p, (uint64_t*)P->sd_rmax.p,
								   (int64_t*)P->aoff.p, (int64_t*)P->minioff.p, (int32_t*)P->rep.p));
	} else {
		CK(mga_dev_seed_count(sc, &B->dev, n, (const mg128_t*)P->mz.p, (const int64_t*)P->mzoff.p, (const int32_t*)P->cnt.p, opt->occ_max1, (int32_t*)P->occ.p, (uint64_t*)P->val.p,
						  (int32_t*)P->na.p, (int32_t*)P->nmini.p, (int32_t*)P->rep.p));
		CK(mga_dev_scan_i32_to_i64(sc, (const int32_t*)P->na.p, n, (int64_t*)P->aoff.p));
		CK(mga_dev_scan_i32_to_i64(sc, (const int32_t*)P->nmini.p, n, (int64_t*)P->minioff.p));
	}
	h_aoff = MGA_MALLOC(int64_t, n + 1); h_minioff = MGA_MALLOC(int64_t, n + 1); h_rep = MGA_MALLOC(int32_t, n);
	CK(mga_d2h_s(sc, h_aoff, P->aoff.p, (size_t)(n + 1) * 8)); CK(mga_d2h_s(sc, h_minioff, P->minioff.p, (size_t)(n + 1) * 8)); CK(mga_d2h_s(sc, h_rep, P->rep.p, (size_t)n * 4));
	CK(mga_ssync(sc));
	if (g_dbg_pipe > 1) PIPE_LOG(" sketch", n, t0);
	t1 = mga_wtime(); st->t_sketch += t1 - t0; t0 = t1;
	n_a = h_aoff[n], n_mini = h_minioff[n];
	CK(mga_dbuf_reserve(&P->a, (size_t)n_a * 16 + 64)); CK(mga_dbuf_reserve(&P->mini, (size_t)n_mini * 4 + 16));
	if (long_q) {
		CK(mga_dev_seed_long_fill(sc, &B->dev, n, (const mg128_t*)P->mz.p, (const int64_t*)P->mzoff.p, n_mz, opt->occ_max1, (const int32_t*)P->occ.p, (const uint64_t*)P->val.p,
								  (const int64_t*)P->sd_offa.p, (const int64_t*)P->sd_offm.p, (mg128_t*)P->a.p, (int32_t*)P->mini.p));
	} else {
		CK(mga_dbuf_reserve(&P->tmp, (size_t)n_a * 16 + 64));
		CK(mga_dev_seed_fill(sc, &B->dev, n, (const mg128_t*)P->mz.p, (const int64_t*)P->mzoff.p, (const int32_t*)P->cnt.p, opt->occ_max1, (const int32_t*)P->occ.p, (const uint64_t*)P->val.p,
							 (const int64_t*)P->aoff.p, (mg128_t*)P->a.p, (const int64_t*)P->minioff.p, (int32_t*)P->mini.p, (mg128_t*)P->tmp.p));
	}
	CK(mga_hbuf_reserve(&P->h_mini, (size_t)n_mini * 4 + 16));
	/* ---- linear chaining ---- */
	CK(mga_hbuf_reserve(&P->h_b, (size_t)n_a * 16 + 16));
	if (!is_rmq && !chunk_long) {
		size_t wsb = mga_dev_lchain_ws_bytes(n_a);
		mga_rescue_par_t rs;
		mga_batch_lchain_par(gi, opt, 0, &par);
		CK(mga_dbuf_reserve(&P->u, (size_t)n_a * 8 + 8)); CK(mga_dbuf_reserve(&P->b, (size_t)n_a * 16 + 16));
		CK(mga_dbuf_reserve(&P->nu, (size_t)n * 4 + 4)); CK(mga_dbuf_reserve(&P->nb, (size_t)n * 4 + 4)); CK(mga_dbuf_reserve(&P->ws, wsb));
		/* the long-join rescue (map-algo.c:407-417) runs inside the same kernel; MGA_HOST_RESCUE=1 keeps it on the host (A/B testing) */
		memset(&rs, 0, sizeof rs);
		rs.enabled = opt->bw_long > opt->bw && (opt->flag & (MG_M_SPLICE | MG_M_SR)) == 0 && !env_int("MGA_HOST_RESCUE", 0);
		rs.max_dist = opt->max_gap, rs.max_dist_inner = opt->max_gap_pre, rs.bw = opt->bw_long, rs.max_skip = opt->max_lc_skip, rs.cap = opt->rmq_size_cap;
		rs.min_cnt = opt->min_lc_cnt, rs.min_sc = opt->min_lc_score, rs.chn_pen_gap = par.chn_pen_gap, rs.chn_pen_skip = par.chn_pen_skip;
		rs.rescue_size = opt->rmq_rescue_size, rs.rescue_ratio = opt->rmq_rescue_ratio;
		if (opt->max_gap_ref <= 0 && opt->max_frag_len > 0) rs.frag_len = opt->max_frag_len, rs.frag_min_gap = opt->max_gap; /* -F */
		CK(mga_dbuf_reserve(&P->rflag, (size_t)n * 4 + 4));
		if (env_int("MGA_LC_ORDER", 1)) { /* the launch lasts as long as its longest read: the reads with the most anchors are launched first (a counting sort of the anchor counts, 16 per bin) */
			enum { NBIN = 4096 };
			/* (the order is built in the context's pinned staging and the copy is only ENQUEUED -- the stream puts it before k_lchain; two mallocs, a pageable
			 * copy and a full synchronisation per chunk stood here while the chunk held the front-phase token.  The buffer is next written by this context's next chunk.) */
			int32_t *ord, cnt_[NBIN + 1];
			CK(mga_hbuf_reserve(&P->h_lcord, (size_t)n * 4 + 16)); CK(mga_dbuf_reserve(&P->lcord, (size_t)n * 4 + 4));
			ord = (int32_t*)P->h_lcord.p;
			memset(cnt_, 0, sizeof cnt_);
			for (i = 0; i < n; ++i) { const int64_t na_ = (h_aoff[i + 1] - h_aoff[i]) >> 4; ++cnt_[NBIN - 1 - (na_ < NBIN ? na_ : NBIN - 1) + 1]; }
			for (i = 0; i < NBIN; ++i) cnt_[i + 1] += cnt_[i];
			for (i = 0; i < n; ++i) { const int64_t na_ = (h_aoff[i + 1] - h_aoff[i]) >> 4; ord[cnt_[NBIN - 1 - (na_ < NBIN ? na_ : NBIN - 1)]++] = (int32_t)i; }
			CK(mga_h2d_s(sc, P->lcord.p, ord, (size_t)n * 4));
			mga_dev_lchain_order(sc, (const int32_t*)P->lcord.p);
		}
		CK(mga_dev_lchain(sc, n, (const mg128_t*)P->a.p, (const int64_t*)P->aoff.p, &par, &rs, (const int64_t*)P->qoff.p, (uint64_t*)P->u.p, (mg128_t*)P->b.p,
						  (int32_t*)P->nu.p, (int32_t*)P->nb.p, (int32_t*)P->rflag.p, P->ws.p, wsb, n_a));
		h_nu = MGA_MALLOC(int32_t, n); h_nb = MGA_MALLOC(int32_t, n);
		CK(mga_hbuf_reserve(&P->h_u, (size_t)n_a * 8 + 8));
		CK(mga_d2h_s(sc, h_nu, P->nu.p, (size_t)n * 4)); CK(mga_d2h_s(sc, h_nb, P->nb.p, (size_t)n * 4));
		if (rs.enabled) { /* otherwise the host evaluates the rescue condition itself */
			h_rflag = MGA_MALLOC(int32_t, n);
			CK(mga_d2h_s(sc, h_rflag, P->rflag.p, (size_t)n * 4));
		}
		/* graph chaining (gc_core.h) runs where it fits: on the device, one wavefront per read (k_gchain.hip, [measured] ~0.13 s of GPU time per 100k reads),
		 * when this GPU has few host threads to itself -- a node whose CPU quota does not grow with its GPUs -- and on the host threads
		 * ([measured] ~3 CPU-s per 100k reads on a 3 Gbp graph) when there are enough of them to keep up.  [measured, cores = threads] 8: 1.42 (host) vs 1.78 Gbp/s
		 * (device); 10: 1.59 vs 1.83; 16: 2.00 vs 1.84.  MGA_DEV_GCHAIN=1 / 0 forces one or the other; same bytes either way. */
		dev_gc = rs.enabled && B->dev.d_arc != 0 && env_int("MGA_DEV_GCHAIN", n_threads <= 12) && !env_int("MGA_HOST_GCHAIN", 0);
		/* Round 6: with many host threads neither side is idle -- the host threads chain the graph for 2.3-2.7 CPU-s per 125 000 reads while the GPU runs 250 ms of kernels --
		 * so a SHARE of the chunks takes the device placement (k_gchain + k_plan fill the chip's idle wave slots at 2 waves per SIMD) and the rest the host's: both placements
		 * give the same bytes (every e2e test runs them), a chunk is a unit.  MGA_DEV_GCHAIN_PCT = percentage of chunks on the device when MGA_DEV_GCHAIN is not set
		 * (default 25: [measured, profiles/r06y_hybrid.txt] the same throughput as 0 on a box whose 16 cores keep up -- 4.40 vs 4.42 Gbp/s -- at 2.9 instead of 4.0 CPU-s per step;
		 * the final bench run of this round landed on a box where they do not: 3.72 Gbp/s at 4.95 CPU-s with all chunks on the host, 4.05 with all on the device);
		 * MGA_DEV_GCHAIN=0 / 1 forces one placement as before. */
		if (rs.enabled && B->dev.d_arc != 0 && getenv("MGA_DEV_GCHAIN") == 0 && !env_int("MGA_HOST_GCHAIN", 0) && n_threads > 12) {
			static int g_gc_turn = 0;
			const int pct = env_int("MGA_DEV_GCHAIN_PCT", 25), turn = __sync_fetch_and_add(&g_gc_turn, 1);
			dev_gc = pct > 0 && ((int64_t)(turn + 1) * pct) / 100 > ((int64_t)turn * pct) / 100;
		}
		if (dev_gc) {
			/* ---- graph chaining: chain records, clean-up, DP + shortest walks, GWFA bridging, ordering, filters -- one wavefront per read ---- */
			const size_t rec = mga_gc_rec_bytes();
			uint32_t *h_hash = MGA_MALLOC(uint32_t, n);
			unsigned long long ctl[9]; /* [8]: GWFA calls / graph searches that ran in the LDS scratch */
			int attempt;
			for (i = 0; i < n; ++i) h_hash[i] = mga_read_hash(qnames ? qnames[i] : 0, qlens[i], opt->seed);
			gc_cap = n_a / (opt->min_lc_cnt > 0 ? opt->min_lc_cnt : 1) + n + 64, lc_cap = gc_cap * 3 + 4096, ga_cap = n_a + 64;
			rc = mga_dbuf_reserve(&P->hash, (size_t)n * 4 + 4) < 0 || mga_h2d_s(sc, P->hash.p, h_hash, (size_t)n * 4) < 0 || mga_ssync(sc) < 0 ? -1 : 0;
			free(h_hash);
			if (rc < 0) goto done;
			CK(mga_dbuf_reserve(&P->gchdr, (size_t)n * sizeof(mga_gc_hdr_t) + 64)); CK(mga_dbuf_reserve(&P->gcctl, 256)); CK(mga_dbuf_reserve(&P->gcretry, (size_t)n * 8 + 8)); /* (two lists: the first launch's retries, the retry launch's own) */
			CK(mga_hbuf_reserve(&P->h_gchdr, (size_t)n * sizeof(mga_gc_hdr_t) + 64)); /* (large read-backs go to pinned memory) */
			for (attempt = 0;; ++attempt) {
				int64_t n_retry;
				CK(mga_dbuf_reserve(&P->gcpool, (size_t)gc_cap * rec)); CK(mga_dbuf_reserve(&P->lcpool, (size_t)lc_cap * sizeof(mg_llchain_t))); CK(mga_dbuf_reserve(&P->apool, (size_t)ga_cap * 16));
				CK(mga_dmemset_s(sc, P->gcctl.p, 0, 256));
				if (env_int("MGA_GC_PROF", 0)) { unsigned long long one = 1; CK(mga_h2d_s(sc, (char*)P->gcctl.p + 15 * 8, &one, 8)); CK(mga_ssync(sc)); }
				CK(mga_dev_gchain(sc, &B->dev, opt, gi->k, par.chn_pen_gap, n, 0, 0, (const int64_t*)P->aoff.p, (const int32_t*)P->nu.p, (const int32_t*)P->nb.p, (const uint64_t*)P->u.p,
								  (const mg128_t*)P->b.p, (const int64_t*)P->minioff.p, (const int32_t*)P->mini.p, (const int64_t*)P->qoff.p, d_seq, (const uint32_t*)P->hash.p,
								  (const int32_t*)P->rflag.p, (mga_gc_hdr_t*)P->gchdr.p, P->gcpool.p, gc_cap, (mg_llchain_t*)P->lcpool.p, lc_cap, (mg128_t*)P->apool.p, ga_cap,
								  (unsigned long long*)P->gcctl.p, (int32_t*)P->gcretry.p));
				CK(mga_d2h_s(sc, ctl, P->gcctl.p, 72)); CK(mga_ssync(sc));
				n_retry = (int64_t)ctl[3];
				if ((int64_t)ctl[1] > gc_cap || (int64_t)ctl[2] > lc_cap || (int64_t)ctl[6] > ga_cap) { /* a record pool was too small: size it to what the kernel asked for and run the chunk again */
					if (attempt >= 2) { mga_set_error("graph chaining: record pools keep overflowing (%lld/%lld/%lld)", (long long)ctl[1], (long long)ctl[2], (long long)ctl[6]); rc = -1; goto done; }
					if ((int64_t)ctl[1] > gc_cap) gc_cap = (int64_t)ctl[1] * 2;
					if ((int64_t)ctl[2] > lc_cap) lc_cap = (int64_t)ctl[2] * 2;
					if ((int64_t)ctl[6] > ga_cap) ga_cap = (int64_t)ctl[6] * 2;
					continue;
				}
				if (n_retry > 0) { /* reads that outgrew the 1 MiB arena: once more with the large arenas, records appended to the same pools */
					unsigned long long zero2[1] = { 0 };
					CK(mga_h2d_s(sc, (char*)P->gcctl.p, zero2, 8)); CK(mga_h2d_s(sc, (char*)P->gcctl.p + 24, zero2, 8)); CK(mga_ssync(sc));
					CK(mga_dev_gchain(sc, &B->dev, opt, gi->k, par.chn_pen_gap, (int)n_retry, (const int32_t*)P->gcretry.p, 1, (const int64_t*)P->aoff.p, (const int32_t*)P->nu.p, (const int32_t*)P->nb.p,
									  (const uint64_t*)P->u.p, (const mg128_t*)P->b.p, (const int64_t*)P->minioff.p, (const int32_t*)P->mini.p, (const int64_t*)P->qoff.p, d_seq,
									  (const uint32_t*)P->hash.p, (const int32_t*)P->rflag.p, (mga_gc_hdr_t*)P->gchdr.p, P->gcpool.p, gc_cap, (mg_llchain_t*)P->lcpool.p, lc_cap,
									  (mg128_t*)P->apool.p, ga_cap, (unsigned long long*)P->gcctl.p, (int32_t*)P->gcretry.p + n)); /* (its own list for what fails again: the input list is still being consumed) */
					CK(mga_d2h_s(sc, ctl, P->gcctl.p, 72)); CK(mga_ssync(sc));
					/* A read that fails again -- more scratch than the large arena, or the record pools ran out during the retry -- is left with a non-zero status
					 * and no records: the host threads chain it, like the reads whose rescue k_lchain deferred (this used to fail the whole job). */
					if ((int64_t)ctl[1] > gc_cap) ctl[1] = (unsigned long long)gc_cap; /* (the pool counters ran past the pools: only what fits was written) */
					if ((int64_t)ctl[2] > lc_cap) ctl[2] = (unsigned long long)lc_cap;
					if ((int64_t)ctl[6] > ga_cap) ctl[6] = (unsigned long long)ga_cap;
					st->n_gc_retry += n_retry;
					{ /* reads the retry launch gave up on as well -- beyond the large arena, or the record pools ran out while it ran -- are chained by the host threads: counted, and said once */
						const long long n_again = (long long)ctl[3];
						if (n_again > 0) {
							static int warned = 0;
							st->n_gc_host += n_again;
							if (mg_verbose >= 2 && !warned) { warned = 1; fprintf(stderr, "[W::%s] graph chaining on the device: %lld of %lld retried reads go to the host threads (record pools %lld/%lld %lld/%lld %lld/%lld)\n", __func__, n_again, (long long)n_retry, (long long)ctl[1], (long long)gc_cap, (long long)ctl[2], (long long)lc_cap, (long long)ctl[6], (long long)ga_cap); }
						}
					}
				}
				break;
			}
			st->n_gwfa += (int64_t)ctl[4], st->n_shortk += (int64_t)ctl[5];
			if (env_int("MGA_GC_PROF", 0)) { /* per-stage cycle sums of this chunk (profiling aid) */
				unsigned long long tk[16];
				static const char *nm[16] = { "", "records", "cleanup", "index", "dp+shortk", "assemble(rest)", "post", "MAX-wave", "gwfa(rest)", "measure", "order", "gw:clear", "gw:runs", "gw:heads", "gw:dedup", "MAX-read" };
				int q_;
				CK(mga_d2h_s(sc, tk, (char*)P->gcctl.p + 128, 128)); CK(mga_ssync(sc));
				fprintf(stderr, "[gc-prof] %d reads, %llu GWFA calls (%llu in the LDS scratch), Mcycles:", n, ctl[4], ctl[8]);
				for (q_ = 1; q_ < 16; ++q_) if (nm[q_][0]) fprintf(stderr, " %s %.1f", nm[q_], tk[q_] * 1e-6);
				fprintf(stderr, "\n");
			}
			if ((int64_t)ctl[7] > st->gc_arena_peak) st->gc_arena_peak = (int64_t)ctl[7];
			dev_plan = want_text && env_int("MGA_DEV_PLAN", 1);
			if (dev_plan) { /* the gap list of the chains just made (k_plan.hip), pass 1: sizes; the fill pass follows the host's strand flags */
				CK(mga_dbuf_reserve(&P->plcnt, (size_t)n * 5 * 4 + 64)); CK(mga_dbuf_reserve(&P->ploff, (size_t)(n + 1) * 5 * 8 + 64)); CK(mga_dbuf_reserve(&P->pltot, 64));
				CK(mga_dev_plan_count(sc, &B->dev, n, (opt->flag & MG_M_PRINT_2ND) != 0, (const mga_gc_hdr_t*)P->gchdr.p, P->gcpool.p, (const mg_llchain_t*)P->lcpool.p,
									  (const mg128_t*)P->apool.p, (int32_t*)P->plcnt.p, (int64_t*)P->ploff.p, (unsigned long long*)P->pltot.p));
				CK(mga_hbuf_reserve(&P->h_ploff, (size_t)(n + 1) * 8 + 16));
				h_ploff = (int64_t*)P->h_ploff.p;
				CK(mga_d2h_s(sc, ptot, P->pltot.p, 64)); CK(mga_d2h_s(sc, h_ploff, P->ploff.p, (size_t)(n + 1) * 8));
			}
			CK(mga_hbuf_reserve(&P->h_gcpool, (size_t)ctl[1] * rec + 16)); CK(mga_hbuf_reserve(&P->h_lcpool, (size_t)ctl[2] * sizeof(mg_llchain_t) + 16)); CK(mga_hbuf_reserve(&P->h_apool, (size_t)ctl[6] * 16 + 16));
			h_gchdr_p = (mga_gc_hdr_t*)P->h_gchdr.p;
			CK(mga_d2h_s(sc, h_gchdr_p, P->gchdr.p, (size_t)n * sizeof(mga_gc_hdr_t)));
			CK(mga_d2h_s(sc, P->h_gcpool.p, P->gcpool.p, (size_t)ctl[1] * rec)); CK(mga_d2h_s(sc, P->h_lcpool.p, P->lcpool.p, (size_t)ctl[2] * sizeof(mg_llchain_t)));
			need_a = !dev_plan || (mg_dbg_flag & 0x8) || (opt->flag & MG_M_WRITE_LCHAIN); /* with the gap list made on the device, only the per-vertex lines (-S / --write-mz) read anchors on the host */
			if (need_a) CK(mga_d2h_s(sc, P->h_apool.p, P->apool.p, (size_t)ctl[6] * 16));
			CK(mga_ssync(sc)); /* (h_nu / h_nb / h_rflag arrived with the first sync) */
			for (i = 0; i < n; ++i) /* the few reads whose rescue k_lchain left to the host tree: their chains, anchors and minimizer positions come down for the host path */
				if (h_gchdr_p[i].status != 0) { /* (MGA_GC_HOST, or a read the device gave up on even in the large arena) */
					const int64_t ao = h_aoff[i], mo = h_minioff[i];
					CK(mga_d2h_s(sc, (uint64_t*)P->h_u.p + ao, (const uint64_t*)P->u.p + ao, (size_t)h_nu[i] * 8));
					CK(mga_d2h_s(sc, (mg128_t*)P->h_b.p + ao, (const mg128_t*)P->b.p + ao, (size_t)h_nb[i] * 16));
					CK(mga_d2h_s(sc, (int32_t*)P->h_mini.p + mo, (const int32_t*)P->mini.p + mo, (size_t)(h_minioff[i + 1] - mo) * 4));
					CK(mga_ssync(sc));
				}
		} else {
			CK(mga_d2h_s(sc, P->h_mini.p, P->mini.p, (size_t)n_mini * 4));
			CK(mga_d2h_s(sc, P->h_u.p, P->u.p, (size_t)n_a * 8)); CK(mga_d2h_s(sc, P->h_b.p, P->b.p, (size_t)n_a * 16));
		}
	} else {
		CK(mga_d2h_s(sc, P->h_mini.p, P->mini.p, (size_t)n_mini * 4));
		CK(mga_d2h_s(sc, P->h_b.p, P->a.p, (size_t)n_a * 16));
	}
	CK(mga_ssync(sc));
	GPU_RELEASE();
	if (g_dbg_pipe > 1) PIPE_LOG(" lchain", n, t0);
	t1 = mga_wtime(); st->t_lchain += t1 - t0; t0 = t1;
	if (env_int("MGA_CHECK", 0) && !is_rmq && !chunk_long && !dev_gc && h_nu && h_nb) { /* debugging aid: invariants of what the device stages handed back (host placement) */
		int64_t bad = 0;
		for (i = 0; i < n && bad < 5; ++i) {
			const int64_t na_i = h_aoff[i + 1] - h_aoff[i], nm_i = h_minioff[i + 1] - h_minioff[i];
			int64_t k_;
			int why = 0;
			if (na_i < 0 || nm_i < 0 || nm_i > qlens[i]) why = 1;
			else if (h_nb[i] < 0 || h_nb[i] > na_i || h_nu[i] < 0 || h_nu[i] > h_nb[i]) why = 2;
			else for (k_ = 0; k_ < nm_i; ++k_) { const int32_t y = ((const int32_t*)P->h_mini.p)[h_minioff[i] + k_]; if (y < 0 || y >= qlens[i] || (k_ > 0 && y <= ((const int32_t*)P->h_mini.p)[h_minioff[i] + k_ - 1])) { why = 3; break; } }
			if (why) { fprintf(stderr, "[check] chunk of %d reads: read %ld fails check %d: aoff %ld..%ld minioff %ld..%ld nu %d nb %d nmz %d qlen %d\n", n, (long)i, why, (long)h_aoff[i], (long)h_aoff[i + 1], (long)h_minioff[i], (long)h_minioff[i + 1], h_nu[i], h_nb[i], h_nmz[i], qlens[i]); ++bad; }
		}
		fprintf(stderr, "[check] chunk of %d reads: %ld bad (n_a %ld n_mini %ld n_mz capacity %ld)\n", n, (long)bad, (long)n_a, (long)n_mini, (long)n_mz);
	}
	/* ---- host: graph chaining + gap list ---- */
	b = mga_batch_init(gi, opt, n, qlens, seqs, qnames, q_off, n_threads);
	b->want_text = want_text;
	const int dev_splice = want_text && B->dev.d_gseq != 0 && B->dev.d_gseq_rc != 0 && env_int("MGA_DEV_SPLICE", 1); /* the gaps' targets are spliced on the device (align.h: want_src) */
	if (dev_splice) mga_batch_set_want_src(b, 1);
	if (dev_plan) {
		if (ptot[6] != 0 || ptot[2] > 0x7fffffffULL) { mga_set_error("gap list: too many WFA problems or target bases in one chunk (%llu problems); lower MGA_CHUNK", ptot[2]); rc = -1; goto done; }
		CK(mga_hbuf_reserve(&P->h_plrev, (size_t)ptot[0] * 4 + 16));
		memset(P->h_plrev.p, 0, (size_t)ptot[0] * 4); /* (reads that chain_worker skips -- max_qlen, empty -- leave their chains' flags unwritten: defined now; their lines are never printed) */
		mga_batch_set_device_plan(b, h_ploff, (int32_t*)P->h_plrev.p, (int64_t)ptot[0]);
	}
	if (dev_gc) mga_batch_set_device_chains(b, h_gchdr_p, P->h_gcpool.p, (const mg_llchain_t*)P->h_lcpool.p, need_a ? (const mg128_t*)P->h_apool.p : 0);
	rq_dev_ctx_t rq_ctx;
	if (long_q && n_a > 0 && env_int("MGA_DEV_RMQ", 1)) { /* -x asm (and the rescue pass of ultra-long -x lr reads): the RMQ chainer's forward passes on the device (k_rmq.hip) */
		rq_ctx.P = P, rq_ctx.opt = opt, rq_ctx.pen_gap = b->pen_gap, rq_ctx.pen_skip = b->pen_skip, rq_ctx.n_total = n_a;
		CK(mga_hbuf_reserve(&P->h_rq, (size_t)((n_a + 1) & ~1LL) * 20 + 64));
		mga_batch_set_rq_device(b, rq_dev_fwd_hook, &rq_ctx, (char*)P->h_rq.p, n_a);
	}
	CK(mga_batch_chain(b, h_nmz, h_rep, (const int32_t*)P->h_mini.p, h_minioff, h_nu, h_nb, (const uint64_t*)P->h_u.p, (const mg128_t*)P->h_b.p, h_aoff, chunk_long ? 3 : long_q ? 2 : is_rmq, h_rflag));
	if (g_dbg_pipe > 1) PIPE_LOG(" hostchain", n, t0);
	t1 = mga_wtime(); st->t_host_chain += t1 - t0; t0 = t1;
	/* ---- WFA over all gaps ---- */
	/* the device-made gap list (dev_plan) comes first in every array; what the host threads planned (all reads, or with dev_plan the few reads chained here) follows */
	const int64_t dp_chain = (int64_t)ptot[0], dp_item = (int64_t)ptot[1], dp_prob = (int64_t)ptot[2], dp_vert = (int64_t)ptot[3], dp_tb = (int64_t)ptot[4];
	const int64_t hp_prob = mga_batch_n_wfa(b), hp_tb = mga_batch_wfa_target_bytes(b);
	n_prob = dp_prob + hp_prob, n_tb = dp_tb + hp_tb;
	if ((opt->flag & MG_M_CIGAR) && (n_prob > 0 || (b->want_text && dp_chain + mga_batch_n_chains(b) > 0))) { /* (text mode: chains made of ready operators only still need their text) */
		mga_wfa_prob_t *h_prob;
		int64_t cells = 0;
		const int64_t n_item = dp_item + (b->want_text ? mga_batch_n_items(b) : 0), n_chain = dp_chain + (b->want_text ? mga_batch_n_chains(b) : 0), n_vert = dp_vert + (b->want_text ? mga_batch_n_verts(b) : 0);
		if (n_prob > 0x7fffffff) { mga_set_error("too many WFA problems in one chunk (%ld); lower MGA_CHUNK", (long)n_prob); rc = -1; goto done; }
		CK(mga_hbuf_reserve(&P->h_prob, (size_t)hp_prob * sizeof(mga_wfa_prob_t) + 16)); CK(mga_hbuf_reserve(&P->h_tseq, dev_splice ? 64 : (size_t)hp_tb + 64));
		h_prob = (mga_wfa_prob_t*)P->h_prob.p;
		if (dev_splice) { /* descriptors instead of bytes: 24 bytes per gap; the vertices they index follow the device-made part (dp_vert) of the walk array */
			CK(mga_hbuf_reserve(&P->h_plsrc, (size_t)hp_prob * sizeof(mga_plan_src_t) + 16));
			mga_batch_wfa_export_src(b, h_prob, 0, (mga_plan_src_t*)P->h_plsrc.p, dp_vert);
		} else {
			mga_batch_wfa_export(b, h_prob, (char*)P->h_tseq.p);
			memset((char*)P->h_tseq.p + hp_tb, 0, 64);
		}
		for (i = 0; dp_tb > 0 && i < hp_prob; ++i) h_prob[i].t_off += dp_tb;
		CK(mga_dbuf_reserve(&P->tseq, (size_t)n_tb + 64)); CK(mga_dbuf_reserve(&P->prob, (size_t)n_prob * sizeof(mga_wfa_prob_t) + 16)); CK(mga_dbuf_reserve(&P->res, (size_t)n_prob * sizeof(mga_wfa_res_t) + 16));
		CK(mga_dbuf_reserve(&P->used, 64));
		if (b->want_text) {
			CK(mga_dbuf_reserve(&P->item, (size_t)n_item * sizeof(mga_cigitem_t) + 16)); CK(mga_dbuf_reserve(&P->chain, (size_t)n_chain * sizeof(mga_txt_chain_t) + 16));
			CK(mga_dbuf_reserve(&P->vert, (size_t)n_vert * 4 + 16)); CK(mga_dbuf_reserve(&P->txtres, (size_t)n_chain * sizeof(mga_txt_res_t) + 16));
		}
		if (dev_plan || dev_splice) CK(mga_dbuf_reserve(&P->plsrc, (size_t)n_prob * sizeof(mga_plan_src_t) + 16));
		if (dev_plan) { /* pass 2 of k_plan.hip: items, problems + their targets, printed chains and walks, straight from the record pools in HBM */
			CK(mga_dbuf_reserve(&P->plrev, (size_t)dp_chain * 4 + 16));
			CK(mga_h2d_s(sc, P->plrev.p, P->h_plrev.p, (size_t)dp_chain * 4));
			CK(mga_dev_plan_fill(sc, &B->dev, n, (opt->flag & MG_M_PRINT_2ND) != 0, (const mga_gc_hdr_t*)P->gchdr.p, P->gcpool.p, (const mg_llchain_t*)P->lcpool.p, (const mg128_t*)P->apool.p,
								 (const int64_t*)P->qoff.p, (const int64_t*)P->ploff.p, (const int32_t*)P->plrev.p, dp_prob, (mga_cigitem_t*)P->item.p, (mga_wfa_prob_t*)P->prob.p,
								 (mga_plan_src_t*)P->plsrc.p, (mga_txt_chain_t*)P->chain.p, (uint32_t*)P->vert.p, (char*)P->tseq.p));
		}
		/* uploads ride the copy engine while another chunk owns the WFA phase */
		CK(mga_h2d_s(sc, (mga_wfa_prob_t*)P->prob.p + dp_prob, h_prob, (size_t)hp_prob * sizeof(mga_wfa_prob_t)));
		if (dev_splice) { /* walk vertices first (the text kernel's inputs travel now instead of after the ladder), then the targets are spliced where the graph's sequence lies */
			const int64_t hp_item = n_item - dp_item, hp_chain = n_chain - dp_chain, hp_vert = n_vert - dp_vert;
			int64_t k;
			CK(mga_hbuf_reserve(&P->h_item, (size_t)hp_item * sizeof(mga_cigitem_t) + 16)); CK(mga_hbuf_reserve(&P->h_chain, (size_t)hp_chain * sizeof(mga_txt_chain_t) + 16));
			CK(mga_hbuf_reserve(&P->h_vert, (size_t)hp_vert * 4 + 16));
			mga_batch_text_export(b, (mga_cigitem_t*)P->h_item.p, (mga_txt_chain_t*)P->h_chain.p, (uint32_t*)P->h_vert.p);
			for (k = 0; dev_plan && k < hp_chain; ++k) { /* behind the device-made part */
				mga_txt_chain_t *c = (mga_txt_chain_t*)P->h_chain.p + k;
				c->item_beg += dp_item, c->item_end += dp_item, c->vert_beg += dp_vert, c->prob_base += dp_prob;
			}
			CK(mga_h2d_s(sc, (mga_cigitem_t*)P->item.p + dp_item, P->h_item.p, (size_t)hp_item * sizeof(mga_cigitem_t)));
			CK(mga_h2d_s(sc, (mga_txt_chain_t*)P->chain.p + dp_chain, P->h_chain.p, (size_t)hp_chain * sizeof(mga_txt_chain_t)));
			CK(mga_h2d_s(sc, (uint32_t*)P->vert.p + dp_vert, P->h_vert.p, (size_t)hp_vert * 4));
			CK(mga_h2d_s(sc, (mga_plan_src_t*)P->plsrc.p + dp_prob, P->h_plsrc.p, (size_t)hp_prob * sizeof(mga_plan_src_t)));
			CK(mga_dmemset_s(sc, (char*)P->tseq.p + n_tb, 0, 64));
			CK(mga_dev_plan_target_verts(sc, &B->dev, hp_prob, (const mga_wfa_prob_t*)P->prob.p + dp_prob, (const mga_plan_src_t*)P->plsrc.p + dp_prob, (const uint32_t*)P->vert.p, (char*)P->tseq.p));
		} else CK(mga_h2d_s(sc, (char*)P->tseq.p + dp_tb, P->h_tseq.p, (size_t)hp_tb + 64));
		GPU_ACQUIRE(P->tok_wfa ? P->tok_wfa : &g_gpu_wfa);
		/* CIGAR pool: a global alignment has at most tl + ql operators, so target bases + query bases bound the chunk; + the abandoned block
		 * tails (<= 512 ops) of every resident wave.  Sized to the bound, the pool cannot overflow whatever the divergence. */
		pool_cap = n_tb + 4096 + 40000LL * 512 + MGA_WFA_FUSE_SLACK + (int64_t)ptot[5] + 4 * n_prob; /* (+ 4 per problem: k_wfa_tb reserves by an upper bound of the operator count) */
		for (i = 0; i < b->n_threads; ++i) pool_cap += b->tp[i].wfa_q_bases;
		CK(mga_dbuf_reserve(&P->pool, (size_t)pool_cap * 4)); CK(mga_dmemset_s(sc, P->used.p, 0, 8));
		/* the tier ladder runs on the device (k_wfa_sched.hip); the host never walks the problems */
		CK(mga_dev_wfa_solve(sc, (int)n_prob, (const mga_wfa_prob_t*)P->prob.p, (const char*)P->tseq.p, d_seq, (mga_wfa_res_t*)P->res.p,
							 (uint32_t*)P->pool.p, pool_cap, (unsigned long long*)P->used.p, &cells, env_int("MGA_EARLY_RELEASE", 1) ? release_token_cb : 0, &held));
		{ /* CIGARs back in problem order: one sequential stream for the host's stitching threads, or the input of the text kernel */
			int64_t n_ops = 0;
			CK(mga_dbuf_reserve(&P->ncig, (size_t)n_prob * 4 + 16)); CK(mga_dbuf_reserve(&P->cigoff, (size_t)(n_prob + 1) * 8)); CK(mga_dbuf_reserve(&P->ord, (size_t)pool_cap * 4));
			CK(mga_dev_wfa_gather(sc, (int)n_prob, (const mga_wfa_res_t*)P->res.p, (const uint32_t*)P->pool.p, (int32_t*)P->ncig.p, (int64_t*)P->cigoff.p,
								  (uint32_t*)P->ord.p, pool_cap, &n_ops));
			GPU_RELEASE(); /* the gather kernel and what follows trail on this chunk's stream while the next chunk's kernels start */
			if (b->want_text) { /* stitching, statistics, cg:Z and ds:Z on the device (k_text.hip): one lane per printed chain */
				const int64_t hp_item = n_item - dp_item, hp_chain = n_chain - dp_chain, hp_vert = n_vert - dp_vert;
				int64_t txt_cap = 4096, k;
				unsigned long long txt_used = 0;
				const int tight = env_int("MGA_TXT_TIGHT", 0); /* (MGA_TXT_TIGHT=1: a deliberately small pool, so that a test sees the second launch) */
				for (k = 0; k < n; ++k) txt_cap += (tight ? 1 : 3) * (int64_t)qlens[k] / (tight ? 4 : 1) + 1024;
				if (!dev_splice) { /* (with the targets spliced on the device these went up before the ladder) */
				CK(mga_hbuf_reserve(&P->h_item, (size_t)hp_item * sizeof(mga_cigitem_t) + 16)); CK(mga_hbuf_reserve(&P->h_chain, (size_t)hp_chain * sizeof(mga_txt_chain_t) + 16));
				CK(mga_hbuf_reserve(&P->h_vert, (size_t)hp_vert * 4 + 16));
				mga_batch_text_export(b, (mga_cigitem_t*)P->h_item.p, (mga_txt_chain_t*)P->h_chain.p, (uint32_t*)P->h_vert.p);
				for (k = 0; dev_plan && k < hp_chain; ++k) { /* behind the device-made part */
					mga_txt_chain_t *c = (mga_txt_chain_t*)P->h_chain.p + k;
					c->item_beg += dp_item, c->item_end += dp_item, c->vert_beg += dp_vert, c->prob_base += dp_prob;
				}
				CK(mga_h2d_s(sc, (mga_cigitem_t*)P->item.p + dp_item, P->h_item.p, (size_t)hp_item * sizeof(mga_cigitem_t)));
				CK(mga_h2d_s(sc, (mga_txt_chain_t*)P->chain.p + dp_chain, P->h_chain.p, (size_t)hp_chain * sizeof(mga_txt_chain_t)));
				CK(mga_h2d_s(sc, (uint32_t*)P->vert.p + dp_vert, P->h_vert.p, (size_t)hp_vert * 4));
				}
				for (k = 0;; ++k) { /* the pool is sized for ordinary reads (~1 byte of cg + ds per base); the kernel counts what it WOULD have written, so a chunk of
				                     * very divergent reads or many printed secondaries gets a pool of exactly that size and a second launch */
					CK(mga_dbuf_reserve(&P->txtpool, (size_t)txt_cap)); CK(mga_dmemset_s(sc, P->used.p, 0, 8));
					CK(mga_dev_text(sc, (int)n_chain, (const mga_txt_chain_t*)P->chain.p, (const mga_cigitem_t*)P->item.p, n_vert, (const uint32_t*)P->vert.p, &B->dev, d_seq,
									n_item + n_ops, (const int32_t*)P->ncig.p, (const int64_t*)P->cigoff.p, (const uint32_t*)P->ord.p, (mga_txt_res_t*)P->txtres.p,
									(char*)P->txtpool.p, txt_cap, (unsigned long long*)P->used.p));
					CK(mga_d2h_s(sc, &txt_used, P->used.p, 8)); CK(mga_ssync(sc));
					if ((int64_t)txt_used <= txt_cap) break;
					if (k >= 2) { mga_set_error("text kernel: output of %lld bytes exceeds the pool of %lld", (long long)txt_used, (long long)txt_cap); rc = -1; goto done; }
					txt_cap = (int64_t)txt_used + ((int64_t)txt_used >> 4) + 4096;
				}
				CK(mga_hbuf_reserve(&P->h_txtres, (size_t)n_chain * sizeof(mga_txt_res_t) + 16));
				CK(mga_d2h_s(sc, P->h_txtres.p, P->txtres.p, (size_t)n_chain * sizeof(mga_txt_res_t)));
				/* Round 6: the whole LINES are written on the device (k_gaf.hip) -- name, coordinates, path column, tags in front of the two strings, every line at its place in
				 * the chunk's text -- unless the job asks for what only the host's writer prints (per-vertex lines of -S / --write-mz) or the chains are chromosome-scale
				 * (their walks have 10^5 vertices: one wavefront per line would fold them alone).  MGA_DEV_GAF=0: the host formats as in rounds 1-5 (A/B, tests). */
				int64_t n_lines = -1;
				if (gaf_part && B->dev.d_gaf_seg != 0 && !(opt->flag & (MG_M_WRITE_LCHAIN | MG_M_WRITE_MZ | MG_M_FRAG_MERGE)) && !(mg_dbg_flag & 0x8) && n_chain > 0 &&
					(n_item + n_ops) / n_chain < 32768 && n_chain < 0x7fffffffLL && env_int("MGA_DEV_GAF", 1)) {
					CK(mga_hbuf_reserve(&P->h_gline, (size_t)(n_chain + n) * sizeof(mga_gaf_line_t) + 64));
					n_lines = gaf_lines_build(b, n, (mga_gaf_line_t*)P->h_gline.p);
				}
				if (n_lines > 0) {
					int64_t qn_bytes = 0, tot_b = 0, *qo;
					char *qn;
					for (i = 0; i < n; ++i) qn_bytes += qnames ? (int64_t)strlen(qnames[i]) : 1;
					CK(mga_hbuf_reserve(&P->h_gqn, (size_t)(n + 1) * 8 + (size_t)qn_bytes + 64));
					qo = (int64_t*)P->h_gqn.p, qn = (char*)(qo + n + 1);
					for (i = 0, qn_bytes = 0; i < n; ++i) { const char *nm = qnames ? qnames[i] : "*"; const size_t l = strlen(nm); qo[i] = qn_bytes; memcpy(qn + qn_bytes, nm, l); qn_bytes += (int64_t)l; }
					qo[n] = qn_bytes;
					CK(mga_dbuf_reserve(&P->g_line, (size_t)n_lines * sizeof(mga_gaf_line_t) + 64)); CK(mga_dbuf_reserve(&P->g_qoff, (size_t)(n + 1) * 8)); CK(mga_dbuf_reserve(&P->g_qn, (size_t)qn_bytes + 64));
					CK(mga_dbuf_reserve(&P->g_len, (size_t)(n_lines + 1) * 4)); CK(mga_dbuf_reserve(&P->g_off, (size_t)(n_lines + 2) * 8));
					CK(mga_h2d_s(sc, P->g_line.p, P->h_gline.p, (size_t)n_lines * sizeof(mga_gaf_line_t))); CK(mga_h2d_s(sc, P->g_qoff.p, qo, (size_t)(n + 1) * 8)); CK(mga_h2d_s(sc, P->g_qn.p, qn, (size_t)qn_bytes + 1));
					CK(mga_dev_gaf(sc, &B->dev, (int)n_lines, (const mga_gaf_line_t*)P->g_line.p, (const char*)P->g_qn.p, (const int64_t*)P->g_qoff.p, opt->flag, (const mga_txt_chain_t*)P->chain.p,
								   (const uint32_t*)P->vert.p, (const mga_txt_res_t*)P->txtres.p, (const char*)P->txtpool.p, (int32_t*)P->g_len.p, 0, 0));
					CK(mga_dev_scan_i32_to_i64(sc, (const int32_t*)P->g_len.p, n_lines, (int64_t*)P->g_off.p));
					CK(mga_d2h_s(sc, &tot_b, (int64_t*)P->g_off.p + n_lines, 8)); CK(mga_ssync(sc));
					CK(mga_dbuf_reserve(&P->g_out, (size_t)tot_b + 64)); CK(mga_hbuf_reserve(&P->h_gout, (size_t)tot_b + 64));
					CK(mga_dev_gaf(sc, &B->dev, (int)n_lines, (const mga_gaf_line_t*)P->g_line.p, (const char*)P->g_qn.p, (const int64_t*)P->g_qoff.p, opt->flag, (const mga_txt_chain_t*)P->chain.p,
								   (const uint32_t*)P->vert.p, (const mga_txt_res_t*)P->txtres.p, (const char*)P->txtpool.p, (int32_t*)P->g_len.p, (const int64_t*)P->g_off.p, (char*)P->g_out.p));
					if (tot_b > 0) CK(mga_d2h_s(sc, P->h_gout.p, P->g_out.p, (size_t)tot_b));
					b->gaf_dev = (const char*)P->h_gout.p, b->gaf_dev_len = tot_b;
				} else if (n_lines == 0) b->gaf_dev = "", b->gaf_dev_len = 0; /* (nothing of this chunk is printed) */
				else { CK(mga_hbuf_reserve(&P->h_txtpool, (size_t)txt_used + 16)); CK(mga_d2h_s(sc, P->h_txtpool.p, P->txtpool.p, (size_t)txt_used)); }
				CK(mga_ssync(sc));
				for (k = 0; k < n_chain; ++k)
					if (((const mga_txt_res_t*)P->h_txtres.p)[k].status != 0) { mga_set_error("text kernel: stitched CIGAR inconsistent with the chain coordinates (galign.c:140), chain %ld", (long)k); rc = -1; goto done; }
				b->txt_res = (const mga_txt_res_t*)P->h_txtres.p, b->txt_pool = b->gaf_dev ? 0 : (const char*)P->h_txtpool.p;
			} else {
				CK(mga_hbuf_reserve(&P->h_ncig, (size_t)n_prob * 4 + 16)); CK(mga_hbuf_reserve(&P->h_cigoff, (size_t)(n_prob + 1) * 8)); CK(mga_hbuf_reserve(&P->h_pool, (size_t)n_ops * 4 + 16));
				CK(mga_d2h_s(sc, P->h_ncig.p, P->ncig.p, (size_t)n_prob * 4)); CK(mga_d2h_s(sc, P->h_cigoff.p, P->cigoff.p, (size_t)(n_prob + 1) * 8));
				CK(mga_d2h_s(sc, P->h_pool.p, P->ord.p, (size_t)n_ops * 4)); CK(mga_ssync(sc));
			}
		}
		st->wfa_cells += cells;
		st->n_wfa += dp_prob, st->n_wfa_dev_plan += dp_prob, st->wfa_t_bases += dp_tb, st->wfa_q_bases += (int64_t)ptot[5];
	}
	if (g_dbg_pipe > 1) PIPE_LOG(" wfa", n, t0);
	t1 = mga_wtime(); st->t_wfa += t1 - t0; t0 = t1;
	/* ---- host: CIGAR stitching + ds ---- */
	if (b->txt_res) {} /* nothing to stitch on the host: the GAF writer copies the device's text */
	else if ((opt->flag & MG_M_CIGAR) && n_prob > 0) CK(mga_batch_finish_ordered(b, (const int32_t*)P->h_ncig.p, (const int64_t*)P->h_cigoff.p, (const uint32_t*)P->h_pool.p));
	else CK(mga_batch_finish(b, 0, 0));
	if (g_dbg_pipe > 1) PIPE_LOG(" hostpost", n, t0);
	t1 = mga_wtime(); st->t_host_post += t1 - t0;
	if (gaf_part) { /* GAF text of this chunk, formatted while other chunks own the GPU; the chains are freed on the way */
		gafw_t w;
		double tg = mga_wtime();
		int direct = 0;
		w.b = b, w.n = n, w.T = n_threads, w.part = gaf_part, w.mode = 0, w.bytes = w.off = 0, w.dst = 0;
		if (b->gaf_dev) { /* the lines came from the device: straight to their place in the job's output if this chunk is the one the output waits for, else into the chunk's pieces */
			pcopy_t pc;
			pc.src = b->gaf_dev, pc.bytes = b->gaf_dev_len, pc.T = n_threads, pc.dst = 0;
			if (sink && env_int("MGA_GAF_DIRECT", 1) && sink->try_reserve(sink->ctx, pc.bytes, &pc.dst)) {
				mga_parallel_for(n_threads, n_threads, pcopy_worker, &pc);
				sink->commit(sink->ctx, pc.bytes);
				st->gaf_bytes += pc.bytes;
			} else { /* piece t takes the bytes [bytes t / T, bytes (t + 1) / T): the pieces are concatenated in order, wherever a line is cut */
				int t_;
				for (t_ = 0; t_ < n_threads; ++t_) {
					const int64_t pb = pc.bytes * t_ / n_threads, pe = pc.bytes * (t_ + 1) / n_threads;
					kstring_t *out = &gaf_part[t_];
					if (pe - pb > 0xfffffff0LL) { mga_set_error("more than 4 GB of GAF text in one output piece: use a smaller -K or more threads"); rc = -1; goto done; }
					if ((size_t)(pe - pb) + 1 > out->m) { size_t cap; char *p = strbuf_get((size_t)(pe - pb) + 1, &cap); if (cap > 0xfffffff0u) cap = 0xfffffff0u; free(out->s); out->s = p, out->m = (unsigned)cap; }
					memcpy(out->s, pc.src + pb, (size_t)(pe - pb));
					out->l = (unsigned)(pe - pb), out->s[out->l] = 0;
				}
			}
			for (i = 0; i < n; ++i) { mg_gchain_free(b->gcs[i]); b->gcs[i] = 0; gcs_out[i] = 0; }
			direct = 1;
		} else
		if (sink && b->txt_res && env_int("MGA_GAF_DIRECT", 1)) { /* straight to the lines' place in the job's output, if this chunk is what the output waits for */
			int64_t *bo = MGA_CALLOC(int64_t, 2 * (size_t)n_threads + 2), tot_b = 0;
			int t_;
			w.bytes = bo, w.off = bo + n_threads + 1, w.mode = 1;
			mga_parallel_for(n_threads, n_threads, gaf_worker, &w);
			int fits = 1; /* kstring_t::l counts in 32 bits: a piece beyond that (chromosome-scale text on few threads) takes the piece path, which fails loudly at 4 GB (gaf.c: ks_room) */
			for (t_ = 0; t_ < n_threads; ++t_) { w.off[t_] = tot_b, tot_b += w.bytes[t_]; if (w.bytes[t_] > 0xfffffff0LL) fits = 0; }
			if (fits && sink->try_reserve(sink->ctx, tot_b, &w.dst)) {
				w.mode = 2;
				mga_parallel_for(n_threads, n_threads, gaf_worker, &w);
				sink->commit(sink->ctx, tot_b);
				st->gaf_bytes += tot_b;
				direct = 1;
			}
			free(bo);
			w.mode = 0, w.bytes = w.off = 0;
		}
		if (!direct) mga_parallel_for(n_threads, n_threads, gaf_worker, &w);
		for (i = 0; i < n; ++i) gcs_out[i] = 0;
		if (g_dbg_pipe > 1) PIPE_LOG(" gaf", n, tg);
		st->t_gaf += mga_wtime() - tg;
	} else {
		mg_gchains_t **r = mga_batch_take_results(b);
		for (i = 0; i < n; ++i) gcs_out[i] = r[i];
		free(r);
	}
	for (i = 0, n_mz = 0; i < n; ++i) n_mz += h_nmz[i]; /* (above, n_mz was the capacity of the minimizer slots) */
	st->n_reads += n, st->n_bases += tot, st->n_mz += n_mz, st->n_probe += n_mz, st->n_hit += n_a;
	for (i = 0; i < n && h_nb; ++i) st->n_anchor_chained += h_nb[i];
	for (i = 0; i < n && h_rflag; ++i) st->n_rescue_dev += h_rflag[i] == 1, st->n_rescue_host += h_rflag[i] == 2;
	mga_batch_stats(b, st);
done:
	if (rc < 0) mga_sctx_abort(sc); /* nothing of this chunk may still be in flight, and no staged read-back may outlive this frame */
	GPU_RELEASE();
	if (b) mga_batch_destroy(b);
	free(q_off); free(h_mzoff); free(h_aoff); free(h_minioff); free(h_nmz); free(h_rep); free(h_nu); free(h_nb); free(h_rflag);
	return rc;
}

/* ------------------------------------------------------------------------------------------------
 * stream: the chunk pipeline as a persistent object
 *
 * A stream owns MGA_PIPE pipeline threads (each with its own HIP stream context and device / pinned buffers) that live as long
 * as the stream.  Batches are SUBMITTED (cut into chunks, queued) and COLLECTED in submission order; chunks of consecutive
 * batches follow each other through the workers without a drain in between -- the reference overlaps its mini-batches the same
 * way (kt_pipeline, gmap.c:163-184).  mg_map_files() keeps one stream for the whole job, the single-batch entry points
 * (mg_map_batch, mga_map_reads) use the index's own stream, one call at a time.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sbatch_s {
	struct sbatch_s *next;
	int n, flags;
	const int *qlens;
	const char **seqs, **qnames;
	mg_gchains_t **gcs;            /* chain mode: results; text mode: scratch array of NULLs */
	int own_gcs;
	const char *d_seq; const int64_t *q_off;
	int seqs_pinned;               /* seqs[] lie back to back in pinned host memory (+64 readable bytes): chunks upload straight from there */
	int want_gaf, n_threads;
	int n_chunks, *cstart, next_chunk, n_done;
	kstring_t *gaf_part; char *done; int next_commit;
	pthread_mutex_t cmtx;
	char *out; int64_t out_len, out_cap;
	int err, complete, inline_ok;
	char errmsg[512];
	void *user;
} sbatch_t;

struct mga_stream_s {
	const mg_idx_t *gi;
	mg_mapopt_t opt;
	int n_threads, n_pipe, chunk, max_inflight;
	int lr_long;                   /* ultra-long -x lr reads: at least this many bases (0: none); fixed when the options are set */
	int job_chunks;                /* chunks the current job has been cut into so far (the ramp-up of the chunk size belongs to the job) */
	gpu_token_t tok_front, tok_wfa; /* chunks of THIS stream in their front (sketch / seeds / chaining) and WFA phase: per stream since round 5 -- a stream opened for a rank's two
	                                 * threads after one for sixteen used to inherit the first one's counts */
	pthread_mutex_t m;
	pthread_cond_t c_work, c_done, c_space;
	sbatch_t *head, *tail, *cur;   /* submitted and not yet collected (FIFO); cur = first batch that still has chunks to hand out */
	int n_inflight, closing, started, n_submitted;
	pipe_ctx_t P[MGA_MAX_PIPE];
	pthread_t thr[MGA_MAX_PIPE];
	pthread_mutex_t api;           /* single-batch entry points on the index's stream: one call at a time */
};

typedef struct { kstring_t *part; int64_t *off; char *dst; } gcopy_t;
static void gaf_copy_worker(void *data, int64_t i, int tid) { gcopy_t *g = (gcopy_t*)data; (void)tid; if (g->part[i].l) memcpy(g->dst + g->off[i], g->part[i].s, g->part[i].l); strbuf_put(g->part[i].s, g->part[i].m); g->part[i].s = 0, g->part[i].m = g->part[i].l = 0; }

/* room for `more` bytes (+ 1) at the end of the batch's output; the caller holds cmtx.  -1: out of memory, the buffer is what it was */
static int out_reserve(sbatch_t *b, int64_t more)
{
	if (b->out_len + more + 1 > b->out_cap) {
		const int64_t cap = (b->out_len + more + 1) * 3 / 2 + (1 << 20);
		char *p = (char*)realloc(b->out, (size_t)cap);
		if (p == 0) return -1;
		b->out = p, b->out_cap = cap;
	}
	return 0;
}

/* chunk c of batch b is formatted: append every chunk that is now complete AND next in read order to the batch's output buffer */
static void commit_chunks(sbatch_t *b, int c)
{
	pthread_mutex_lock(&b->cmtx);
	b->done[c] = 1;
	while (b->next_commit < b->n_chunks && b->done[b->next_commit]) {
		const int T = b->n_threads;
		gcopy_t g;
		int64_t off[T + 1], tot = 0;
		int k;
		g.part = b->gaf_part + (size_t)b->next_commit * T, g.off = off;
		for (k = 0; k < T; ++k) off[k] = tot, tot += g.part[k].l;
		if (out_reserve(b, tot) < 0) { /* the job fails loudly (the batch's error is what its caller gets back); the chunks behind this one are marked done and never copied */
			if (!b->err) { b->err = 1; snprintf(b->errmsg, sizeof b->errmsg, "out of memory: %lld bytes of GAF output", (long long)(b->out_len + tot)); }
			break;
		}
		g.dst = b->out + b->out_len;
		mga_parallel_for(T < 16 ? T : 16, T, gaf_copy_worker, &g);
		b->out_len += tot;
		++b->next_commit;
	}
	pthread_mutex_unlock(&b->cmtx);
}

static pthread_mutex_t g_stats_mtx = PTHREAD_MUTEX_INITIALIZER; /* the index's counters are shared by every stream and mg_tbuf_t */
static void stats_merge(mga_stats_t *d, const mga_stats_t *s)
{
	d->n_reads += s->n_reads, d->n_bases += s->n_bases, d->n_mz += s->n_mz, d->n_probe += s->n_probe, d->n_hit += s->n_hit;
	d->n_anchor_chained += s->n_anchor_chained, d->n_wfa += s->n_wfa, d->wfa_t_bases += s->wfa_t_bases, d->wfa_q_bases += s->wfa_q_bases;
	d->wfa_cells += s->wfa_cells;
	d->t_sketch += s->t_sketch, d->t_seed += s->t_seed, d->t_lchain += s->t_lchain, d->t_host_chain += s->t_host_chain, d->t_wfa += s->t_wfa, d->t_host_post += s->t_host_post;
	d->t_gaf += s->t_gaf;
	d->n_rescue_dev += s->n_rescue_dev, d->n_rescue_host += s->n_rescue_host;
	d->n_gwfa += s->n_gwfa, d->n_shortk += s->n_shortk, d->n_gc_retry += s->n_gc_retry, d->n_gc_host += s->n_gc_host;
	if (s->gc_arena_peak > d->gc_arena_peak) d->gc_arena_peak = s->gc_arena_peak;
	d->n_wfa_dev_plan += s->n_wfa_dev_plan;
	d->gaf_bytes += s->gaf_bytes;
}

static void pipe_ctx_free(pipe_ctx_t *P)
{
	size_t i;
	if (P->sc) mga_sctx_abort(P->sc);
	for (i = 0; i < sizeof P->dall / sizeof P->dall[0]; ++i) mga_dbuf_free(&P->dall[i]);
	for (i = 0; i < sizeof P->hall / sizeof P->hall[0]; ++i) mga_hbuf_free(&P->hall[i]);
	mga_sctx_destroy(P->sc);
	memset(P, 0, sizeof *P);
}

/* the sink of a chunk's GAF lines (gaf_sink_t): room at the end of the batch's output, if every earlier chunk has been committed */
typedef struct { sbatch_t *b; int c; } gaf_sink_ctx_t;
static int sink_try_reserve(void *ctx_, int64_t bytes, char **dst)
{
	gaf_sink_ctx_t *x = (gaf_sink_ctx_t*)ctx_;
	sbatch_t *b = x->b;
	int ok = 0;
	pthread_mutex_lock(&b->cmtx);
	if (b->next_commit == x->c && !b->err) { /* nobody else can append until this chunk is marked done: the room stays where it is while the pieces are written */
		if (out_reserve(b, bytes) == 0) { *dst = b->out + b->out_len; ok = 1; } /* (no room: the chunk goes the piece-wise way, whose commit reports the failure) */
	}
	pthread_mutex_unlock(&b->cmtx);
	return ok;
}
static void sink_commit(void *ctx_, int64_t bytes)
{
	gaf_sink_ctx_t *x = (gaf_sink_ctx_t*)ctx_;
	pthread_mutex_lock(&x->b->cmtx);
	x->b->out_len += bytes; /* (commit_chunks() then marks the chunk done with empty pieces and lets the chunks behind it follow) */
	pthread_mutex_unlock(&x->b->cmtx);
}

/* one chunk of one batch on pipeline context P; the caller holds no lock */
static void stream_run_chunk(mga_stream_t *S, pipe_ctx_t *P, sbatch_t *b, int c)
{
	const int st = b->cstart[c], en = b->cstart[c + 1];
	mga_stats_t cst;
	double tc = mga_wtime();
	int64_t tcpu = cpu_now();
	int rc;
	gaf_sink_ctx_t sink_ctx = { b, c };
	gaf_sink_t sink = { sink_try_reserve, sink_commit, &sink_ctx };
	memset(&cst, 0, sizeof cst);
	rc = b->err ? 0 : map_chunk(P, S->gi, en - st, b->qlens + st, b->seqs + st, b->qnames ? b->qnames + st : 0, b->gcs + st, &S->opt, b->n_threads,
								b->d_seq, b->q_off ? b->q_off + st : 0, b->seqs_pinned, &cst, b->gaf_part ? b->gaf_part + (size_t)c * b->n_threads : 0, S->lr_long, b->gaf_part ? &sink : 0);
	PIPE_LOG("map_chunk", c, tc);
	if (rc == 0 && b->gaf_part && !b->err) { /* the chunk's GAF text was formatted inside map_chunk(); append it to the output in read order */
		double t0 = mga_wtime();
		int k;
		for (k = 0; k < b->n_threads; ++k) cst.gaf_bytes += b->gaf_part[(size_t)c * b->n_threads + k].l;
		{ int64_t t_ = cpu_now(); commit_chunks(b, c); CPU_ADD(C_COMMIT, t_); }
		cst.t_gaf += mga_wtime() - t0;
	}
	CPU_ADD(C_PIPE, tcpu);
	pthread_mutex_lock(&g_stats_mtx); stats_merge(&S->gi->B->st, &cst); pthread_mutex_unlock(&g_stats_mtx);
	pthread_mutex_lock(&S->m);
	if (rc < 0 && !b->err) { b->err = 1; snprintf(b->errmsg, sizeof b->errmsg, "%s", mga_last_error()); }
	if (++b->n_done == b->n_chunks) { b->complete = 1; pthread_cond_broadcast(&S->c_done); }
	pthread_mutex_unlock(&S->m);
}

typedef struct { mga_stream_t *S; int k; } stream_thr_t;

static void *stream_worker(void *a)
{
	stream_thr_t *t = (stream_thr_t*)a;
	mga_stream_t *S = t->S;
	pipe_ctx_t *P = &S->P[t->k];
	free(t);
	if (mga_dev_bind_thread() < 0) return 0;
	pthread_mutex_lock(&S->m);
	for (;;) {
		sbatch_t *b;
		int c;
		while (!S->closing && (S->cur == 0 || S->cur->next_chunk >= S->cur->n_chunks)) {
			if (S->cur && S->cur->next) { S->cur = S->cur->next; continue; }
			pthread_cond_wait(&S->c_work, &S->m);
		}
		if (S->cur == 0 || S->cur->next_chunk >= S->cur->n_chunks) break; /* closing and nothing left */
		b = S->cur, c = b->next_chunk++;
		pthread_mutex_unlock(&S->m);
		stream_run_chunk(S, P, b, c);
		pthread_mutex_lock(&S->m);
	}
	pthread_mutex_unlock(&S->m);
	return 0;
}

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void segv_trace(int sig) { void *bt[64]; int n = backtrace(bt, 64); (void)sig; backtrace_symbols_fd(bt, n, 2); _exit(139); } /* MGA_SEGV_TRACE=1: frames of a crash to stderr (addr2line on the .so) */

mga_stream_t *mga_stream_open(const mg_idx_t *gi, const mg_mapopt_t *opt, int n_threads)
{
	mga_stream_t *S;
	int i;
	if (mga_dev_init() < 0) return 0;
	if (env_int("MGA_SEGV_TRACE", 0)) signal(SIGSEGV, segv_trace);
	if (g_dbg_pipe < 0) { g_dbg_pipe = env_int("MGA_DEBUG_PIPE", 0); g_gpu_wfa.avail = env_int("MGA_WFA_SLOTS", 2); g_gpu_front.avail = env_int("MGA_FRONT_SLOTS", 1); }
	g_cpu_on = g_dbg_pipe > 0;
	S = MGA_CALLOC(mga_stream_t, 1);
	S->gi = gi, S->opt = *opt, S->n_threads = n_threads > 0 ? n_threads : 1;
	S->lr_long = (opt->flag & MG_M_RMQ) ? 0 : lr_long_bases();
	/* round 5: a rank with 5 .. 12 host threads chains on the device (k_gchain: long-tailed launches, 117 ms of a 125 000-read step at a quarter of the wave slots): SIX chunks
	 * in flight, two of them in the front phase, fill those tails with other chunks' kernels -- [measured, bench workload, --placement device, 16 threads] 2.98 / 3.00 Gbp/s
	 * (4 chunks, one in the front phase) -> 3.24 (6 / 2) -> 3.27-3.30 with the persistent WFA grids at half size; with the chaining on the host threads the same knobs stay
	 * inside the run-to-run noise (3.51-3.60 vs 3.55-3.65) */
	S->n_pipe = env_int("MGA_PIPE", n_threads > 4 ? 6 : 4); /* round 6, chaining on the host threads, 16 of them (profiles/r06u_knobs.txt): 6 pipeline threads + 2 chunks in the front phase 4.68 Gbp/s, 5 + 2: 4.58, 4 + 1 (the default until then): 4.48-4.51 */ /* round 5, a rank pinned to 2 of 16 cores, two chunks in the front phase: 4 pipeline threads 2.51 Gbp/s, 3: 2.28 (round 4, one chunk in the front phase: 3 was the better one) */ /* [measured, round 4, a rank pinned to 2 of 16 cores] 3 pipeline threads: 2.31 Gbp/s at 0.70 CPU-s per step, 4: 2.23 at 0.82, 2: 2.07; with 16 threads and the chaining on the host 4 is the best (3.27 vs 2.96 vs 2.51) */
	if (S->n_pipe > MGA_MAX_PIPE) S->n_pipe = MGA_MAX_PIPE;
	if (S->n_pipe < 1) S->n_pipe = 1;
	S->chunk = env_int("MGA_CHUNK", 16384); /* [measured] larger launches amortise the tails of the WFA tiers: 4096 -> 8192 reads +5 %, -> 16384 another +5 % */
	if (S->chunk < 1) S->chunk = 1;
	S->max_inflight = env_int("MGA_INFLIGHT", 3);
	pthread_mutex_init(&S->m, 0); pthread_mutex_init(&S->api, 0);
	pthread_cond_init(&S->c_work, 0); pthread_cond_init(&S->c_done, 0); pthread_cond_init(&S->c_space, 0);
	pthread_mutex_init(&S->tok_front.m, 0); pthread_cond_init(&S->tok_front.c, 0); pthread_mutex_init(&S->tok_wfa.m, 0); pthread_cond_init(&S->tok_wfa.c, 0);
	S->tok_wfa.avail = env_int("MGA_WFA_SLOTS", 2);                 /* two chunks may be in their WFA phase: the second one fills the tails of the first */
	S->tok_front.avail = env_int("MGA_FRONT_SLOTS", 2); /* ... and with the chaining on the device two in the front phase ([measured] + 8 % at 16 threads, + 7 % at 2) */
	for (i = 0; i < S->n_pipe; ++i) {
		if ((S->P[i].sc = mga_sctx_create()) == 0) { while (--i >= 0) pipe_ctx_free(&S->P[i]); free(S); return 0; }
		S->P[i].tok_front = &S->tok_front, S->P[i].tok_wfa = &S->tok_wfa;
	}
	return S;
}

static void stream_start(mga_stream_t *S) /* worker threads are created with the first multi-chunk batch: one-read calls never need them */
{
	int i;
	if (S->started) return;
	for (i = 0; i < S->n_pipe; ++i) {
		stream_thr_t *t = MGA_CALLOC(stream_thr_t, 1);
		t->S = S, t->k = i;
		pthread_create(&S->thr[i], 0, stream_worker, t);
	}
	S->started = 1;
}

void mga_stream_set_opt(mga_stream_t *S, const mg_mapopt_t *opt, int n_threads) { S->opt = *opt; S->lr_long = (opt->flag & MG_M_RMQ) ? 0 : lr_long_bases(); if (n_threads > 0) S->n_threads = n_threads; } /* only while nothing is in flight */

/* chunk boundaries of a batch: MGA_CHUNK reads each at most; the FIRST batch of a job ramps up from chunk/4 and the LAST one ramps down
 * (fill and drain of the pipeline cost about one chunk time each).
 * MGA_CUT=1 (round 5, default): chunks of EQUAL size inside a batch instead of full chunks + a remainder -- a 50 000-read mini-batch is 4 x 12 500, not 3 x 16 384 + 848:
 * a chunk of a few hundred reads pays a pass's whole chain of launches and synchronisations for no work ([measured, bench workload] its three -K batches were cut into
 * 12 chunks, two of them of 848 and 424 reads).  MGA_TAIL=<levels>: the LAST batch ends in chunks of 1/2, 1/4, ... of a chunk (the drain of the pipeline: the last chunk
 * walks its stages with nothing behind it).  The output never depends on the cut (test_pipeline_knobs_do_not_change_the_output). */
typedef struct { int even, n_head, head_sz[8], n_tail, tail_tot, body_sz, tail_sz[8]; } cut_plan_t;

/* job_pos: chunks the JOB has been cut into before this batch; ramp_levels: the job's first chunks are chunk >> levels, ..., chunk >> 1 (round 5: the ramp belongs to the job,
 * not to its first batch -- the reader's first batch is a short one of 6 400 reads, so that the GPU starts early, and the ramp used to end with it) */
static void cut_plan_init(cut_plan_t *cp, int n, int chunk, int ramp, int flags, int even, int tail_levels, int job_pos, int ramp_levels)
{
	int k, used = 0, body;
	memset(cp, 0, sizeof *cp);
	cp->even = even, cp->body_sz = chunk;
	if (!even) return;
	if (ramp && (flags & MGA_SB_LAST)) for (k = 0; k < tail_levels && k < 8 && (chunk >> (k + 1)) >= 64; ++k) cp->tail_sz[cp->n_tail++] = chunk >> (k + 1), cp->tail_tot += chunk >> (k + 1);
	if (n < cp->tail_tot + chunk / 2) cp->n_tail = 0, cp->tail_tot = 0; /* too small for a tapered end */
	if (ramp_levels > 8) ramp_levels = 8;
	for (k = job_pos; ramp && k < ramp_levels; ++k) { /* the ramp pieces that are still due, as long as something of the batch is left behind them */
		const int sz = chunk >> (ramp_levels - k);
		if (sz < 1 || n - used - cp->tail_tot < sz + sz / 2) break;
		cp->head_sz[cp->n_head++] = sz, used += sz;
	}
	body = n - used - cp->tail_tot;
	if (body > 0) { const int nc = (body + chunk - 1) / chunk; cp->body_sz = (body + nc - 1) / nc; }
}

/* size of chunk number m of a batch of which `left` reads are left */
static int cut_plan_size(const cut_plan_t *cp, int chunk, int ramp, int flags, int m, int left)
{
	int sz = chunk, k, suf;
	if (!cp->even) { /* rounds 1-4: full chunks + what is left */
		if (ramp) {
			if (flags & MGA_SB_FIRST) { if (m == 0) sz = chunk / 4; else if (m == 1) sz = chunk / 2; }
			if ((flags & MGA_SB_LAST) && left <= chunk + chunk / 2) sz = left > chunk / 2 + chunk / 8 ? left - chunk / 2 : left; /* tail: the last chunk is chunk/2 (or what is left) */
		}
		return sz;
	}
	sz = m < cp->n_head ? cp->head_sz[m] : cp->body_sz;
	if (cp->n_tail > 0 && m >= cp->n_head) {
		if (left <= cp->tail_tot) { /* in the tail: the largest run of tail pieces that fits; what a base-capped chunk before left over is absorbed by this piece */
			for (k = 0, suf = cp->tail_tot; k < cp->n_tail && suf > left; ++k) suf -= cp->tail_sz[k];
			sz = k < cp->n_tail ? cp->tail_sz[k] + (left - suf) : left;
		} else if (left - sz < cp->tail_tot) sz = left - cp->tail_tot; /* the body's last chunk ends where the tail pieces begin */
	}
	return sz;
}

/* (tests) the read counts of the chunks a batch of n reads of equal length is cut into; returns their number */
int mga_debug_cut2(int n, int chunk, int flags, int even, int tail_levels, int job_pos, int ramp_levels, int *sizes, int max_sizes)
{
	cut_plan_t cp;
	int pos = 0, m = 0;
	const int ramp = n > 6 * chunk || (flags & (MGA_SB_FIRST | MGA_SB_LAST)) != (MGA_SB_FIRST | MGA_SB_LAST);
	cut_plan_init(&cp, n, chunk, ramp, flags, even, tail_levels, job_pos, ramp_levels);
	while (pos < n) {
		int sz = cut_plan_size(&cp, chunk, ramp, flags, m, n - pos);
		if (sz < 1) sz = 1;
		if (sz > n - pos) sz = n - pos;
		if (m < max_sizes) sizes[m] = sz;
		++m, pos += sz;
	}
	return m;
}
/* (the round-4 ramp: chunk / 4 and chunk / 2 at the head of the job's first batch) */
int mga_debug_cut(int n, int chunk, int flags, int even, int tail_levels, int *sizes, int max_sizes) { return mga_debug_cut2(n, chunk, flags, even, tail_levels, (flags & MGA_SB_FIRST) ? 0 : 2, 2, sizes, max_sizes); }

static void batch_cut(mga_stream_t *S, sbatch_t *b)
{
	const int n = b->n, chunk = S->chunk;
	const int ramp = env_int("MGA_RAMP", 1) && (n > 6 * chunk || (b->flags & (MGA_SB_FIRST | MGA_SB_LAST)) != (MGA_SB_FIRST | MGA_SB_LAST)); /* a batch of a longer job always ramps */
	/* a chunk is also bounded in BASES: its device buffers grow with bases and anchors, not with reads, and a chunk of 16384 reads of 100 kb
	 * each would be ten times the footprint every measurement was taken at.  MGA_CHUNK reads of 12 kb is the cap (10 kb reads: never reached). */
	const int64_t base_cap = (S->opt.flag & MG_M_RMQ) ? INT64_MAX : (int64_t)chunk * env_int("MGA_CHUNK_READ_BASES", 12288); /* (-x asm: a batch is a handful of contigs chained in phases over ALL of them) */
	int pos = 0, m = 0, cap = n / (chunk / 16 > 0 ? chunk / 16 : 1) + 16;
	cut_plan_t cp;
	{ /* MGA_JOBRAMP=<levels> (default 3): the job's first chunks are chunk/8, chunk/4, chunk/2, whatever batches they fall into; 0: the round-4 rule (chunk/4, chunk/2 in the first batch) */
		const int lv = env_int("MGA_JOBRAMP", 3);
		cut_plan_init(&cp, n, chunk, ramp, b->flags, env_int("MGA_CUT", 1), env_int("MGA_TAIL", 1), lv > 0 ? S->job_chunks : ((b->flags & MGA_SB_FIRST) ? 0 : 2), lv > 0 ? lv : 2);
	}
	b->cstart = MGA_MALLOC(int, cap + 1);
	while (pos < n) {
		int sz = cut_plan_size(&cp, chunk, ramp, b->flags, m, n - pos), left = n - pos, k;
		int64_t bases = 0;
		if (sz < 1) sz = 1;
		if (sz > left) sz = left;
		{ /* ultra-long -x lr reads (>= MGA_LONG_READ bases, default 256 k) travel in chunks of their own -- at most 64 Mbp of them -- through the long-query path (map_chunk) */
			const int lr_long = S->lr_long;
			const int first_long = lr_long > 0 && b->qlens[pos] >= lr_long;
			for (k = 0; k < sz; ++k) {
				if (lr_long > 0 && k > 0 && (b->qlens[pos + k] >= lr_long) != first_long) break;
				bases += b->qlens[pos + k];
				if ((bases > base_cap || (first_long && bases > (64LL << 20))) && k > 0) break; /* (a single read longer than the cap is a chunk of its own) */
			}
		}
		sz = k;
		if (m == cap) { cap += cap / 2 + 8; b->cstart = MGA_REALLOC(int, b->cstart, cap + 1); }
		b->cstart[m++] = pos; pos += sz;
	}
	b->cstart[m] = n;
	b->n_chunks = m;
	S->job_chunks += m;
}

/* Queue a batch.  Everything passed in is BORROWED until the batch has been collected.  gcs != NULL: chain mode (results in gcs[]);
 * want_gaf: text mode (GAF bytes of the batch, input order).  out/out_cap: an output buffer to reuse (may be NULL).  Blocks while
 * MGA_INFLIGHT batches are already in flight. */
int mga_stream_submit(mga_stream_t *S, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs, int want_gaf,
					  const char *d_seq, const int64_t *q_off, int seqs_pinned, int flags, char *out, int64_t out_cap, void *user)
{
	sbatch_t *b = MGA_CALLOC(sbatch_t, 1);
	int i;
	b->n = n, b->flags = flags, b->qlens = qlens, b->seqs = seqs, b->qnames = qnames, b->d_seq = d_seq, b->q_off = q_off, b->seqs_pinned = seqs_pinned;
	b->want_gaf = want_gaf, b->user = user, b->out = out, b->out_cap = out ? out_cap : 0;
	b->n_threads = S->n_threads;
	if (gcs) b->gcs = gcs; else b->gcs = MGA_CALLOC(mg_gchains_t*, n > 0 ? n : 1), b->own_gcs = 1;
	for (i = 0; i < n; ++i) b->gcs[i] = 0;
	if (flags & MGA_SB_FIRST) S->job_chunks = 0; /* a new job: its ramp starts over (batches are submitted by one thread) */
	batch_cut(S, b);
	if (want_gaf) { b->gaf_part = MGA_CALLOC(kstring_t, (size_t)(b->n_chunks > 0 ? b->n_chunks : 1) * b->n_threads); b->done = MGA_CALLOC(char, b->n_chunks > 0 ? b->n_chunks : 1); }
	pthread_mutex_init(&b->cmtx, 0);
	if (b->n_chunks == 0) b->complete = 1;
	pthread_mutex_lock(&S->m);
	while (S->n_inflight >= S->max_inflight) pthread_cond_wait(&S->c_space, &S->m);
	if (S->n_submitted++ == 0) { g_job_t0 = mga_wtime(); if (g_cpu_on) memset((void*)g_cpu_ns, 0, sizeof g_cpu_ns); }
	if (S->tail) S->tail->next = b; else S->head = b;
	S->tail = b;
	if (S->cur == 0) S->cur = b;
	++S->n_inflight;
	/* Only a job that is ONE single-chunk batch (FIRST|LAST: one-read calls, small mg_map_batch calls) may run inline on the collecting thread
	 * (mga_stream_collect, context 0).  Any other batch starts the workers NOW: a single-chunk first batch of a longer job (e.g. 64 Mbp of 50 kb reads)
	 * used to run inline on P[0] while the next batch started worker 0 on the SAME context -- two threads on one set of device / staging buffers
	 * (the "chunks of more than 16384 reads" fault of round 2: any chunk size that made the job's first batch a single chunk). */
	b->inline_ok = b->n_chunks <= 1 && !S->started && (flags & (MGA_SB_FIRST | MGA_SB_LAST)) == (MGA_SB_FIRST | MGA_SB_LAST);
	if (!b->inline_ok) { stream_start(S); pthread_cond_broadcast(&S->c_work); }
	pthread_mutex_unlock(&S->m);
	return 0;
}

/* Collect the oldest batch: blocks until all of its chunks are done.  Returns 1 (a batch: out, out_len and out_cap receive its GAF buffer,
 * now owned by the caller), 0 (nothing in flight) or -1 (the batch failed: message in mga_last_error()). */
int mga_stream_collect(mga_stream_t *S, char **out, int64_t *out_len, int64_t *out_cap, void **user)
{
	sbatch_t *b;
	int rc = 1, i;
	pthread_mutex_lock(&S->m);
	b = S->head;
	if (b == 0) { pthread_mutex_unlock(&S->m); return 0; }
	if (b->inline_ok && !S->started) { /* a single-chunk, single-batch job on a stream without workers runs right here, on context 0 (one-read calls: no thread hand-off) */
		while (b->next_chunk < b->n_chunks) {
			int c = b->next_chunk++;
			pthread_mutex_unlock(&S->m);
			stream_run_chunk(S, &S->P[0], b, c);
			pthread_mutex_lock(&S->m);
		}
	}
	while (!b->complete) pthread_cond_wait(&S->c_done, &S->m);
	S->head = b->next;
	if (S->head == 0) S->tail = 0;
	if (S->cur == b) S->cur = b->next;
	--S->n_inflight;
	pthread_cond_broadcast(&S->c_space);
	pthread_mutex_unlock(&S->m);
	if (b->err) {
		for (i = 0; i < b->n; ++i) { mg_gchain_free(b->gcs[i]); b->gcs[i] = 0; }
		if (b->gaf_part) for (i = 0; i < b->n_chunks * b->n_threads; ++i) strbuf_put(b->gaf_part[i].s, b->gaf_part[i].m);
		mga_set_error("%s", b->errmsg[0] ? b->errmsg : "mapping pipeline failed");
		rc = -1;
	}
	if (b->want_gaf) {
		if (b->out == 0) b->out = (char*)malloc(1), b->out_cap = 1;
		b->out[b->out_len] = 0;
	}
	if (out) *out = b->out; else free(b->out);
	if (out_len) *out_len = b->out_len;
	if (out_cap) *out_cap = b->out_cap;
	if (user) *user = b->user;
	if (g_cpu_on && rc > 0) {
		struct timespec ts;
		fprintf(stderr, "[pipe] host CPU seconds by stage (cumulative, batch of %d reads done at %.3f s):", b->n, mga_wtime() - g_job_t0);
		for (i = 0; i < C_N; ++i) fprintf(stderr, " %s %.3f", g_cname[i], g_cpu_ns[i] * 1e-9);
		clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
		fprintf(stderr, "; process total so far %.3f\n", ts.tv_sec + ts.tv_nsec * 1e-9);
	}
	pthread_mutex_destroy(&b->cmtx);
	if (b->own_gcs) free(b->gcs);
	free(b->cstart); free(b->gaf_part); free(b->done); free(b);
	return rc;
}

void mga_stream_close(mga_stream_t *S)
{
	int i;
	if (S == 0) return;
	while (mga_stream_collect(S, 0, 0, 0, 0) != 0) {} /* drop what was never collected */
	pthread_mutex_lock(&S->m);
	S->closing = 1;
	pthread_cond_broadcast(&S->c_work);
	pthread_mutex_unlock(&S->m);
	if (S->started) for (i = 0; i < S->n_pipe; ++i) pthread_join(S->thr[i], 0);
	for (i = 0; i < S->n_pipe; ++i) pipe_ctx_free(&S->P[i]);
	pthread_mutex_destroy(&S->m); pthread_mutex_destroy(&S->api);
	pthread_cond_destroy(&S->c_work); pthread_cond_destroy(&S->c_done); pthread_cond_destroy(&S->c_space);
	free(S);
}

/* the index's own stream: created on first use, closed by mg_idx_destroy() */
static pthread_mutex_t g_idx_stream_mtx = PTHREAD_MUTEX_INITIALIZER;
static mga_stream_t *idx_stream(const mg_idx_t *gi, const mg_mapopt_t *opt, int n_threads)
{
	struct mg_idx_bucket_s *B = gi->B;
	pthread_mutex_lock(&g_idx_stream_mtx);
	if (B->stream == 0) B->stream = mga_stream_open(gi, opt, n_threads);
	pthread_mutex_unlock(&g_idx_stream_mtx);
	return (mga_stream_t*)B->stream;
}
/* the index's stream for a whole job (mg_map_files): locked until released, so that its pipeline contexts -- device buffers, pinned staging,
 * HIP streams: hundreds of MB that are expensive to allocate -- are reused from job to job instead of being rebuilt */
mga_stream_t *mga_idx_stream_acquire(const mg_idx_t *gi, const mg_mapopt_t *opt, int n_threads)
{
	mga_stream_t *S = idx_stream(gi, opt, n_threads);
	if (S == 0) return 0;
	pthread_mutex_lock(&S->api);
	mga_stream_set_opt(S, opt, n_threads);
	S->n_submitted = 0; /* a new job: its first batch restarts the debug clocks */
	return S;
}
void mga_idx_stream_release(mga_stream_t *S) { pthread_mutex_unlock(&S->api); }

void mga_idx_stream_close(mg_idx_t *gi) { if (gi && gi->B && gi->B->stream) { mga_stream_close((mga_stream_t*)gi->B->stream); gi->B->stream = 0; } }

/* one batch through the index's stream, start to finish.  Calls are serialized per index (the pipeline contexts are
 * shared state); callers that want concurrency use one mg_tbuf_t per thread with mg_map()/mg_map_frag(), or their own stream. */
static int map_all(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs,
				   const mg_mapopt_t *opt, int n_threads, const char *d_seq, const int64_t *q_off, char **gaf, int64_t *gaf_len)
{
	mga_stream_t *S;
	struct mg_idx_bucket_s *B = gi->B;
	int rc, i;
	for (i = 0; i < n; ++i) gcs[i] = 0;
	if (n <= 0) return 0;
	if ((S = idx_stream(gi, opt, n_threads)) == 0) return -1;
	pthread_mutex_lock(&S->api);
	mga_stream_set_opt(S, opt, n_threads);
	if (gaf) {
		char *out = B->gaf_out; int64_t cap = B->gaf_cap, len = 0;
		B->gaf_out = 0, B->gaf_cap = 0;
		mga_stream_submit(S, n, qlens, seqs, qnames, gcs, 1, d_seq, q_off, 0, MGA_SB_FIRST | MGA_SB_LAST, out, cap, 0);
		rc = mga_stream_collect(S, &out, &len, &cap, 0);
		B->gaf_out = out, B->gaf_cap = cap; /* the GAF text stays owned by the index */
		*gaf = out, *gaf_len = rc > 0 ? len : 0;
	} else {
		mga_stream_submit(S, n, qlens, seqs, qnames, gcs, 0, d_seq, q_off, 0, MGA_SB_FIRST | MGA_SB_LAST, 0, 0, 0);
		rc = mga_stream_collect(S, 0, 0, 0, 0);
	}
	pthread_mutex_unlock(&S->api);
	return rc > 0 ? 0 : -1;
}

/* map + format: the GAF text (input order) of n reads; *gaf points into a grow-only buffer OWNED BY THE INDEX, valid until the
 * next call on this index or mg_idx_destroy(); chains are not returned */
int mga_map_gaf(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, const mg_mapopt_t *opt, int n_threads,
				const char *d_seq, const int64_t *q_off, char **gaf, int64_t *gaf_len)
{
	mg_gchains_t **gcs = MGA_CALLOC(mg_gchains_t*, n > 0 ? n : 1);
	int rc;
	*gaf = 0, *gaf_len = 0;
	rc = map_all(gi, n, qlens, seqs, qnames, gcs, opt, n_threads, d_seq, q_off, gaf, gaf_len);
	free(gcs);
	return rc;
}

int mg_map_batch(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs,
				 const mg_mapopt_t *opt, int n_threads)
{
	return map_all(gi, n, qlens, seqs, qnames, gcs, opt, n_threads, 0, 0, 0, 0);
}

/* same as mg_map_batch() for reads that already sit in HBM: d_seq holds the reads back to back (+64 readable bytes),
 * q_off[n+1] are their absolute offsets. */
int mga_map_batch_resident(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs,
						   const mg_mapopt_t *opt, int n_threads, const char *d_seq, const int64_t *q_off)
{
	return map_all(gi, n, qlens, seqs, qnames, gcs, opt, n_threads, d_seq, q_off, 0, 0);
}

/* ---- the reference's per-read API (map-algo.c:14-32,340-502).  Its threading contract is one mg_tbuf_t per worker thread
 * (gmap.c:84-86, ggen.c:36): here a mg_tbuf_t owns a pipeline context (HIP stream + device buffers), created on first use, so
 * that kt_for workers calling mg_map() concurrently never share device state.  b == NULL falls back to the index's stream. ---- */
struct mg_tbuf_s { pipe_ctx_t P; };
mg_tbuf_t *mg_tbuf_init(void) { return (mg_tbuf_t*)calloc(1, sizeof(mg_tbuf_t)); }
void mg_tbuf_destroy(mg_tbuf_t *b) { if (b == 0) return; if (b->P.sc) { mga_dev_bind_thread(); pipe_ctx_free(&b->P); } free(b); }

void mg_map_frag(const mg_idx_t *gi, int n_segs, const int *qlens, const char **seqs, mg_gchains_t **gcs, mg_tbuf_t *b, const mg_mapopt_t *opt, const char *qname)
{
	int i;
	for (i = 0; i < n_segs; ++i) gcs[i] = 0;
	if (n_segs != 1) { /* multi-segment (paired short reads) is outside the accelerated path: the reference itself returns NULLs for n_segs out of range */
		if (mg_verbose >= 1) fprintf(stderr, "[E::%s] only single-segment reads are supported by the MI355X path\n", __func__);
		return;
	}
	if (b) { /* the caller's own pipeline context: concurrent callers with distinct mg_tbuf_t never share device state */
		mga_stats_t cst;
		int rc;
		memset(&cst, 0, sizeof cst);
		if (g_dbg_pipe < 0) { g_dbg_pipe = env_int("MGA_DEBUG_PIPE", 0); g_gpu_wfa.avail = env_int("MGA_WFA_SLOTS", 2); g_gpu_front.avail = env_int("MGA_FRONT_SLOTS", 1); }
		rc = mga_dev_init() < 0 || mga_dev_bind_thread() < 0 || (b->P.sc == 0 && (b->P.sc = mga_sctx_create()) == 0) ? -1 : 0;
		if (rc == 0) rc = map_chunk(&b->P, gi, 1, qlens, seqs, &qname, gcs, opt, 1, 0, 0, 0, &cst, 0, (opt->flag & MG_M_RMQ) ? 0 : lr_long_bases(), 0);
		if (rc < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, mga_last_error()); abort(); /* no CPU fallback */ }
		pthread_mutex_lock(&g_stats_mtx); stats_merge(&gi->B->st, &cst); pthread_mutex_unlock(&g_stats_mtx);
		return;
	}
	if (mg_map_batch(gi, 1, qlens, seqs, &qname, gcs, opt, 1) < 0) {
		fprintf(stderr, "[E::%s] %s\n", __func__, mga_last_error());
		abort(); /* no CPU fallback */
	}
}

mg_gchains_t *mg_map(const mg_idx_t *gi, int qlen, const char *seq, mg_tbuf_t *b, const mg_mapopt_t *opt, const char *qname)
{
	mg_gchains_t *gcs;
	mg_map_frag(gi, 1, &qlen, &seq, &gcs, b, opt, qname);
	return gcs;
}

void mga_get_stats(const mg_idx_t *gi, mga_stats_t *st, int reset)
{
	pthread_mutex_lock(&g_stats_mtx);
	*st = gi->B->st;
	if (reset) memset(&gi->B->st, 0, sizeof(mga_stats_t));
	pthread_mutex_unlock(&g_stats_mtx);
}
