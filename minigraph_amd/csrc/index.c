/*
 * index.c -- graph minimizer index: build (GPU sketch + host table construction) and its replica in HBM.
 *
 * Replaces mg_index()/mg_idx_destroy() (reference index.c:26-48,186-230).  The reference keeps 2^b
 * khashl buckets; what mapping observes of the index is only
 *    lookup(h) -> occurrence count and the ascending list of y = seg<<32 | lastPos<<1 | strand
 *    (mg_idx_get, index.c:50-72; lists sorted at index.c:156) and the occurrence quantiles used by
 *    mg_opt_update (mg_idx_cal_quantile, index.c:74-93).
 * Both are layout-independent, so the MI355X layout is ONE flat open-addressing table of 16-byte
 * slots {key|LIST, value} probed with a single 16-byte load per step, plus one contiguous position
 * array, sized for HBM (3 Gbp graph: ~5e8 minimizers -> 16 GB table + 4 GB positions).
 *
 * Build: all segments are sketched in one launch of the sketch kernel (one wavefront per segment),
 * then a stable LSD radix sort by hash on the host groups occurrences; groups are sorted by y.
 */
#include <stdio.h>
#include "mga_host.h"
#include "mga_idxhash.h"

static void radix_by_key(int64_t n, mg128_t *a, int key_bits) /* stable LSD radix sort on x>>8, 16-bit digits */
{
	mg128_t *b = MGA_MALLOC(mg128_t, n), *src = a, *dst = b, *t;
	int64_t *cnt = MGA_MALLOC(int64_t, 65536), i;
	int sh;
	for (sh = 0; sh < key_bits; sh += 16) {
		int64_t sum = 0, c;
		memset(cnt, 0, 65536 * sizeof(int64_t));
		for (i = 0; i < n; ++i) ++cnt[src[i].x >> (8 + sh) & 0xffff];
		for (i = 0; i < 65536; ++i) c = cnt[i], cnt[i] = sum, sum += c;
		for (i = 0; i < n; ++i) dst[cnt[src[i].x >> (8 + sh) & 0xffff]++] = src[i];
		t = src, src = dst, dst = t;
	}
	if (src != a) memcpy(a, src, (size_t)n * sizeof(mg128_t));
	free(b); free(cnt);
}

static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

/* the host-visible part of the index handle: parameters + both orientations of every segment
 * (gfa_edseq_init, gfa-ed.c:24-42).  No device work; mg_index() completes it with the minimizer table. */
static void upper_worker(void *data, int64_t s, int tid) /* uppercase in place, index.c:215-220 */
{
	gfa_seg_t *p = &((gfa_t*)data)->seg[s];
	int32_t q;
	(void)tid;
	if (p->seq) for (q = 0; q < p->len; ++q) { unsigned char c = (unsigned char)p->seq[q]; p->seq[q] = (char)(c - (((c >= 'a') & (c <= 'z')) << 5)); }
}

typedef struct { const gfa_t *g; gfa_edseq_t *es; } esw_t;
static void edseq_worker(void *data, int64_t s, int tid) /* both orientations of a segment, gfa_edseq_init (gfa-ed.c:24-42) */
{
	esw_t *w = (esw_t*)data;
	const gfa_seg_t *p = &w->g->seg[s];
	char *t = (char*)malloc((size_t)p->len + 1);
	int32_t q;
	(void)tid;
	for (q = 0; q < p->len; ++q) t[p->len - q - 1] = (char)mga_comp_table[(uint8_t)p->seq[q]];
	t[p->len] = 0;
	w->es[s<<1].seq = p->seq, w->es[s<<1|1].seq = t;
	w->es[s<<1].len = w->es[s<<1|1].len = p->len;
}

mg_idx_t *mga_idx_hostpart_mt(gfa_t *g, const mg_idxopt_t *io, int n_threads)
{
	mg_idx_t *gi = MGA_CALLOC(mg_idx_t, 1);
	gfa_edseq_t *es = MGA_MALLOC(gfa_edseq_t, (size_t)g->n_seg * 2 + 1);
	esw_t w;
	int k = io->k, wd = io->w, b = io->bucket_bits;
	mga_tables_init();
	if (k * 2 < b) b = k * 2; /* mg_idx_init, index.c:19-29 */
	if (wd < 1) wd = 1;
	mga_parallel_for(n_threads, g->n_seg, upper_worker, g);
	w.g = g, w.es = es;
	mga_parallel_for(n_threads, g->n_seg, edseq_worker, &w);
	gi->g = g, gi->w = wd, gi->k = k, gi->b = b, gi->n_seg = (int32_t)g->n_seg, gi->es = es, gi->B = 0;
	return gi;
}

mg_idx_t *mga_idx_hostpart(gfa_t *g, const mg_idxopt_t *io) { return mga_idx_hostpart_mt(g, io, 1); }

/* the same for a graph image: sequences are upper case already, the reverse complements go into ONE block (rc, segment s at off[s]) */
typedef struct { const gfa_t *g; gfa_edseq_t *es; const int64_t *off; char *rc; } esb_t;
static void edseq_blob_worker(void *data, int64_t s, int tid)
{
	esb_t *w = (esb_t*)data;
	const gfa_seg_t *p = &w->g->seg[s];
	char *t = w->rc + w->off[s];
	int32_t q;
	(void)tid;
	for (q = 0; q < p->len; ++q) t[p->len - q - 1] = (char)mga_comp_table[(uint8_t)p->seq[q]];
	w->es[s<<1].seq = p->seq, w->es[s<<1|1].seq = t;
	w->es[s<<1].len = w->es[s<<1|1].len = p->len;
}
mg_idx_t *mga_idx_hostpart_blob(gfa_t *g, const mg_idxopt_t *io, int n_threads, const int64_t *off, char *rc)
{
	mg_idx_t *gi = MGA_CALLOC(mg_idx_t, 1);
	gfa_edseq_t *es = MGA_MALLOC(gfa_edseq_t, (size_t)g->n_seg * 2 + 1);
	esb_t w;
	int k = io->k, wd = io->w, b = io->bucket_bits;
	mga_tables_init();
	if (k * 2 < b) b = k * 2;
	if (wd < 1) wd = 1;
	w.g = g, w.es = es, w.off = off, w.rc = rc;
	mga_parallel_for(n_threads, g->n_seg, edseq_blob_worker, &w);
	gi->g = g, gi->w = wd, gi->k = k, gi->b = b, gi->n_seg = (int32_t)g->n_seg, gi->es = es, gi->B = 0;
	return gi;
}

/* the index of a graph whose (upper-cased) forward sequences lie back to back in `cat` (segment s at off[s]): upload, device build of the minimizer table
 * (k_index.hip), graph replica, host handle.  es_rc != NULL: the reverse-complement images of the segments, same offsets, owned by the caller (graph image). */
mg_idx_t *mga_idx_from_cat(gfa_t *g, const mg_idxopt_t *io, int n_threads, const char *cat, const int64_t *off, const int32_t *seg_len, int64_t tot, char *es_rc)
{
	mga_sctx_t *sc = mga_sctx_default();
	struct mg_idx_bucket_s *B = MGA_CALLOC(struct mg_idx_bucket_s, 1);
	mg_idx_t *gi;
	int k = io->k, w = io->w < 1 ? 1 : io->w;
	const int dbg = getenv("MGA_DEBUG_INDEX") && atoi(getenv("MGA_DEBUG_INDEX")) > 0; /* phase times of the build to stderr */
	double t0 = mga_wtime(), t1;
#define IDX_T(what) do { if (dbg) { mga_dsync(); t1 = mga_wtime(); fprintf(stderr, "[index] %-28s %.3f s\n", what, t1 - t0); t0 = t1; } } while (0)
	B->dev.n_seg = (int32_t)g->n_seg;
	B->dev.d_seg_len = (int32_t*)mga_dmalloc((size_t)(g->n_seg + 1) * 4);
	B->dev.d_gseq = (char*)mga_dmalloc((size_t)tot + 64);
	B->dev.d_gseq_off = (int64_t*)mga_dmalloc((size_t)(g->n_seg + 1) * 8);
	IDX_T("device buffers");
	if (sc == 0 || !B->dev.d_seg_len || !B->dev.d_gseq || !B->dev.d_gseq_off ||
		mga_h2d(B->dev.d_seg_len, seg_len, (size_t)g->n_seg * 4) < 0 || mga_h2d_big(B->dev.d_gseq, cat, (size_t)tot, n_threads) < 0 || mga_dmemset((char*)B->dev.d_gseq + tot, 0, 64) < 0 ||
		mga_h2d(B->dev.d_gseq_off, off, (size_t)(g->n_seg + 1) * 8) < 0 || mga_dev_text_tables(mga_comp_table, mga_nt4_table) < 0) {
		mga_dfree(B->dev.d_seg_len); mga_dfree(B->dev.d_gseq); mga_dfree(B->dev.d_gseq_off);
		free(B);
		return 0;
	}
	IDX_T("sequence -> HBM");
	if (mga_dev_index_build(sc, (int)g->n_seg, B->dev.d_gseq, B->dev.d_gseq_off, w, k, &B->dev, &B->n_keys, &B->n_mz, &B->occ_hist, &B->max_occ_seen) < 0) {
		mga_dfree(B->dev.d_seg_len); mga_dfree(B->dev.d_gseq); mga_dfree(B->dev.d_gseq_off);
		free(B);
		return 0;
	}
	IDX_T("minimizer table (device)");
	if (mga_dev_graph_upload(sc, g, mga_comp_table, &B->dev) < 0) { /* arcs + reverse complements: graph chaining runs on the device (k_gchain.hip) */
		mga_dfree(B->dev.d_seg_len); mga_dfree(B->dev.d_gseq); mga_dfree(B->dev.d_gseq_off); mga_dfree(B->dev.d_tab); mga_dfree(B->dev.d_pos);
		mga_dfree(B->dev.d_arc); mga_dfree(B->dev.d_arc_idx); mga_dfree(B->dev.d_gseq_rc); mga_dfree(B->dev.d_gaf_seg); mga_dfree(B->dev.d_gaf_sseq); mga_dfree(B->dev.d_gaf_names); free(B->occ_hist); free(B);
		return 0;
	}
	IDX_T("graph replica");
	gi = es_rc ? mga_idx_hostpart_blob(g, io, n_threads, off, es_rc) : mga_idx_hostpart_mt(g, io, n_threads);
	gi->B = B;
	IDX_T("host reverse complements");
#undef IDX_T
	return gi;
}

mg_idx_t *mg_index(gfa_t *g, const mg_idxopt_t *io, int n_threads, mg_mapopt_t *mo)
{
	mg_idx_t *gi;
	struct mg_idx_bucket_s *B;
	int64_t tot = 0, n_mz, i, j, n_keys = 0, n_pos = 0, n_single = 0;
	int64_t *off, *mz_off = 0;
	uint32_t *rid, s;
	int32_t *seg_len;
	char *cat;
	mg128_t *mz = 0, *tab;
	uint64_t *pos, n_slots;
	int bits, k = io->k, w = io->w, b = io->bucket_bits;
	double t0 = mga_wtime();

	mga_tables_init();
	if (mga_dev_init() < 0) return 0;
	for (i = 0; i < (int64_t)g->n_arc; ++i) /* mg_gfa_overlap, index.c:177-184 */
		if (g->arc[i].ov != 0 || g->arc[i].ow != 0) {
			if (mg_verbose >= 1) fprintf(stderr, "[E::%s] minigraph doesn't work with graphs containing overlapping segments\n", __func__);
			return 0;
		}
	mga_parallel_for(n_threads, g->n_seg, upper_worker, g);
	if (k * 2 < b) b = k * 2; /* mg_idx_init, index.c:19-29 */
	if (w < 1) w = 1;

	/* sketch every segment on the GPU: rid = segment id (index.c:200-205) */
	off = MGA_MALLOC(int64_t, g->n_seg + 1);
	rid = MGA_MALLOC(uint32_t, g->n_seg + 1);
	seg_len = MGA_MALLOC(int32_t, g->n_seg + 1);
	for (s = 0; s < g->n_seg; ++s) {
		off[s] = tot, rid[s] = s, seg_len[s] = g->seg[s].len;
		if (g->seg[s].seq) tot += g->seg[s].len;
	}
	off[g->n_seg] = tot;
	cat = (char*)malloc((size_t)tot + 1);
	for (s = 0; s < g->n_seg; ++s) if (g->seg[s].seq) memcpy(cat + off[s], g->seg[s].seq, (size_t)g->seg[s].len);
	if (!(getenv("MGA_HOST_INDEX") && atoi(getenv("MGA_HOST_INDEX")) > 0)) { /* the whole build on the device (k_index.hip); MGA_HOST_INDEX=1 keeps the host build below (A/B) */
		free(rid);
		gi = mga_idx_from_cat(g, io, n_threads, cat, off, seg_len, tot, 0);
		free(off); free(seg_len); free(cat);
		if (gi == 0) return 0;
		if (mg_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f] indexed the graph on the device: %ld minimizers, %ld distinct, table 2^%d slots\n", __func__, mga_wtime() - t0, (long)gi->B->n_mz, (long)gi->B->n_keys, gi->B->dev.bits);
		if (mo) mg_opt_update(gi, mo, 0);
		return gi;
	}
	if (mga_sketch_batch((int)g->n_seg, cat, off, rid, w, k, &mz, &mz_off) < 0) { free(off); free(rid); free(seg_len); free(cat); return 0; }
	free(rid);
	n_mz = mz_off[g->n_seg];
	free(mz_off);

	/* group occurrences by hash; each group's positions ascending */
	radix_by_key(n_mz, mz, 2 * k);
	for (i = 0; i < n_mz; i = j) {
		for (j = i + 1; j < n_mz && mz[j].x >> 8 == mz[i].x >> 8; ++j) {}
		++n_keys;
		if (j - i > 1) n_pos += j - i; else ++n_single;
	}
	B = MGA_CALLOC(struct mg_idx_bucket_s, 1);
	B->n_keys = n_keys, B->n_mz = n_mz;
	for (bits = 10; (1LL << bits) < n_keys * 2; ++bits) {}
	n_slots = 1ULL << bits;
	tab = MGA_MALLOC(mg128_t, n_slots);
	for (i = 0; i < (int64_t)n_slots; ++i) tab[i].x = MGA_IDX_EMPTY, tab[i].y = 0;
	pos = MGA_MALLOC(uint64_t, n_pos + 1);
	n_pos = 0;
	for (i = 0; i < n_mz; i = j) {
		uint64_t key = mz[i].x >> 8, sl = mga_idx_slot(key, bits);
		int64_t c;
		for (j = i + 1; j < n_mz && mz[j].x >> 8 == key; ++j) {}
		c = j - i;
		while (tab[sl].x != MGA_IDX_EMPTY) sl = (sl + 1) & (n_slots - 1);
		if (c == 1) tab[sl].x = key, tab[sl].y = mz[i].y;
		else {
			int64_t q;
			for (q = 0; q < c; ++q) pos[n_pos + q] = mz[i + q].y;
			qsort(pos + n_pos, (size_t)c, 8, cmp_u64);
			tab[sl].x = key | MGA_IDX_LIST, tab[sl].y = (uint64_t)n_pos << 32 | (uint64_t)c;
			n_pos += c;
		}
		if (c > B->max_occ_seen) {
			B->occ_hist = MGA_REALLOC(int64_t, B->occ_hist, c + 1);
			memset(B->occ_hist + B->max_occ_seen + 1, 0, (size_t)(c - B->max_occ_seen) * sizeof(int64_t));
			if (B->max_occ_seen == 0) B->occ_hist[0] = 0;
			B->max_occ_seen = c;
		}
		++B->occ_hist[c];
	}
	free(mz);

	/* replica in HBM */
	B->dev.n_slots = n_slots, B->dev.bits = bits, B->dev.n_pos = n_pos, B->dev.n_seg = (int32_t)g->n_seg;
	B->dev.d_tab = (mg128_t*)mga_dmalloc((size_t)n_slots * 16);
	B->dev.d_pos = (uint64_t*)mga_dmalloc((size_t)(n_pos + 1) * 8);
	B->dev.d_seg_len = (int32_t*)mga_dmalloc((size_t)(g->n_seg + 1) * 4);
	B->dev.d_gseq = (char*)mga_dmalloc((size_t)tot + 64);       /* forward segment sequences: the text kernel reads target bases from here */
	B->dev.d_gseq_off = (int64_t*)mga_dmalloc((size_t)(g->n_seg + 1) * 8);
	if (!B->dev.d_tab || !B->dev.d_pos || !B->dev.d_seg_len || !B->dev.d_gseq || !B->dev.d_gseq_off ||
		mga_h2d(B->dev.d_tab, tab, (size_t)n_slots * 16) < 0 || mga_h2d(B->dev.d_pos, pos, (size_t)n_pos * 8) < 0 ||
		mga_h2d(B->dev.d_seg_len, seg_len, (size_t)g->n_seg * 4) < 0 || mga_h2d(B->dev.d_gseq, cat, (size_t)tot) < 0 ||
		mga_h2d(B->dev.d_gseq_off, off, (size_t)(g->n_seg + 1) * 8) < 0 || mga_dev_text_tables(mga_comp_table, mga_nt4_table) < 0) {
		free(tab); free(pos); free(seg_len); free(cat); free(off); free(B->occ_hist); free(B);
		return 0;
	}
	free(tab); free(pos); free(seg_len); free(cat); free(off);

	if (mga_dev_graph_upload(mga_sctx_default(), g, mga_comp_table, &B->dev) < 0) { free(B->occ_hist); free(B); return 0; }
	gi = mga_idx_hostpart(g, io);
	gi->B = B;
	if (mg_verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] indexed the graph: %ld minimizers, %ld distinct, table 2^%d slots\n", __func__, mga_wtime() - t0, (long)n_mz, (long)n_keys, bits);
	(void)n_threads; (void)n_single;
	if (mo) mg_opt_update(gi, mo, 0);
	return gi;
}

void mg_idx_destroy(mg_idx_t *gi)
{
	int32_t i;
	if (gi == 0) return;
	if (gi->B) {
		mga_idx_mf_free(gi);
		mga_idx_stream_close(gi); /* pipeline threads, HIP streams and buffers of the single-batch entry points */
		mga_dfree(gi->B->dev.d_tab); mga_dfree(gi->B->dev.d_pos); mga_dfree(gi->B->dev.d_seg_len); mga_dfree(gi->B->dev.d_gseq); mga_dfree(gi->B->dev.d_gseq_off);
		mga_dfree(gi->B->dev.d_arc); mga_dfree(gi->B->dev.d_arc_idx); mga_dfree(gi->B->dev.d_gseq_rc);
		mga_dfree(gi->B->dev.d_gaf_seg); mga_dfree(gi->B->dev.d_gaf_sseq); mga_dfree(gi->B->dev.d_gaf_names);
		free(gi->B->occ_hist); free(gi->B->gaf_out);
	}
	if (gi->es) {
		if (!(gi->B && gi->B->img_rc)) for (i = 0; i < gi->n_seg; ++i) free((char*)gi->es[i<<1|1].seq);
		free(gi->es);
	}
	if (gi->B && gi->B->img_g) mga_graph_image_release(gi->B); /* an index loaded from a graph image owns its gfa_t, the mapped file and the reverse-complement block */
	free(gi->B);
	free(gi);
}

/* mg_idx_cal_quantile (index.c:74-93): the kk-th smallest occurrence count over distinct minimizers,
 * kk = (size_t)((1.0 - (double)f) * n).  Selection by value needs only the histogram. */
void mga_idx_cal_quantile(const mg_idx_t *gi, int32_t m, const float f[], int32_t q[])
{
	const struct mg_idx_bucket_s *B = gi->B;
	int32_t i;
	for (i = 0; i < m; ++i) {
		size_t kk = (size_t)((1.0 - (double)f[i]) * B->n_keys);
		int64_t c, acc = 0;
		q[i] = 0;
		for (c = 1; c <= B->max_occ_seen; ++c) {
			acc += B->occ_hist[c];
			if ((int64_t)kk < acc) { q[i] = (int32_t)c; break; }
		}
		if (c > B->max_occ_seen) q[i] = (int32_t)B->max_occ_seen;
	}
}
