/*
 * image.c -- the graph as ONE binary image: what gfa_read + gfa_finalize + the upper-casing and concatenation of mg_index leave --
 * segments, stable sequences, names, arcs in their final order, the arc index, all sequence back to back -- written once and then mapped instead of
 * parsed.  Replaces, for repeated runs and for the N ranks of a node, gfa_read (gfa-io.c:294), gfa_finalize (gfa-base.c:421-430), gfa_edseq_init
 * (gfa-ed.c:24-42) and the host half of mg_index (index.c:186-230); mg_index() itself is unchanged.
 *
 * The minimizer table is NOT in the image: the device builds it from the sequence in ~0.2 s for a 3 Gbp graph (k_index.hip), which is less than
 * reading its 20 GB back would take, and it depends on (k, w), which stay load-time options.  [measured, 3.02 Gbp graph in 831 k segments] text GFA
 * -> index 3.3-6 s per process (parse, upper-case, concatenate, upload, build); image -> index: see DESIGN.md.
 *
 * Layout (little endian, sections 64-byte aligned, offsets from the file start):
 *   header | seg records | segment names | stable-sequence records | their names | arcs (gfa_arc_t) | arc index (uint64 per vertex) |
 *   sequence offsets (int64, n_seg + 1) | sequence bytes (upper case, back to back)
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <pthread.h>
#include "mga_host.h"

#define IMG_MAGIC "MGAGIMG1"

typedef struct {
	char magic[8];
	uint32_t version, pad;
	uint64_t n_seg, n_sseq, n_arc, max_rank, tot_seq;
	uint64_t off_seg, off_names, names_bytes, off_sseq, off_snames, snames_bytes, off_arc, off_idx, off_seqoff, off_seq, file_bytes;
} img_hdr_t;
typedef struct { int32_t len, snid, soff, rank; uint32_t del_circ, name_off; } img_seg_t;
typedef struct { int32_t min, max, rank; uint32_t name_off; } img_sseq_t;

static uint64_t al64(uint64_t x) { return (x + 63) & ~(uint64_t)63; }

static int put(FILE *fp, uint64_t *pos, const void *p, size_t n) /* write n bytes, then pad to the next 64-byte boundary */
{
	static const char zero[64] = { 0 };
	const uint64_t end = al64(*pos + n);
	if (n && fwrite(p, 1, n, fp) != n) return -1;
	if (end > *pos + n && fwrite(zero, 1, (size_t)(end - *pos - n), fp) != end - *pos - n) return -1;
	*pos = end;
	return 0;
}

/* g: a graph as gfa_read() returns it; sequences are written upper-cased whether or not mg_index() has run on it */
int mga_graph_image_save(const gfa_t *g, const char *path)
{
	img_hdr_t h;
	img_seg_t *sr = MGA_MALLOC(img_seg_t, g->n_seg + 1);
	img_sseq_t *qr = MGA_MALLOC(img_sseq_t, g->n_sseq + 1);
	int64_t *off = MGA_MALLOC(int64_t, g->n_seg + 1);
	char *names, *snames, *buf = 0;
	size_t nb = 0, sb = 0, m_buf = 0;
	uint64_t pos = 0;
	uint32_t s;
	FILE *fp = fopen(path, "wb");
	int rc = -1;
	if (fp == 0) { mga_set_error("graph image: cannot write %s", path); free(sr); free(qr); free(off); return -1; }
	memset(&h, 0, sizeof h);
	memcpy(h.magic, IMG_MAGIC, 8);
	h.version = 1, h.n_seg = g->n_seg, h.n_sseq = g->n_sseq, h.n_arc = g->n_arc, h.max_rank = g->max_rank;
	for (s = 0; s < g->n_seg; ++s) nb += strlen(g->seg[s].name) + 1;
	for (s = 0; s < g->n_sseq; ++s) sb += strlen(g->sseq[s].name) + 1;
	names = (char*)malloc(nb + 1), snames = (char*)malloc(sb + 1);
	if (sr == 0 || qr == 0 || off == 0 || names == 0 || snames == 0) { mga_set_error("graph image: out of memory"); fclose(fp); free(sr); free(qr); free(off); free(names); free(snames); return -1; }
	for (s = 0, nb = 0; s < g->n_seg; ++s) {
		const gfa_seg_t *p = &g->seg[s];
		const size_t l = strlen(p->name) + 1;
		sr[s].len = p->len, sr[s].snid = p->snid, sr[s].soff = p->soff, sr[s].rank = p->rank, sr[s].del_circ = (uint32_t)p->del | (uint32_t)p->circ << 16, sr[s].name_off = (uint32_t)nb;
		memcpy(names + nb, p->name, l); nb += l;
		off[s] = (int64_t)h.tot_seq;
		if (p->seq == 0 && p->len > 0) { mga_set_error("graph image: segment %s has no sequence", p->name); fclose(fp); free(sr); free(qr); free(off); free(names); free(snames); return -1; }
		h.tot_seq += (uint64_t)p->len;
	}
	off[g->n_seg] = (int64_t)h.tot_seq;
	for (s = 0, sb = 0; s < g->n_sseq; ++s) {
		const size_t l = strlen(g->sseq[s].name) + 1;
		qr[s].min = g->sseq[s].min, qr[s].max = g->sseq[s].max, qr[s].rank = g->sseq[s].rank, qr[s].name_off = (uint32_t)sb;
		memcpy(snames + sb, g->sseq[s].name, l); sb += l;
	}
	h.names_bytes = nb, h.snames_bytes = sb;
	pos = al64(sizeof h);
	h.off_seg = pos; pos = al64(pos + (uint64_t)g->n_seg * sizeof(img_seg_t));
	h.off_names = pos; pos = al64(pos + nb);
	h.off_sseq = pos; pos = al64(pos + (uint64_t)g->n_sseq * sizeof(img_sseq_t));
	h.off_snames = pos; pos = al64(pos + sb);
	h.off_arc = pos; pos = al64(pos + g->n_arc * sizeof(gfa_arc_t));
	h.off_idx = pos; pos = al64(pos + (uint64_t)g->n_seg * 2 * 8);
	h.off_seqoff = pos; pos = al64(pos + ((uint64_t)g->n_seg + 1) * 8);
	h.off_seq = pos; pos = al64(pos + h.tot_seq + 64);
	h.file_bytes = pos;
	pos = 0;
	if (put(fp, &pos, &h, sizeof h) < 0 || put(fp, &pos, sr, (size_t)g->n_seg * sizeof(img_seg_t)) < 0 || put(fp, &pos, names, nb) < 0 ||
		put(fp, &pos, qr, (size_t)g->n_sseq * sizeof(img_sseq_t)) < 0 || put(fp, &pos, snames, sb) < 0 || put(fp, &pos, g->arc, (size_t)g->n_arc * sizeof(gfa_arc_t)) < 0 ||
		put(fp, &pos, g->idx, (size_t)g->n_seg * 2 * 8) < 0 || put(fp, &pos, off, ((size_t)g->n_seg + 1) * 8) < 0) goto done;
	for (s = 0; s < g->n_seg; ++s) { /* sequence bytes, upper case (index.c:215-220), through a block buffer */
		const gfa_seg_t *p = &g->seg[s];
		int32_t q;
		if (p->seq == 0 || p->len == 0) continue;
		if ((size_t)p->len > m_buf) { char *nbuf; m_buf = (size_t)p->len + ((size_t)p->len >> 1) + 4096; nbuf = (char*)realloc(buf, m_buf); if (nbuf == 0) goto done; buf = nbuf; }
		for (q = 0; q < p->len; ++q) { const unsigned char c = (unsigned char)p->seq[q]; buf[q] = (char)(c - (((c >= 'a') & (c <= 'z')) << 5)); }
		if (fwrite(buf, 1, (size_t)p->len, fp) != (size_t)p->len) goto done;
	}
	pos = h.off_seq + h.tot_seq;
	{ char z[128] = { 0 }; const uint64_t end = h.file_bytes; while (pos < end) { const size_t n = end - pos < 128 ? (size_t)(end - pos) : 128; if (fwrite(z, 1, n, fp) != n) goto done; pos += n; } }
	rc = 0;
done:
	if (fclose(fp) != 0) rc = -1;
	if (rc < 0) mga_set_error("graph image: write to %s failed", path);
	free(sr); free(qr); free(off); free(names); free(snames); free(buf);
	return rc;
}

/* ---- pageable host memory -> HBM: threads copy blocks into pinned staging buffers, the copy engine drains them (a plain hipMemcpy from a mapped file
 * runs at the speed of ONE thread's page-cache copy) ---- */
typedef struct { char *d; const char *h; size_t bytes, blk; int n_blk; volatile int next; int rc; pthread_mutex_t m; } h2d_big_t;
static void *h2d_big_worker(void *a)
{
	h2d_big_t *w = (h2d_big_t*)a;
	char *stage;
	if (mga_dev_bind_thread() < 0 || (stage = (char*)mga_hmalloc_pinned(w->blk)) == 0) { w->rc = -1; return 0; }
	for (;;) { /* fill the pinned block (page-cache copy, this thread's share of the host bandwidth), hand it to the copy engine; the other threads fill theirs meanwhile */
		int b;
		size_t o, n;
		pthread_mutex_lock(&w->m); b = w->next++; pthread_mutex_unlock(&w->m);
		if (b >= w->n_blk) break;
		o = (size_t)b * w->blk, n = w->bytes - o < w->blk ? w->bytes - o : w->blk;
		memcpy(stage, w->h + o, n);
		if (mga_h2d(w->d + o, stage, n) < 0) { w->rc = -1; break; }
	}
	mga_hfree_pinned(stage);
	return 0;
}
int mga_h2d_big(void *d, const void *h, size_t bytes, int n_threads)
{
	h2d_big_t w;
	pthread_t thr[8];
	int i, T = n_threads < 1 ? 1 : n_threads > 8 ? 8 : n_threads;
	if (bytes < ((size_t)64 << 20) || T == 1) return mga_h2d(d, h, bytes);
	w.d = (char*)d, w.h = (const char*)h, w.bytes = bytes, w.blk = (size_t)16 << 20, w.n_blk = (int)((bytes + w.blk - 1) / w.blk), w.next = 0, w.rc = 0;
	pthread_mutex_init(&w.m, 0);
	for (i = 0; i < T; ++i) pthread_create(&thr[i], 0, h2d_big_worker, &w);
	for (i = 0; i < T; ++i) pthread_join(thr[i], 0);
	pthread_mutex_destroy(&w.m);
	if (w.rc < 0) mga_set_error("upload of %zu bytes through staging buffers failed", bytes);
	return w.rc;
}

/* ---- load ---- */
/* A mapped file is untrusted input: every section must lie inside the file (overflow-safe: count * size is checked by division), the sequence offsets must ascend from 0 to
 * tot_seq in steps of the segments' lengths, names must start inside their blocks and the blocks must end with a NUL, arcs must name vertices of the graph and the arc index
 * must stay inside the arc array.  Returns 0 when the image can be used as it is. */
static int sect_ok(uint64_t off, uint64_t count, uint64_t size, uint64_t file) { return off <= file && (size == 0 || count <= (file - off) / size); }
static int image_valid(const char *base, uint64_t file)
{
	const img_hdr_t *h = (const img_hdr_t*)base;
	const img_seg_t *sr;
	const img_sseq_t *qr;
	const int64_t *off;
	const gfa_arc_t *arc;
	const uint64_t *idx;
	uint64_t s, n_vtx;
	if (memcmp(h->magic, IMG_MAGIC, 8) != 0 || h->version != 1) return -1;
	if (h->file_bytes > file) return -2;
	if (h->n_seg >= 0x7fffffffULL || h->n_sseq > 0xffffffffULL || h->max_rank > 0xffffffffULL) return -2;
	if (!sect_ok(h->off_seg, h->n_seg, sizeof(img_seg_t), file) || !sect_ok(h->off_names, h->names_bytes, 1, file) || !sect_ok(h->off_sseq, h->n_sseq, sizeof(img_sseq_t), file) ||
		!sect_ok(h->off_snames, h->snames_bytes, 1, file) || !sect_ok(h->off_arc, h->n_arc, sizeof(gfa_arc_t), file) || !sect_ok(h->off_idx, h->n_seg * 2, 8, file) ||
		!sect_ok(h->off_seqoff, h->n_seg + 1, 8, file) || !sect_ok(h->off_seq, h->tot_seq, 1, file) || file - h->off_seq - h->tot_seq < 64) return -3; /* (64 readable bytes behind the sequence) */
	if ((h->off_seg | h->off_sseq | h->off_arc | h->off_idx | h->off_seqoff) & 7) return -4;
	if (h->n_seg > 0 && (h->names_bytes == 0 || base[h->off_names + h->names_bytes - 1] != 0)) return -5;
	if (h->n_sseq > 0 && (h->snames_bytes == 0 || base[h->off_snames + h->snames_bytes - 1] != 0)) return -5;
	sr = (const img_seg_t*)(base + h->off_seg), qr = (const img_sseq_t*)(base + h->off_sseq), off = (const int64_t*)(base + h->off_seqoff);
	arc = (const gfa_arc_t*)(base + h->off_arc), idx = (const uint64_t*)(base + h->off_idx);
	if (off[0] != 0 || (uint64_t)off[h->n_seg] != h->tot_seq) return -6;
	for (s = 0; s < h->n_seg; ++s) {
		if (sr[s].len < 0 || off[s + 1] < off[s] || off[s + 1] - off[s] != (int64_t)sr[s].len) return -6; /* (a segment without sequence has length 0 in an image: save writes what it has) */
		if (sr[s].name_off >= h->names_bytes) return -7;
		if (sr[s].snid >= 0 && (uint64_t)sr[s].snid >= h->n_sseq) return -7;
	}
	for (s = 0; s < h->n_sseq; ++s) if (qr[s].name_off >= h->snames_bytes) return -7;
	n_vtx = h->n_seg * 2;
	for (s = 0; s < h->n_arc; ++s) if ((arc[s].v_lv >> 32) >= n_vtx || arc[s].w >= n_vtx) return -8;
	for (s = 0; s < n_vtx; ++s) { const uint64_t st = idx[s] >> 32, n = (uint32_t)idx[s]; if (st > h->n_arc || n > h->n_arc - st) return -9; }
	return 0;
}

void mga_graph_image_release(struct mg_idx_bucket_s *B)
{
	gfa_t *g = B->img_g;
	uint32_t s;
	if (g) {
		for (s = 0; s < g->n_seg; ++s) g->seg[s].seq = 0, g->seg[s].name = 0; /* they point into the mapped file */
		for (s = 0; s < g->n_sseq; ++s) g->sseq[s].name = 0;
		g->arc = 0, g->idx = 0; /* (mapped too) */
		gfa_destroy(g);
	}
	if (B->img_map) munmap(B->img_map, B->img_map_bytes);
	free(B->img_rc); free(B->img_off);
	B->img_g = 0, B->img_map = 0, B->img_rc = 0, B->img_off = 0;
}

/* the index of the graph in `path` (a file written by mga_graph_image_save): the file is mapped, nothing is parsed or copied on the host, the sequence goes to
 * HBM and the device builds the minimizer table.  The returned index OWNS its graph (gi->g): mg_idx_destroy() releases everything; do not gfa_destroy() it. */
mg_idx_t *mga_index_load_image(const char *path, const mg_idxopt_t *io, int n_threads, mg_mapopt_t *mo)
{
	int fd = open(path, O_RDONLY);
	struct stat st;
	const img_hdr_t *h;
	char *base;
	gfa_t *g;
	mg_idx_t *gi;
	const img_seg_t *sr;
	const img_sseq_t *qr;
	const int64_t *off;
	int32_t *seg_len;
	char *rc;
	uint32_t s;
	double t0 = mga_wtime();
	if (n_threads < 1) n_threads = 1;
	mga_tables_init();
	if (fd < 0 || fstat(fd, &st) < 0 || (size_t)st.st_size < sizeof(img_hdr_t)) { mga_set_error("graph image: cannot open %s", path); if (fd >= 0) close(fd); return 0; }
	base = (char*)mmap(0, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0); /* private: nothing is written, but gfa_seg_t::seq is a char* */
	close(fd);
	if (base == MAP_FAILED) { mga_set_error("graph image: cannot map %s", path); return 0; }
	h = (const img_hdr_t*)base;
	{
		const int why = image_valid(base, (uint64_t)st.st_size);
		if (why != 0) {
			if (why == -1) mga_set_error("graph image: %s is not a graph image of this version", path);
			else mga_set_error("graph image: %s is truncated or corrupt (check %d)", path, -why);
			munmap(base, (size_t)st.st_size); return 0;
		}
	}
	if (mga_dev_init() < 0) { munmap(base, (size_t)st.st_size); return 0; } /* (after the file checks: a corrupt image is reported as such with or without a GPU) */
	(void)madvise(base, (size_t)st.st_size, MADV_WILLNEED);
	sr = (const img_seg_t*)(base + h->off_seg), qr = (const img_sseq_t*)(base + h->off_sseq), off = (const int64_t*)(base + h->off_seqoff);
	g = MGA_CALLOC(gfa_t, 1);
	g->n_seg = g->m_seg = (uint32_t)h->n_seg, g->n_sseq = g->m_sseq = (uint32_t)h->n_sseq, g->n_arc = g->m_arc = h->n_arc, g->max_rank = (uint32_t)h->max_rank;
	g->seg = MGA_CALLOC(gfa_seg_t, g->n_seg + 1), g->sseq = MGA_CALLOC(gfa_sseq_t, g->n_sseq + 1);
	g->arc = (gfa_arc_t*)(base + h->off_arc), g->idx = (uint64_t*)(base + h->off_idx);
	seg_len = MGA_MALLOC(int32_t, g->n_seg + 1);
	for (s = 0; s < g->n_seg; ++s) {
		gfa_seg_t *p = &g->seg[s];
		p->len = sr[s].len, p->snid = sr[s].snid, p->soff = sr[s].soff, p->rank = sr[s].rank, p->del = sr[s].del_circ & 0xffff, p->circ = sr[s].del_circ >> 16;
		p->name = base + h->off_names + sr[s].name_off, p->seq = base + h->off_seq + off[s];
		seg_len[s] = p->len;
	}
	for (s = 0; s < g->n_sseq; ++s) g->sseq[s].name = base + h->off_snames + qr[s].name_off, g->sseq[s].min = qr[s].min, g->sseq[s].max = qr[s].max, g->sseq[s].rank = qr[s].rank;
	for (s = 0; s < g->n_arc; ++s) if (g->arc[s].ov != 0 || g->arc[s].ow != 0) break; /* mg_gfa_overlap, index.c:177-184 */
	rc = (char*)malloc((size_t)h->tot_seq + 1);
	if (s < g->n_arc || rc == 0) {
		if (s < g->n_arc && mg_verbose >= 1) fprintf(stderr, "[E::%s] minigraph doesn't work with graphs containing overlapping segments\n", __func__);
		gi = 0;
	} else gi = mga_idx_from_cat(g, io, n_threads, base + h->off_seq, off, seg_len, (int64_t)h->tot_seq, rc);
	free(seg_len);
	if (gi == 0) {
		struct mg_idx_bucket_s tmp;
		memset(&tmp, 0, sizeof tmp);
		tmp.img_g = g, tmp.img_map = base, tmp.img_map_bytes = (size_t)st.st_size, tmp.img_rc = rc;
		mga_graph_image_release(&tmp);
		return 0;
	}
	gi->B->img_g = g, gi->B->img_map = base, gi->B->img_map_bytes = (size_t)st.st_size, gi->B->img_rc = rc;
	if (mg_verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] graph image %s: %u segments, %lu bp; %ld minimizers, %ld distinct\n", __func__, mga_wtime() - t0, path, g->n_seg, (unsigned long)h->tot_seq,
				(long)gi->B->n_mz, (long)gi->B->n_keys);
	if (mo) mg_opt_update(gi, mo, 0);
	return gi;
}
