/*
 * mapfiles.c -- file-level driver: mg_map_files() (reference gmap.c:163-211) on top of the chunk pipeline (mapper.c: mga_stream_*).
 * Reads FASTA/FASTQ (plain or gzip), upper-cases and converts U->T like the reference (gmap.c:81, bseq.c:50-58), maps
 * mini-batches of opt->mini_batch_size bases and writes GAF in input order.  Two readers with kseq's record semantics
 * (bseq.c:61-98 via kseq.h): a sequential one for anything gzopen() can read, and a parallel one for plain FASTA files
 * (memory-mapped, split at "\n>", parsed straight into pinned memory that the chunks are uploaded from).
 */
#include <zlib.h>
#include <stdio.h>
#include <ctype.h>
#include "mga_host.h"

#define RD_BUF (4 << 20)
typedef struct { gzFile fp; char *buf; int64_t beg, end; int eof; } rd_t; /* fp == NULL: buf[0..end) is the whole input (a memory-mapped file) */

static int rd_fill(rd_t *r) /* 1 if the buffer holds data */
{
	if (r->beg < r->end) return 1;
	if (r->eof || r->fp == 0) return 0;
	r->end = gzread(r->fp, r->buf, RD_BUF), r->beg = 0;
	if (r->end <= 0) { r->eof = 1, r->end = 0; return 0; }
	return 1;
}
static inline int rd_getc(rd_t *r) { return rd_fill(r) ? (unsigned char)r->buf[r->beg++] : -1; }

typedef struct { char *s; size_t l, m; } str_t;
static inline void str_room(str_t *s, size_t extra) { if (s->l + extra + 2 > s->m) { s->m = (s->l + extra + 2) * 2; if (s->m < 256) s->m = 256; s->s = (char*)realloc(s->s, s->m); } }
static inline void str_c(str_t *s, int c) { str_room(s, 1); s->s[s->l++] = (char)c; s->s[s->l] = 0; }

/* the rest of the current line: appended to s when s != NULL, the '\n' is consumed.  A trailing '\r' is dropped the way kseq does it (kseq.h ks_getuntil2:
 * "str->l > 1 && last == '\r'", with str->l counted from the start of the record's field: l0) */
static void rd_line_from(rd_t *r, str_t *s, size_t l0)
{
	while (rd_fill(r)) {
		char *p = r->buf + r->beg, *q = (char*)memchr(p, '\n', (size_t)(r->end - r->beg));
		const size_t n = q ? (size_t)(q - p) : (size_t)(r->end - r->beg);
		if (s) { str_room(s, n); memcpy(s->s + s->l, p, n); s->l += n; s->s[s->l] = 0; }
		r->beg += (int64_t)n + (q ? 1 : 0);
		if (q) break;
	}
	if (s && s->l - l0 > 1 && s->s[s->l - 1] == '\r') s->s[--s->l] = 0;
}
static void rd_line(rd_t *r, str_t *s) { rd_line_from(r, s, 0); }

/* one FASTA/FASTQ record (kseq semantics, bseq.c:61-98 via kseq.h); returns 0, or -1 at EOF.  `last` carries a record
 * marker that was read as the first character of a line between calls.  Lines are moved with memchr/memcpy. */
static int read_record_x(rd_t *r, int *last, str_t *name, str_t *seq, int append) /* append: the bases go behind what seq already holds (the caller's slab) */
{
	int c;
	if (*last == 0) { /* find the next record: the next '>' or '@' WHEREVER it stands, as kseq does (kseq.h:180-184; ADVICE r2: a marker behind garbage on the same line) */
		while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') {}
		if (c < 0) return -1;
		*last = c;
	}
	const size_t seq0 = append ? seq->l : 0;
	name->l = 0, seq->l = seq0;
	str_room(name, 1); str_room(seq, 1);
	name->s[0] = seq->s[seq0] = 0;
	{ /* header line: the name ends at the first white space, the comment is dropped */
		size_t k;
		rd_line(r, name);
		for (k = 0; k < name->l; ++k) if (isspace((unsigned char)name->s[k])) break;
		name->l = k, name->s[k] = 0;
	}
	while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		str_c(seq, c);
		rd_line_from(r, seq, seq0);
	}
	if (c == '>' || c == '@') *last = c;
	else *last = 0;
	if (c == '+') { /* FASTQ: skip the '+' line and as many quality characters as bases */
		size_t ql = 0;
		int tail = 0; /* last character of the current quality line so far (a line may arrive in several buffer fills) */
		rd_line(r, 0);
		while (ql < seq->l - seq0 && rd_fill(r)) {
			char *p = r->buf + r->beg, *q = (char*)memchr(p, '\n', (size_t)(r->end - r->beg));
			size_t n = q ? (size_t)(q - p) : (size_t)(r->end - r->beg);
			r->beg += (int64_t)n + (q ? 1 : 0);
			if (n > 0) tail = p[n - 1];
			ql += n;
			if (q) { if (ql > 1 && tail == '\r') --ql; tail = 0; } /* (the running length decides about a '\r', as for the bases) */
		}
		*last = 0;
		if (ql != seq->l - seq0) return -1; /* a quality string of another length ends the file: kseq_read returns -2 and bseq.c:66 stops reading */
	}
	return 0;
}
static int read_record(rd_t *r, int *last, str_t *name, str_t *seq) { return read_record_x(r, last, name, seq, 0); }

/* upper case + U -> T, as the reference does per base (gmap.c:81, bseq.c:50-58); branch-free so that it vectorises */
__attribute__((optimize("O3"))) static void seq_normalize(char *s, size_t l)
{
	size_t k;
	for (k = 0; k < l; ++k) {
		unsigned char c = (unsigned char)s[k];
		c = (unsigned char)(c - ((c == 'u') | (c == 'U')));
		c = (unsigned char)(c - (((c >= 'a') & (c <= 'z')) << 5));
		s[k] = (char)c;
	}
}

/* the same while copying: one pass over the bases instead of memcpy + seq_normalize */
__attribute__((optimize("O3"))) static void seq_copy_normalize(char *__restrict__ d, const char *__restrict__ s, size_t l)
{
	size_t k;
	for (k = 0; k < l; ++k) {
		unsigned char c = (unsigned char)s[k];
		c = (unsigned char)(c - ((c == 'u') | (c == 'U')));
		c = (unsigned char)(c - (((c >= 'a') & (c <= 'z')) << 5));
		d[k] = (char)c;
	}
}

/* ---- a read set kept resident: host copy + HBM copy (bench.py, repeated passes over one batch) ---- */
struct mga_reads_s {
	int n;
	int *qlens;
	char **seqs, **names;
	int64_t *q_off, n_bases;
	char *d_seq;
};

mga_reads_t *mga_reads_load(const char *fn, int64_t max_reads)
{
	rd_t r;
	int last = 0, m = 0;
	str_t name = {0, 0, 0}, seq = {0, 0, 0};
	mga_reads_t *rd;
	char *h;
	int i;
	if (mga_dev_init() < 0) return 0;
	memset(&r, 0, sizeof r);
	r.fp = gzopen(fn, "r");
	if (r.fp == 0) { mga_set_error("cannot open '%s'", fn); return 0; }
	r.buf = (char*)malloc(RD_BUF);
	rd = MGA_CALLOC(mga_reads_t, 1);
	while ((max_reads <= 0 || rd->n < max_reads) && read_record(&r, &last, &name, &seq) == 0) {
		if (rd->n == m) { m = m ? m << 1 : 256; rd->qlens = MGA_REALLOC(int, rd->qlens, m); rd->seqs = MGA_REALLOC(char*, rd->seqs, m); rd->names = MGA_REALLOC(char*, rd->names, m); }
		seq_normalize(seq.s, seq.l);
		rd->seqs[rd->n] = (char*)malloc(seq.l + 1); memcpy(rd->seqs[rd->n], seq.s, seq.l + 1);
		rd->names[rd->n] = (char*)malloc(name.l + 1); memcpy(rd->names[rd->n], name.s, name.l + 1);
		rd->qlens[rd->n++] = (int)seq.l;
		rd->n_bases += (int64_t)seq.l;
	}
	free(name.s); free(seq.s); free(r.buf); gzclose(r.fp);
	rd->q_off = MGA_MALLOC(int64_t, rd->n + 1);
	h = (char*)calloc((size_t)rd->n_bases + 64, 1);
	for (i = 0, rd->n_bases = 0; i < rd->n; ++i) { rd->q_off[i] = rd->n_bases; memcpy(h + rd->n_bases, rd->seqs[i], (size_t)rd->qlens[i]); rd->n_bases += rd->qlens[i]; }
	rd->q_off[rd->n] = rd->n_bases;
	rd->d_seq = (char*)mga_dmalloc((size_t)rd->n_bases + 64);
	if (rd->d_seq == 0 || mga_h2d(rd->d_seq, h, (size_t)rd->n_bases + 64) < 0) { free(h); mga_reads_free(rd); return 0; }
	free(h);
	return rd;
}

void mga_reads_free(mga_reads_t *rd)
{
	int i;
	if (rd == 0) return;
	for (i = 0; i < rd->n; ++i) { free(rd->seqs[i]); free(rd->names[i]); }
	free(rd->seqs); free(rd->names); free(rd->qlens); free(rd->q_off);
	mga_dfree(rd->d_seq);
	free(rd);
}

int mga_reads_count(const mga_reads_t *rd) { return rd->n; }
int64_t mga_reads_bases(const mga_reads_t *rd) { return rd->n_bases; }

int mga_map_gaf(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, const mg_mapopt_t *opt, int n_threads,
				const char *d_seq, const int64_t *q_off, char **gaf, int64_t *gaf_len);

/* map a resident read set and format its GAF (input order) into the index-owned output buffer; formatting runs inside the
 * mapping pipeline, chunk by chunk */
int mga_map_reads(const mg_idx_t *gi, const mga_reads_t *rd, const mg_mapopt_t *opt, int n_threads, char **gaf, int64_t *gaf_len)
{
	if (n_threads < 1) n_threads = 1;
	if (getenv("MGA_UPLOAD_READS") && atoi(getenv("MGA_UPLOAD_READS")) > 0) /* measurement aid: ignore the resident copy, upload the reads chunk by chunk (the PCIe-inclusive rate) */
		return mga_map_gaf(gi, rd->n, rd->qlens, (const char**)rd->seqs, (const char**)rd->names, opt, n_threads, 0, 0, gaf, gaf_len);
	return mga_map_gaf(gi, rd->n, rd->qlens, (const char**)rd->seqs, (const char**)rd->names, opt, n_threads, rd->d_seq, rd->q_off, gaf, gaf_len);
}

/* ---- mg_map_files (gmap.c:163-211): read -> map -> write as overlapped stages, like the reference's kt_pipeline (gmap.c:70-141).
 * A reader thread parses mini-batches of opt->mini_batch_size bases; the calling thread submits them to ONE chunk pipeline that
 * lives for the whole job (mapper.c: mga_stream_*; chunks of consecutive mini-batches follow each other without a drain); a
 * collector thread takes the finished batches in input order and hands their GAF text to the writer thread. ---- */
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

typedef struct {
	int n, m, last, pinned;
	int seg;                 /* output segment this batch belongs to (sharded jobs: rank-order concatenation happens per segment) */
	int *qlens;
	char **seqs, **names;
	size_t *seq_off, *name_off;
	str_t slab;              /* sequential reader: the bases of every record, back to back */
	char *base; size_t base_cap; int base_pinned; /* parallel reader: the same (+64 bytes of slack), in pinned host memory when a device is present: uploaded without a staging copy */
	str_t nslab;             /* "name\0" of every record */
} fbatch_t;

#define CHAN_CAP 8
typedef struct { pthread_mutex_t m; pthread_cond_t c; void *item[CHAN_CAP]; int n, cap, closed; } chan_t; /* small bounded queue */
static void chan_init(chan_t *c, int cap) { pthread_mutex_init(&c->m, 0); pthread_cond_init(&c->c, 0); c->n = 0, c->cap = cap < CHAN_CAP ? cap : CHAN_CAP, c->closed = 0; }
static void chan_put(chan_t *c, void *item) /* item == NULL closes the channel */
{
	pthread_mutex_lock(&c->m);
	if (item) { while (c->n == c->cap) pthread_cond_wait(&c->c, &c->m); c->item[c->n++] = item; }
	else c->closed = 1;
	pthread_cond_broadcast(&c->c);
	pthread_mutex_unlock(&c->m);
}
static void *chan_get(chan_t *c) /* FIFO; NULL once closed and drained */
{
	void *item = 0;
	int i;
	pthread_mutex_lock(&c->m);
	while (c->n == 0 && !c->closed) pthread_cond_wait(&c->c, &c->m);
	if (c->n > 0) { item = c->item[0]; for (i = 1; i < c->n; ++i) c->item[i - 1] = c->item[i]; --c->n; pthread_cond_broadcast(&c->c); }
	pthread_mutex_unlock(&c->m);
	return item;
}

static void fbatch_free(fbatch_t *b) { if (b) { free(b->qlens); free(b->seqs); free(b->names); free(b->slab.s); free(b->nslab.s); free(b->seq_off); free(b->name_off); if (b->base_pinned) mga_hfree_pinned(b->base); else free(b->base); free(b); } }
static int fbatch_base_reserve(fbatch_t *b, size_t bytes, int want_pinned)
{
	if (b->base && bytes <= b->base_cap) return 0;
	if (b->base_pinned) mga_hfree_pinned(b->base); else free(b->base);
	b->base_cap = bytes + (bytes >> 3) + 4096;
	b->base = want_pinned ? (char*)mga_hmalloc_pinned(b->base_cap) : (char*)malloc(b->base_cap);
	b->base_pinned = want_pinned;
	if (b->base == 0) { b->base_cap = 0; mga_set_error("out of %s memory for a read batch of %zu bytes", want_pinned ? "pinned" : "host", bytes); return -1; }
	return 0;
}
static void fbatch_reset(fbatch_t *b) { b->n = 0, b->last = 0, b->pinned = 0, b->slab.l = 0, b->nslab.l = 0; }
static void fbatch_room(fbatch_t *b, int n)
{
	if (n <= b->m) return;
	b->m = n + (n >> 1) + 1024;
	b->qlens = MGA_REALLOC(int, b->qlens, b->m); b->name_off = MGA_REALLOC(size_t, b->name_off, b->m); b->seq_off = MGA_REALLOC(size_t, b->seq_off, b->m);
	b->seqs = MGA_REALLOC(char*, b->seqs, b->m); b->names = MGA_REALLOC(char*, b->names, b->m);
}
static void fbatch_finish(fbatch_t *b)
{
	char *base = b->pinned ? b->base : b->slab.s; /* (pinned: "contiguous in b->base", whether or not that memory is page-locked) */
	int i;
	for (i = 0; i < b->n; ++i) b->seqs[i] = base + b->seq_off[i], b->names[i] = b->nslab.s + b->name_off[i];
}

/* ---- parallel FASTA reader.  A record starts at a '>' that begins a line, exactly where kseq ends the previous one, so a plain
 * FASTA file can be cut at any "\n>" and the pieces parsed independently: pass 1 (all threads) lists the records of a window of the
 * file with their base counts, pass 2 (all threads) copies the bases of the records of one mini-batch to their final place in
 * pinned memory.  A line starting with '+' or '@' (FASTQ, or something kseq would treat as such) hands the rest of the file to
 * the sequential reader. ---- */
typedef struct { const char *hdr, *body, *end; int64_t nb; int32_t name_len; } fa_rec_t; /* hdr: behind the '>'; [body, end): the sequence lines */
typedef struct { fa_rec_t *a; int64_t n, m; int anomaly; } fa_list_t;
typedef struct {
	const char *map; int64_t size, pos;   /* pos: start of the first record that has not been listed yet */
	fa_rec_t *pend; int64_t n_pend, m_pend, head; int64_t pend_bases;
	int n_threads;
	int fallback;                         /* the rest of the file ([pos, size)) goes through the sequential reader */
} fa_fast_t;

static const char *fa_next_rec(const char *p, const char *end) /* first '>' at the start of a line in [p, end); end if none (p itself counts when it follows a '\n') */
{
	while (p < end) {
		const char *q = (const char*)memchr(p, '\n', (size_t)(end - p));
		if (q == 0 || q + 1 >= end) return end;
		if (q[1] == '>') return q + 1;
		p = q + 1;
	}
	return end;
}

/* MGA_DEBUG_PIPE accounting (mapper.c) */
void mga_cpu_note(int which, int64_t ns);
int64_t mga_cpu_now(void);
typedef struct { const char **cut; fa_list_t *list; } fa_scan_t;
static void fa_scan_worker1(void *data, int64_t j, int tid);
static void fa_scan_worker(void *data, int64_t j, int tid) { int64_t t0 = mga_cpu_now(); fa_scan_worker1(data, j, tid); if (t0) mga_cpu_note(15, mga_cpu_now() - t0); }
static void fa_scan_worker1(void *data, int64_t j, int tid)
{
	fa_scan_t *S = (fa_scan_t*)data;
	fa_list_t *L = &S->list[j];
	const char *p = S->cut[j], *end = S->cut[j + 1];
	(void)tid;
	while (p < end && !L->anomaly) { /* p is at a '>' */
		fa_rec_t r;
		const char *q = (const char*)memchr(p, '\n', (size_t)(end - p)), *e;
		int32_t k, hl;
		r.hdr = p + 1;
		e = q ? q : end;
		hl = (int32_t)(e - r.hdr);
		for (k = 0; k < hl; ++k) if (isspace((unsigned char)r.hdr[k])) break; /* the name ends at the first white space (kseq) */
		r.name_len = k, r.nb = 0;
		p = q ? q + 1 : end;
		r.body = p;
		while (p < end) {
			const char c = *p;
			int64_t n;
			if (c == '>') break;
			if (c == '+' || c == '@') { L->anomaly = 1; break; }
			q = (const char*)memchr(p, '\n', (size_t)(end - p));
			n = q ? q - p : end - p;
			if (n > 0 && p[n - 1] == '\r' && r.nb + n > 1) --n; /* kseq drops a line's trailing CR only once the record holds more than that one character (kseq.h ks_getuntil2) */
			r.nb += n;
			p = q ? q + 1 : end;
		}
		r.end = p;
		if (L->n == L->m) { L->m = L->m ? L->m << 1 : 1024; L->a = MGA_REALLOC(fa_rec_t, L->a, L->m); }
		L->a[L->n++] = r;
	}
}

/* list the records of the next window of about `want` bytes; returns 0, or 1 at an anomaly (nothing listed, F->fallback set) */
static int fa_scan_window(fa_fast_t *F, int64_t want)
{
	const char *w0 = F->map + F->pos, *fend = F->map + F->size, *w1;
	int K = F->n_threads, j;
	const char *cut[65];
	fa_list_t list[64];
	fa_scan_t S;
	int64_t tot = 0;
	if (want < (1 << 20)) want = 1 << 20;
	w1 = F->pos + want >= F->size ? fend : fa_next_rec(w0 + want, fend);
	if (K > 64) K = 64;
	if ((w1 - w0) / K < (1 << 18)) K = (int)((w1 - w0) >> 18) + 1;
	cut[0] = w0, cut[K] = w1;
	for (j = 1; j < K; ++j) { cut[j] = fa_next_rec(w0 + (w1 - w0) / K * j, w1); if (cut[j] < cut[j - 1]) cut[j] = cut[j - 1]; }
	memset(list, 0, sizeof(fa_list_t) * K);
	S.cut = cut, S.list = list;
	mga_parallel_for(K, K, fa_scan_worker, &S);
	for (j = 0; j < K; ++j) { tot += list[j].n; if (list[j].anomaly) F->fallback = 1; }
	if (F->fallback) { for (j = 0; j < K; ++j) free(list[j].a); return 1; }
	if (F->head > 0) { memmove(F->pend, F->pend + F->head, (size_t)(F->n_pend - F->head) * sizeof(fa_rec_t)); F->n_pend -= F->head, F->head = 0; }
	if (F->n_pend + tot > F->m_pend) { F->m_pend = (F->n_pend + tot) * 3 / 2 + 1024; F->pend = MGA_REALLOC(fa_rec_t, F->pend, F->m_pend); }
	for (j = 0; j < K; ++j) {
		int64_t i;
		for (i = 0; i < list[j].n; ++i) F->pend_bases += list[j].a[i].nb;
		memcpy(F->pend + F->n_pend, list[j].a, (size_t)list[j].n * sizeof(fa_rec_t));
		F->n_pend += list[j].n;
		free(list[j].a);
	}
	F->pos = w1 - F->map;
	return 0;
}

typedef struct { const fa_rec_t *rec; fbatch_t *b; } fa_fill_t;
static void fa_fill_worker1(void *data, int64_t i, int tid);
static void fa_fill_worker(void *data, int64_t i, int tid) { int64_t t0 = mga_cpu_now(); fa_fill_worker1(data, i, tid); if (t0) mga_cpu_note(15, mga_cpu_now() - t0); }
static void fa_fill_worker1(void *data, int64_t i, int tid)
{
	fa_fill_t *f = (fa_fill_t*)data;
	const fa_rec_t *r = &f->rec[i];
	char *dst = f->b->base + f->b->seq_off[i], *d0 = dst;
	const char *p = r->body;
	(void)tid;
	if (r->end - r->body == r->nb + 1 && r->end[-1] == '\n') { /* the usual case, one line without a CR (pass 1 counted its bases): no second search for the line end */
		seq_copy_normalize(dst, p, (size_t)r->nb);
		dst += r->nb, p = r->end;
	}
	while (p < r->end) {
		const char *q = (const char*)memchr(p, '\n', (size_t)(r->end - p));
		size_t n = q ? (size_t)(q - p) : (size_t)(r->end - p);
		const char *nx = q ? q + 1 : r->end;
		if (n > 0 && p[n - 1] == '\r' && (size_t)(dst - d0) + n > 1) --n; /* (the same rule as in pass 1) */
		seq_copy_normalize(dst, p, n); dst += n;
		p = nx;
	}
	(void)d0;
	memcpy(f->b->nslab.s + f->b->name_off[i], r->hdr, (size_t)r->name_len);
	f->b->nslab.s[f->b->name_off[i] + r->name_len] = 0;
}

/* the next mini-batch of >= batch_bases bases (bseq.c:61-98) into b; returns the number of records (0: none left in the fast part) */
static int fa_fast_batch(fa_fast_t *F, int64_t batch_bases, fbatch_t *b, int want_pinned)
{
	int64_t size = 0, noff = 0, i, n = 0;
	fa_fill_t f;
	while (!F->fallback && F->pend_bases < batch_bases && F->pos < F->size)
		if (fa_scan_window(F, batch_bases - F->pend_bases + (batch_bases >> 5) + (1 << 20))) break;
	while (F->head + n < F->n_pend && size < batch_bases) size += F->pend[F->head + n++].nb;
	if (n == 0) return 0;
	if (n > 0x7fffffff) n = 0x7fffffff;
	fbatch_reset(b);
	fbatch_room(b, (int)n);
	if (fbatch_base_reserve(b, (size_t)size + 64, want_pinned) < 0) return -1;
	for (i = 0, size = 0; i < n; ++i) {
		const fa_rec_t *r = &F->pend[F->head + i];
		b->seq_off[i] = (size_t)size, b->name_off[i] = (size_t)noff, b->qlens[i] = (int)r->nb;
		size += r->nb, noff += r->name_len + 1;
	}
	memset(b->base + size, 0, 64);
	b->nslab.l = 0; str_room(&b->nslab, (size_t)noff); b->nslab.l = (size_t)noff;
	b->n = (int)n, b->pinned = 1;
	f.rec = F->pend + F->head, f.b = b;
	mga_parallel_for(F->n_threads, n, fa_fill_worker, &f);
	F->head += n, F->pend_bases -= size;
	return (int)n;
}

typedef struct { int n_fn; const char **fn; int64_t batch_bases, first_bases; chan_t *out, *free_b; int n_threads, want_pinned; int rank, world; volatile int err; } reader_t;

/* sharded jobs, sequential reader: every rank parses the whole input and keeps the contiguous slice [n*rank/world, n*(rank+1)/world) of
 * each mini-batch */
static void fbatch_keep_slice(fbatch_t *b, int rank, int world)
{
	const int st = (int)((int64_t)b->n * rank / world), en = (int)((int64_t)b->n * (rank + 1) / world);
	int i;
	for (i = st; i < en; ++i) b->qlens[i - st] = b->qlens[i], b->seq_off[i - st] = b->seq_off[i], b->name_off[i - st] = b->name_off[i];
	b->n = en - st;
	b->pinned = 0; /* (a slice of a sequentially parsed batch is contiguous as well, but that path stages its uploads anyway) */
}

/* sequential reader (anything gzopen can read, or the rest of a memory-mapped file): records until the batch holds batch_bases bases */
static int seq_batch(rd_t *r, int *last, str_t *name, int64_t batch_bases, fbatch_t *b, int *eof)
{
	int64_t size = 0;
	fbatch_reset(b);
	while (size < batch_bases) {
		size_t s0 = b->slab.l, sl;
		if (b->slab.m == 0) str_room(&b->slab, (size_t)(batch_bases < (1LL << 30) ? batch_bases : (1LL << 30)) / 2 + 65536); /* one allocation for most batches */
		if (read_record_x(r, last, name, &b->slab, 1) < 0) { *eof = 1; break; } /* the bases go straight into the slab */
		fbatch_room(b, b->n + 1);
		sl = b->slab.l - s0;
		seq_normalize(b->slab.s + s0, sl);
		b->seq_off[b->n] = s0;
		str_room(&b->nslab, name->l + 2);
		b->name_off[b->n] = b->nslab.l; memcpy(b->nslab.s + b->nslab.l, name->s, name->l + 1); b->nslab.l += name->l + 1;
		b->qlens[b->n++] = (int)sl;
		size += (int64_t)sl;
	}
	str_room(&b->slab, 64); memset(b->slab.s + b->slab.l, 0, 64);
	return b->n;
}

static void *reader_main1(void *a);
static void *reader_main(void *a) { int64_t t0 = mga_cpu_now(); void *r = reader_main1(a); if (t0) mga_cpu_note(14, mga_cpu_now() - t0); return r; }
static void *reader_main1(void *a)
{
	reader_t *R = (reader_t*)a;
	int f, seg = 0, n_out = 0;
	fbatch_t *b = 0;
	const int world = R->world > 1 ? R->world : 1, rank = R->world > 1 ? R->rank : 0;
	for (f = 0; f < R->n_fn && !R->err; ++f) {
		rd_t r;
		int last = 0, eof = 0, fd = -1, use_fast = 0, seq_mode = 0;
		int64_t map_size = 0;
		str_t name = {0, 0, 0};
		fa_fast_t F;
		const int is_stdin = !(R->fn[f] && strcmp(R->fn[f], "-"));
		memset(&r, 0, sizeof r); memset(&F, 0, sizeof F);
		if (!is_stdin && !getenv("MGA_NO_FAST_READER")) { /* plain FASTA in a regular file: memory-map it */
			struct stat st;
			fd = open(R->fn[f], O_RDONLY);
			if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
				void *m = mmap(0, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
				if (m != MAP_FAILED) {
					if (((const char*)m)[0] == '>') {
						F.map = (const char*)m, F.size = st.st_size, F.n_threads = R->n_threads, use_fast = 1, map_size = st.st_size;
						if (world > 1) { /* this rank's byte range of the file: the records whose '>' lies in it; rank-order concatenation restores the input order */
							const char *e = F.map + F.size;
							F.pos = rank == 0 ? 0 : fa_next_rec(F.map + (int64_t)((__int128)F.size * rank / world), e) - F.map;
							F.size = rank == world - 1 ? F.size : fa_next_rec(F.map + (int64_t)((__int128)F.size * (rank + 1) / world), e) - F.map;
							if (F.pos > F.size) F.pos = F.size;
						}
						madvise((char*)m + (F.pos & ~4095LL), (size_t)(F.size - (F.pos & ~4095LL)), MADV_WILLNEED); /* (advice values are not flags: one call, one advice) */
					} else munmap(m, (size_t)st.st_size);
				}
			}
			if (fd >= 0) close(fd);
		}
		if (!use_fast) {
			r.fp = !is_stdin ? gzopen(R->fn[f], "r") : gzdopen(0, "r");
			if (r.fp == 0) { if (mg_verbose >= 1) fprintf(stderr, "ERROR: failed to open file '%s'\n", R->fn[f]); R->err = 1; break; }
			r.buf = (char*)malloc(RD_BUF);
		}
		while (!eof && !R->err) {
			int n;
			if (b == 0) b = (fbatch_t*)chan_get(R->free_b);
			if (use_fast && !seq_mode) {
				n = fa_fast_batch(&F, n_out == 0 && R->first_bases > 0 ? R->first_bases : R->batch_bases, b, R->want_pinned);
				if (n < 0) { R->err = 1; break; }
				if (n == 0) { /* the fast part is exhausted: either the file is, or the sequential reader takes over at F.pos */
					if (!F.fallback || F.pos >= F.size) { eof = 1; break; }
					r.fp = 0, r.buf = (char*)F.map + F.pos, r.beg = 0, r.end = F.size - F.pos;
					seq_mode = 1;
					continue;
				}
				if (F.pos >= F.size && F.head >= F.n_pend) eof = 1;
			} else {
				n = seq_batch(&r, &last, &name, n_out == 0 && R->first_bases > 0 && world == 1 ? R->first_bases : R->batch_bases, b, &eof);
				if (n > 0 && world > 1 && !use_fast) fbatch_keep_slice(b, rank, world);
			}
			if (n == 0) break;
			b->seg = seg;
			if (!use_fast) ++seg; /* sequential input: every mini-batch is a segment of its own */
			b->last = eof && f == R->n_fn - 1;
			fbatch_finish(b);
			chan_put(R->out, b);
			b = 0, ++n_out;
		}
		free(name.s);
		if (use_fast) { free(F.pend); munmap((void*)F.map, (size_t)map_size); ++seg; /* memory-mapped input: the file is one segment */ }
		else { free(r.buf); gzclose(r.fp); }
	}
	if (b) chan_put(R->free_b, b);
	chan_put(R->out, 0);
	return 0;
}

/* parser check without a device: number of records, bases, and an FNV-1a hash over "name\nSEQ\n" of every record, through the same
 * reader thread mg_map_files() uses (the parallel FASTA reader where it applies -- MGA_NO_FAST_READER=1 forces the sequential one --
 * in batches of batch_bases bases) */
static int reads_parse_shard(const char *fn, int64_t batch_bases, int n_threads, int rank, int world, FILE *dump, int64_t **seg_n, int *n_seg,
							 int64_t *n_reads, int64_t *n_bases, uint64_t *hash);
int mga_reads_parse_x(const char *fn, int64_t batch_bases, int n_threads, int64_t *n_reads, int64_t *n_bases, uint64_t *hash)
{
	return reads_parse_shard(fn, batch_bases, n_threads, 0, 1, 0, 0, 0, n_reads, n_bases, hash);
}
/* the shard of rank/world exactly as mga_map_files_shard() would map it, written as FASTA (one line per sequence) to out_path;
 * seg_n[0..n_seg) (malloc'ed) = records of each output segment.  No device needed: launchers and tests use it to look at the split. */
int mga_reads_shard_dump(const char *fn, int64_t batch_bases, int n_threads, int rank, int world, const char *out_path, int64_t **seg_n, int *n_seg)
{
	int64_t nr, nb; uint64_t h;
	FILE *fp = fopen(out_path, "wb");
	int rc;
	if (fp == 0) { mga_set_error("cannot open '%s' for writing", out_path); return -1; }
	rc = reads_parse_shard(fn, batch_bases, n_threads, rank, world, fp, seg_n, n_seg, &nr, &nb, &h);
	if (fclose(fp) != 0) rc = -1;
	return rc;
}
static int reads_parse_shard(const char *fn, int64_t batch_bases, int n_threads, int rank, int world, FILE *dump, int64_t **seg_n, int *n_seg,
							 int64_t *n_reads, int64_t *n_bases, uint64_t *hash)
{
	int64_t *sn = 0; int nsn = 0, msn = 0;
	chan_t c_in, c_free;
	reader_t R;
	pthread_t t_rd;
	fbatch_t *fb[2], *b;
	uint64_t h = 0xcbf29ce484222325ULL;
	int i;
	size_t k;
	*n_reads = *n_bases = 0; if (hash) *hash = 0;
	chan_init(&c_in, 1); chan_init(&c_free, 2);
	for (i = 0; i < 2; ++i) { fb[i] = MGA_CALLOC(fbatch_t, 1); chan_put(&c_free, fb[i]); }
	R.n_fn = 1, R.fn = &fn, R.batch_bases = batch_bases > 0 ? batch_bases : 500000000, R.out = &c_in, R.free_b = &c_free, R.err = 0, R.n_threads = n_threads > 0 ? n_threads : 1, R.want_pinned = 0, R.rank = rank, R.world = world, R.first_bases = 0;
	pthread_create(&t_rd, 0, reader_main, &R);
	while ((b = (fbatch_t*)chan_get(&c_in)) != 0) {
		while (b->seg >= nsn) { MGA_GROW(int64_t, sn, nsn, msn); sn[nsn++] = 0; }
		sn[b->seg] += b->n;
		for (i = 0; i < b->n; ++i) {
			const char *nm = b->names[i], *sq = b->seqs[i];
			if (hash == 0) { ++*n_reads, *n_bases += b->qlens[i]; continue; } /* (timing the reader alone: no checksum) */
			if (dump) { fputc('>', dump); fputs(nm, dump); fputc('\n', dump); fwrite(sq, 1, (size_t)b->qlens[i], dump); fputc('\n', dump); }
			for (k = 0; nm[k]; ++k) h = (h ^ (unsigned char)nm[k]) * 0x100000001b3ULL;
			h = (h ^ '\n') * 0x100000001b3ULL;
			for (k = 0; k < (size_t)b->qlens[i]; ++k) h = (h ^ (unsigned char)sq[k]) * 0x100000001b3ULL;
			h = (h ^ '\n') * 0x100000001b3ULL;
			++*n_reads, *n_bases += b->qlens[i];
		}
		chan_put(&c_free, b);
	}
	pthread_join(t_rd, 0);
	for (i = 0; i < 2; ++i) fbatch_free(fb[i]);
	if (hash) *hash = h;
	if (seg_n) *seg_n = sn, *n_seg = nsn; else free(sn);
	if (R.err) { mga_set_error("cannot read '%s'", fn); return -1; }
	return 0;
}
int mga_reads_parse(const char *fn, int64_t *n_reads, int64_t *n_bases, uint64_t *hash) { return mga_reads_parse_x(fn, 0, 4, n_reads, n_bases, hash); }

/* where the GAF text goes: a stream, or one buffer in memory (bench.py's "file -> GAF buffer" interval) */
typedef struct { FILE *fp; int to_mem; char *mem; int64_t mem_len, mem_cap; int64_t *seg_len; int n_seg, m_seg; } sink_t;
typedef struct { char *buf; int64_t len, cap; fbatch_t *fb; int seg; } wbuf_t;
typedef struct { sink_t *sink; chan_t *in; pthread_mutex_t *pool_m; wbuf_t **pool; int *n_pool; volatile int err; } writer_t;

/* A mini-batch's GAF text (hundreds of MB) into a REGULAR file: slices of it by several threads with pwrite() at their final offsets -- filling fresh page-cache pages is
 * ~2 GB/s per thread ([measured, round 6] one fwrite per mini-batch made the writer the job's bottleneck: 518 ms per 1.1 GB step against 310 ms of mapping).  Pipes and
 * terminals take the plain fwrite. */
typedef struct { int fd; const char *buf; int64_t len, off; int err; } wslice_t;
static void *wslice_main(void *a)
{
	wslice_t *x = (wslice_t*)a;
	int64_t done = 0;
	while (done < x->len) {
		const ssize_t k = pwrite(x->fd, x->buf + done, (size_t)(x->len - done), (off_t)(x->off + done));
		if (k <= 0) { x->err = 1; break; }
		done += k;
	}
	return 0;
}
static int write_parallel(FILE *fp, const char *buf, int64_t len)
{
	enum { NW = 4 };
	struct stat st;
	const int fd = fileno(fp);
	off_t at;
	wslice_t sl[NW];
	pthread_t th[NW];
	int k, bad = 0;
	static int64_t min_len = -1; /* MGA_PWRITE_MIN=<bytes>: the tests push small outputs through this path */
	if (min_len < 0) { const char *e = getenv("MGA_PWRITE_MIN"); min_len = e && *e ? atoll(e) : (64 << 20); }
	if (len < min_len || len < NW || fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return -1;
	if (fflush(fp) != 0 || (at = lseek(fd, 0, SEEK_CUR)) < 0) return -1; /* (the stream's buffer is empty now: its position is the descriptor's) */
	for (k = 0; k < NW; ++k) {
		const int64_t b = len * k / NW, e = len * (k + 1) / NW;
		sl[k].fd = fd, sl[k].buf = buf + b, sl[k].len = e - b, sl[k].off = (int64_t)at + b, sl[k].err = 0;
		if (k > 0) pthread_create(&th[k], 0, wslice_main, &sl[k]);
	}
	wslice_main(&sl[0]);
	for (k = 1; k < NW; ++k) pthread_join(th[k], 0);
	for (k = 0; k < NW; ++k) bad |= sl[k].err;
	if (bad || lseek(fd, at + (off_t)len, SEEK_SET) < 0) return -2;
	return 0;
}

static void *writer_main(void *a)
{
	writer_t *W = (writer_t*)a;
	wbuf_t *w;
	while ((w = (wbuf_t*)chan_get(W->in)) != 0) {
		sink_t *s = W->sink;
		int64_t t0_ = mga_cpu_now();
		while (w->seg >= s->n_seg) { MGA_GROW(int64_t, s->seg_len, s->n_seg, s->m_seg); s->seg_len[s->n_seg++] = 0; }
		s->seg_len[w->seg] += w->len;
		if (!W->err && w->len > 0) {
			if (s->fp) {
				const int pw = write_parallel(s->fp, w->buf, w->len); /* -1: not a large write to a regular file */
				if (pw == -2 || (pw == -1 && fwrite(w->buf, 1, (size_t)w->len, s->fp) != (size_t)w->len)) { fprintf(stderr, "[E::%s] failed to write the results\n", __func__); W->err = 1; }
			}
			else {
				if (s->mem_len + w->len + 1 > s->mem_cap) { s->mem_cap = (s->mem_len + w->len + 1) * 3 / 2; mga_host_unpin(s->mem); s->mem = (char*)realloc(s->mem, (size_t)s->mem_cap); }
				memcpy(s->mem + s->mem_len, w->buf, (size_t)w->len); s->mem_len += w->len;
			}
		}
		if (t0_) mga_cpu_note(16, mga_cpu_now() - t0_);
		pthread_mutex_lock(W->pool_m); W->pool[(*W->n_pool)++] = w; pthread_mutex_unlock(W->pool_m); /* the buffer goes back to the submitter */
	}
	return 0;
}

typedef struct { mga_stream_t *S; chan_t *sub, *out, *free_b; volatile int err; char errmsg[512]; } collector_t;
static void *collector_main(void *a)
{
	collector_t *C = (collector_t*)a;
	wbuf_t *w;
	while ((w = (wbuf_t*)chan_get(C->sub)) != 0) { /* one token per submitted batch, in submission order */
		void *user = 0;
		int rc = mga_stream_collect(C->S, &w->buf, &w->len, &w->cap, &user);
		if (rc < 0 && !C->err) { C->err = 1; snprintf(C->errmsg, sizeof C->errmsg, "%s", mga_last_error()); }
		if (rc <= 0) w->len = 0;
		if (mg_verbose >= 3 && w->fb) fprintf(stderr, "[M::%s] mapped %d sequences\n", "mg_map_files", w->fb->n);
		if (w->fb) chan_put(C->free_b, w->fb); /* the batch's memory is free to be parsed into again */
		w->fb = 0;
		chan_put(C->out, w);
	}
	chan_put(C->out, 0);
	return 0;
}

#define MF_NB 5 /* read batches in rotation: one being parsed, one parsed, up to three in the pipeline */
typedef struct { fbatch_t *fb[MF_NB]; wbuf_t wb[MF_NB + 2]; } mf_cache_t;

void mga_idx_mf_free(mg_idx_t *gi)
{
	mf_cache_t *c;
	int k;
	if (gi == 0 || gi->B == 0 || (c = (mf_cache_t*)gi->B->mf_cache) == 0) return;
	for (k = 0; k < MF_NB; ++k) fbatch_free(c->fb[k]);
	for (k = 0; k < MF_NB + 2; ++k) free(c->wb[k].buf);
	free(c);
	gi->B->mf_cache = 0;
}

/* map the files against an existing index; GAF to fp, or (fp == NULL) into one buffer in memory: *mem / *mem_cap on entry = a buffer to
 * reuse (or NULL), on return the GAF text (*mem_len bytes, NUL-terminated; release with mga_free()).
 * *t_map (optional): seconds from the first byte read to the last byte handed to the sink -- the interval between the reference's
 * mg_opt_update and its last worker_pipeline log line (gmap.c:186-211).  The job runs on the index's own chunk pipeline and keeps its
 * read batches and output buffers with the index, so a second job on the same index allocates nothing. */
int mga_map_files_shard(const mg_idx_t *gi, int n_fn, const char **fn, const mg_mapopt_t *opt, int n_threads, int shard_rank, int shard_world,
						FILE *fp, char **mem, int64_t *mem_len, int64_t *mem_cap, int64_t **seg_len, int *n_seg, double *t_map)
{
	int ret = 0, k, n_sub = 0, n_pool = 0;
	chan_t c_in, c_free, c_sub, c_out;
	reader_t R;
	writer_t W;
	collector_t C;
	sink_t sink;
	wbuf_t *pool[MF_NB + 2];
	fbatch_t *b;
	mf_cache_t *mc;
	pthread_mutex_t pool_m = PTHREAD_MUTEX_INITIALIZER;
	pthread_t t_rd, t_wr, t_co;
	mga_stream_t *S;
	const double t0 = mga_wtime();
	memset(&sink, 0, sizeof sink);
	if (mem) { /* a buffer to reuse comes with its capacity; one without is released, not dropped */
		sink.mem = *mem, sink.mem_cap = *mem && mem_cap ? *mem_cap : 0;
		if (sink.mem_cap == 0 && sink.mem) { mga_host_unpin(sink.mem); free(sink.mem); sink.mem = 0; }
		*mem = 0, *mem_len = 0; sink.to_mem = 1;
	}
	if (opt->flag & (MG_M_FRAG_MODE | MG_M_CAL_COV)) { /* gmap.c:44-48,119-126,199-214: multi-segment fragments and --cov are not on the accelerated path */
		if (mg_verbose >= 1) fprintf(stderr, "[E::%s] --frag and --cov are outside the MI355X long-read path (single-segment reads, GAF output)\n", __func__);
		mga_host_unpin(sink.mem); free(sink.mem);
		return -1;
	}
	if (n_threads < 1) n_threads = 1;
	if ((S = mga_idx_stream_acquire(gi, opt, n_threads)) == 0) { mga_host_unpin(sink.mem); free(sink.mem); return -1; }
	if (gi->B->mf_cache == 0) { /* (under the stream's job lock) */
		mc = MGA_CALLOC(mf_cache_t, 1);
		for (k = 0; k < MF_NB; ++k) mc->fb[k] = MGA_CALLOC(fbatch_t, 1);
		gi->B->mf_cache = mc;
	}
	mc = (mf_cache_t*)gi->B->mf_cache;
	if (sink.to_mem && sink.mem == 0) { /* one allocation of about the size of the input for the whole output (a base-aligned read prints ~0.9 bytes per base) */
		struct stat st;
		int64_t tot = 0;
		for (k = 0; k < n_fn; ++k) if (fn[k] && stat(fn[k], &st) == 0 && S_ISREG(st.st_mode)) tot += st.st_size;
		sink.mem_cap = (shard_world > 1 ? tot / shard_world : tot) + (tot >> 4) + (1 << 20);
		sink.mem = (char*)malloc((size_t)sink.mem_cap);
	}
	chan_init(&c_in, 1); chan_init(&c_free, MF_NB); chan_init(&c_sub, MF_NB + 2); chan_init(&c_out, 2);
	sink.fp = fp;
	for (k = 0; k < MF_NB; ++k) chan_put(&c_free, mc->fb[k]);
	for (k = 0; k < MF_NB + 2; ++k) pool[n_pool++] = &mc->wb[k];
	R.n_fn = n_fn, R.fn = fn, R.batch_bases = opt->mini_batch_size, R.out = &c_in, R.free_b = &c_free, R.err = 0;
	R.n_threads = n_threads < 8 ? n_threads : 8, R.want_pinned = 1, R.rank = shard_rank, R.world = shard_world;
	R.first_bases = 64000000 < opt->mini_batch_size ? 64000000 : opt->mini_batch_size; /* a short first batch: the GPU starts while the reader is still on the second one */
	W.sink = &sink, W.in = &c_out, W.pool_m = &pool_m, W.pool = pool, W.n_pool = &n_pool, W.err = 0;
	C.S = S, C.sub = &c_sub, C.out = &c_out, C.free_b = &c_free, C.err = 0, C.errmsg[0] = 0;
	pthread_create(&t_rd, 0, reader_main, &R);
	pthread_create(&t_co, 0, collector_main, &C);
	pthread_create(&t_wr, 0, writer_main, &W);
	while ((b = (fbatch_t*)chan_get(&c_in)) != 0) {
		wbuf_t *w = 0;
		if (C.err || W.err) { chan_put(&c_free, b); continue; } /* keep draining the reader */
		for (;;) { /* an output buffer: MF_NB + 2 circulate, the submit below blocks long before they run out */
			pthread_mutex_lock(&pool_m); if (n_pool > 0) w = pool[--n_pool]; pthread_mutex_unlock(&pool_m);
			if (w) break;
			usleep(200);
		}
		w->fb = b, w->len = 0, w->seg = b->seg;
		mga_stream_submit(S, b->n, b->qlens, (const char**)b->seqs, (const char**)b->names, 0, 1, 0, 0, b->pinned && b->base_pinned, (n_sub == 0 ? MGA_SB_FIRST : 0) | (b->last ? MGA_SB_LAST : 0), w->buf, w->cap, b);
		w->buf = 0, w->cap = 0;
		++n_sub;
		chan_put(&c_sub, w);
	}
	chan_put(&c_sub, 0);
	pthread_join(t_rd, 0); pthread_join(t_co, 0); pthread_join(t_wr, 0);
	if (C.err) { mga_set_error("%s", C.errmsg); ret = -1; }
	if (R.err || W.err) ret = -1;
	mga_idx_stream_release(S);
	if (mem && ret == 0) { if (sink.mem == 0) sink.mem = (char*)calloc(1, 1), sink.mem_cap = 1; sink.mem[sink.mem_len] = 0; *mem = sink.mem, *mem_len = sink.mem_len; if (mem_cap) *mem_cap = sink.mem_cap; }
	else { mga_host_unpin(sink.mem); free(sink.mem); }
	if (seg_len && ret == 0) *seg_len = sink.seg_len, *n_seg = sink.n_seg; else free(sink.seg_len);
	if (t_map) *t_map = mga_wtime() - t0;
	return ret;
}

int mga_map_files_idx(const mg_idx_t *gi, int n_fn, const char **fn, const mg_mapopt_t *opt, int n_threads, FILE *fp, char **mem, int64_t *mem_len, double *t_map)
{
	return mga_map_files_shard(gi, n_fn, fn, opt, n_threads, 0, 1, fp, mem, mem_len, 0, 0, 0, t_map);
}

int mg_map_files_fp(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads, FILE *out)
{
	mg_mapopt_t opt = *opt0;
	mg_idx_t *gi;
	int ret;
	double t_map = 0.0;
	if (opt.flag & (MG_M_FRAG_MODE | MG_M_CAL_COV)) {
		if (mg_verbose >= 1) fprintf(stderr, "[E::%s] --frag and --cov are outside the MI355X long-read path (single-segment reads, GAF output)\n", __func__);
		return -1;
	}
	if ((gi = mg_index(g, ipt, n_threads, &opt)) == 0) return -1;
	ret = mga_map_files_idx(gi, n_fn, fn, &opt, n_threads, out, 0, 0, &t_map);
	if (ret < 0) fprintf(stderr, "[E::%s] %s\n", __func__, mga_last_error());
	if (mg_verbose >= 3) fprintf(stderr, "[M::%s] mapping phase %.3f s\n", __func__, t_map);
	mg_idx_destroy(gi);
	return ret;
}

int mg_map_files(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads)
{
	return mg_map_files_fp(g, n_fn, fn, ipt, opt0, n_threads, stdout);
}

int mga_map_files_to_path(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads, const char *out_path)
{
	FILE *fp = fopen(out_path, "wb");
	int ret;
	if (fp == 0) { mga_set_error("cannot open '%s' for writing", out_path); return -1; }
	ret = mg_map_files_fp(g, n_fn, fn, ipt, opt0, n_threads, fp);
	if (fclose(fp) != 0) ret = -1;
	return ret;
}
