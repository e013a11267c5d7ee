/*
 * gfa_load.c -- rGFA / FASTA reader producing the reference-compatible in-memory graph (gfa_t).
 *
 * Out of scope to accelerate (load-time I/O); it exists so that the library is usable stand-alone
 * and so that the graph handed to the hot path is IDENTICAL to what the reference builds, including
 * the order of arcs leaving a vertex: gfa_finalize() (gfa-base.c:421-430) sorts arcs with the
 * unstable klib radix sort on v_lv, and all arcs of one vertex share one key, so arc order -- which
 * decides ties in mg_shortest_k() and GWFA -- is a function of that exact permutation.
 *
 * Behaviour follows gfa-io.c:130-340 (parsers) and gfa-base.c:64-195,232-325,421-430 (finalize);
 * only what the mapping path reads is materialised (no aux tags beyond LN/SN/SO/SR on S-lines and
 * SR/L1/L2 on L-lines, no unitigs).
 */
#include <zlib.h>
#include <stdio.h>
#include <ctype.h>
#include <limits.h>
#include "mga_host.h"

/* ---- string -> id open-addressing map (names never removed) ---- */
typedef struct { uint32_t cap, n; char **key; uint32_t *val; } smap_t;

static void smap_grow(smap_t *h)
{
	uint32_t ocap = h->cap, i;
	char **ok = h->key; uint32_t *ov = h->val;
	h->cap = ocap ? ocap << 1 : 1024;
	h->key = MGA_CALLOC(char*, h->cap);
	h->val = MGA_CALLOC(uint32_t, h->cap);
	for (i = 0; i < ocap; ++i)
		if (ok[i]) {
			uint32_t j = mga_hash_str(ok[i]) * 2654435769U & (h->cap - 1);
			while (h->key[j]) j = (j + 1) & (h->cap - 1);
			h->key[j] = ok[i], h->val[j] = ov[i];
		}
	free(ok); free(ov);
}

/* returns the slot of key; *absent = 1 if it was inserted (caller sets val and may replace key pointer) */
static uint32_t smap_put(smap_t *h, const char *key, int *absent)
{
	uint32_t j;
	if (h->n * 2 >= h->cap) smap_grow(h);
	j = mga_hash_str(key) * 2654435769U & (h->cap - 1);
	while (h->key[j] && strcmp(h->key[j], key) != 0) j = (j + 1) & (h->cap - 1);
	*absent = h->key[j] == 0;
	if (*absent) h->key[j] = (char*)key, ++h->n;
	return j;
}

static void smap_free(smap_t *h) { if (h) { free(h->key); free(h->val); free(h); } }

static char *dup_str(const char *s, size_t n)
{
	char *t = (char*)malloc(n + 1);
	memcpy(t, s, n); t[n] = 0;
	return t;
}

/* gfa_add_seg, gfa-base.c:64-86 */
static int32_t add_seg(gfa_t *g, const char *name)
{
	smap_t *h = (smap_t*)g->h_names;
	int absent;
	uint32_t k = smap_put(h, name, &absent);
	if (absent) {
		gfa_seg_t *s;
		if (g->n_seg == g->m_seg) {
			uint32_t old = g->m_seg;
			g->m_seg = old ? old << 1 : 16;
			g->seg = MGA_REALLOC(gfa_seg_t, g->seg, g->m_seg);
			memset(&g->seg[old], 0, (size_t)(g->m_seg - old) * sizeof(gfa_seg_t));
		}
		s = &g->seg[g->n_seg++];
		h->key[k] = s->name = dup_str(name, strlen(name));
		s->del = 0, s->len = 0, s->snid = s->soff = s->rank = -1;
		h->val[k] = g->n_seg - 1;
	}
	return (int32_t)h->val[k];
}

/* gfa_sseq_add, gfa-base.c:88-104 */
static int32_t add_sseq(gfa_t *g, const char *name)
{
	smap_t *h = (smap_t*)g->h_snames;
	int absent;
	uint32_t k = smap_put(h, name, &absent);
	if (absent) {
		gfa_sseq_t *ss;
		if (g->n_sseq == g->m_sseq) { g->m_sseq = g->m_sseq ? g->m_sseq + (g->m_sseq >> 1) : 16; g->sseq = MGA_REALLOC(gfa_sseq_t, g->sseq, g->m_sseq); }
		ss = &g->sseq[g->n_sseq++];
		h->val[k] = g->n_sseq - 1;
		h->key[k] = ss->name = dup_str(name, strlen(name));
		ss->min = ss->max = ss->rank = -1;
	}
	return (int32_t)h->val[k];
}

/* gfa_sseq_update, gfa-base.c:114-126 */
static void update_sseq(gfa_t *g, const gfa_seg_t *s)
{
	gfa_sseq_t *ps;
	if (s->snid < 0 || s->snid >= (int32_t)g->n_sseq) return;
	ps = &g->sseq[s->snid];
	if (ps->min < 0 || s->soff < ps->min) ps->min = s->soff;
	if (ps->max < 0 || s->soff + s->len > ps->max) ps->max = s->soff + s->len;
	if (ps->rank < 0) ps->rank = s->rank;
	else if (ps->rank != s->rank && mg_verbose >= 2)
		fprintf(stderr, "[W] stable sequence '%s' associated with different ranks on segment '%s': %d != %d\n", ps->name, s->name, ps->rank, s->rank);
}

/* gfa_add_arc1, gfa-base.c:136-155 */
static gfa_arc_t *add_arc(gfa_t *g, uint32_t v, uint32_t w, int32_t ov, int32_t ow, int64_t link_id, int comp)
{
	gfa_arc_t *a;
	if (g->m_arc == g->n_arc) {
		uint64_t old = g->m_arc;
		g->m_arc = old ? old << 1 : 16;
		g->arc = MGA_REALLOC(gfa_arc_t, g->arc, g->m_arc);
		memset(&g->arc[old], 0, (size_t)(g->m_arc - old) * sizeof(gfa_arc_t));
		g->link_aux = MGA_REALLOC(gfa_aux_t, g->link_aux, g->m_arc);
		memset(&g->link_aux[old], 0, (size_t)(g->m_arc - old) * sizeof(gfa_aux_t));
	}
	a = &g->arc[g->n_arc++];
	a->v_lv = (uint64_t)v << 32;
	a->w = w, a->ov = ov, a->ow = ow, a->rank = -1;
	a->link_id = link_id >= 0 ? (uint64_t)link_id : g->n_arc - 1;
	if (link_id >= 0) a->rank = g->arc[link_id].rank;
	a->del = a->strong = 0;
	a->comp = comp;
	return a;
}

/* optional tags "XX:T:value": find one on the rest of a line; returns pointer to value or NULL */
static const char *find_tag(const char *rest, const char *tag, char type)
{
	const char *p = rest;
	while (p && *p) {
		const char *q = strchr(p, '\t');
		size_t l = q ? (size_t)(q - p) : strlen(p);
		if (l >= 5 && p[0] == tag[0] && p[1] == tag[1] && p[2] == ':' && p[3] == type && p[4] == ':') return p + 5;
		p = q ? q + 1 : 0;
	}
	return 0;
}

static int parse_S(gfa_t *g, char *s) /* gfa-io.c:130-192 */
{
	char *name = s + 2, *seq, *rest = 0, *p;
	int32_t sid, LN = -1;
	uint32_t len = 0;
	gfa_seg_t *sg;
	const char *t;
	if ((p = strchr(name, '\t')) == 0) return -1;
	*p = 0, seq = p + 1;
	if ((p = strchr(seq, '\t')) != 0) *p = 0, rest = p + 1;
	if (rest && (t = find_tag(rest, "LN", 'i')) != 0) LN = (int32_t)strtol(t, 0, 10);
	if (seq[0] == '*') { if (LN >= 0) len = LN; seq = 0; }
	else len = (uint32_t)strlen(seq);
	if (LN >= 0 && (int32_t)len != LN && mg_verbose >= 2)
		fprintf(stderr, "[W] for segment '%s', LN:i:%d tag is different from sequence length %d\n", name, LN, len);
	sid = add_seg(g, name);
	sg = &g->seg[sid];
	sg->len = len, sg->seq = seq ? dup_str(seq, len) : 0;
	if (rest) {
		int has_tag = 0;
		if ((t = find_tag(rest, "SN", 'Z')) != 0) {
			const char *e = strchr(t, '\t');
			char *nm = dup_str(t, e ? (size_t)(e - t) : strlen(t));
			sg->snid = add_sseq(g, nm), sg->soff = 0;
			free(nm);
			if ((t = find_tag(rest, "SO", 'i')) != 0) sg->soff = (int32_t)strtol(t, 0, 10);
			has_tag = 1;
		}
		if ((t = find_tag(rest, "SR", 'i')) != 0) {
			sg->rank = (int32_t)strtol(t, 0, 10);
			if (sg->rank > (int32_t)g->max_rank) g->max_rank = sg->rank;
			has_tag = 1;
		}
		if (has_tag || *rest) update_sseq(g, sg); /* gfa-io.c:184: any non-empty aux block */
	}
	return 0;
}

static int parse_L(gfa_t *g, char *s) /* gfa-io.c:194-264 */
{
	char *f[5], *rest = 0, *p = s + 2;
	int i, oriv, oriw, n_f = 0;
	int32_t ov = INT32_MAX, ow = INT32_MAX;
	uint32_t v, w;
	gfa_arc_t *arc;
	const char *t;
	for (i = 0; i < 5 && p; ++i) {
		char *q = strchr(p, '\t');
		f[n_f++] = p;
		if (q) *q = 0, p = q + 1; else p = 0;
	}
	rest = p;
	if (n_f < 4) return -1;
	if ((f[1][0] != '+' && f[1][0] != '-') || (f[3][0] != '+' && f[3][0] != '-')) return -2;
	oriv = f[1][0] != '+', oriw = f[3][0] != '+';
	if (n_f == 4) ov = ow = 0; /* no overlap field */
	else {
		char *q = f[4];
		if (*q == '*') ov = ow = 0;
		else if (*q == ':') { ov = INT32_MAX; ow = isdigit((unsigned char)q[1]) ? (int32_t)strtol(q + 1, &q, 10) : INT32_MAX; }
		else if (isdigit((unsigned char)*q)) {
			char *r;
			ov = (int32_t)strtol(q, &r, 10);
			if (isupper((unsigned char)*r)) { /* CIGAR */
				ov = ow = 0;
				do {
					long l = strtol(q, &q, 10);
					if (*q == 'M' || *q == 'D' || *q == 'N') ov += (int32_t)l;
					if (*q == 'M' || *q == 'I' || *q == 'S') ow += (int32_t)l;
					++q;
				} while (isdigit((unsigned char)*q));
			} else if (*r == ':') ow = isdigit((unsigned char)r[1]) ? (int32_t)strtol(r + 1, &r, 10) : INT32_MAX;
			else return -1;
		} else return -1;
	}
	v = (uint32_t)add_seg(g, f[0]) << 1 | oriv;
	w = (uint32_t)add_seg(g, f[2]) << 1 | oriw;
	arc = add_arc(g, v, w, ov, ow, -1, 0);
	if (rest) {
		if ((t = find_tag(rest, "SR", 'i')) != 0) arc->rank = (int32_t)strtol(t, 0, 10);
		if ((t = find_tag(rest, "L1", 'i')) != 0 && ov != INT32_MAX) {
			int32_t l1 = ov + (int32_t)strtol(t, 0, 10);
			if (g->seg[v>>1].len < l1) g->seg[v>>1].len = l1;
		}
		if ((t = find_tag(rest, "L2", 'i')) != 0 && ow != INT32_MAX) {
			int32_t l2 = ow + (int32_t)strtol(t, 0, 10);
			if (g->seg[w>>1].len < l2) g->seg[w>>1].len = l2;
		}
	}
	return 0;
}

/* ---- finalize (gfa-base.c:157-325,421-430) ---- */

static void arc_sort(gfa_t *g) /* radix_sort_arc on v_lv: exact permutation */
{
	int64_t n = (int64_t)g->n_arc, i, *perm;
	uint64_t *key;
	gfa_arc_t *tmp;
	if (n <= 1) return;
	key = MGA_MALLOC(uint64_t, n); perm = MGA_MALLOC(int64_t, n); tmp = MGA_MALLOC(gfa_arc_t, n);
	for (i = 0; i < n; ++i) key[i] = g->arc[i].v_lv;
	mga_ksort_perm(n, key, 8, perm);
	for (i = 0; i < n; ++i) tmp[i] = g->arc[perm[i]];
	memcpy(g->arc, tmp, (size_t)n * sizeof(gfa_arc_t));
	free(key); free(perm); free(tmp);
}

static void arc_index(gfa_t *g) /* gfa-base.c:174-195 */
{
	uint64_t i, last, n = g->n_arc;
	free(g->idx);
	g->idx = MGA_CALLOC(uint64_t, (size_t)g->n_seg * 2);
	for (i = 1, last = 0; i <= n; ++i)
		if (i == n || (uint32_t)(g->arc[i-1].v_lv >> 32) != (uint32_t)(g->arc[i].v_lv >> 32))
			g->idx[(uint32_t)(g->arc[i-1].v_lv >> 32)] = last << 32 | (i - last), last = i;
}

static int arc_is_sorted(const gfa_t *g)
{
	uint64_t e;
	for (e = 1; e < g->n_arc; ++e)
		if (g->arc[e-1].v_lv > g->arc[e].v_lv) return 0;
	return 1;
}

static void fix_semi_arc(gfa_t *g) /* gfa-base.c:232-267: infer a missing overlap length from the complement arc */
{
	uint32_t v, n_vtx = gfa_n_vtx(g);
	for (v = 0; v < n_vtx; ++v) {
		int i, j, nv = (int)gfa_arc_n(g, v);
		gfa_arc_t *av = gfa_arc_a(g, v);
		for (i = 0; i < nv; ++i) {
			uint32_t w;
			int c = 0, jv = -1, nw, multi = 0;
			gfa_arc_t *aw;
			if (av[i].del || (av[i].ow != INT32_MAX && av[i].ov != INT32_MAX)) continue;
			w = av[i].w ^ 1;
			nw = (int)gfa_arc_n(g, w), aw = gfa_arc_a(g, w);
			for (j = 0; j < nw; ++j)
				if (!aw[j].del && aw[j].w == (v ^ 1)) ++c, jv = j;
			if (c == 1) {
				if (av[i].ov != INT32_MAX && aw[jv].ow != INT32_MAX && av[i].ov != aw[jv].ow) multi = 1;
				if (av[i].ow != INT32_MAX && aw[jv].ov != INT32_MAX && av[i].ow != aw[jv].ov) multi = 1;
			}
			if (c == 1 && !multi) {
				if (aw[jv].ov != INT32_MAX) av[i].ow = aw[jv].ov;
				if (aw[jv].ow != INT32_MAX) av[i].ov = aw[jv].ow;
			} else {
				if (mg_verbose >= 2) fprintf(stderr, "[W] can't infer overlap length for %s%c -> %s%c\n", g->seg[v>>1].name, "+-"[v&1], g->seg[w>>1].name, "+-"[(w^1)&1]);
				av[i].del = 1;
			}
		}
	}
}

static void fix_symm_add(gfa_t *g) /* gfa-base.c:269-303: make sure every arc has its complement */
{
	uint32_t v, n_vtx = gfa_n_vtx(g);
	for (v = 0; v < n_vtx; ++v) {
		int i, nv = (int)gfa_arc_n(g, v);
		gfa_arc_t *av = gfa_arc_a(g, v);
		for (i = 0; i < nv; ++i) {
			int j, nw;
			gfa_arc_t *aw, *avi = &av[i];
			if (avi->del || avi->comp) continue;
			nw = (int)gfa_arc_n(g, avi->w ^ 1), aw = gfa_arc_a(g, avi->w ^ 1);
			for (j = 0; j < nw; ++j) {
				gfa_arc_t *awj = &aw[j];
				if (awj->del || awj->comp) continue;
				if (awj->w == (v ^ 1) && awj->ov == avi->ow && awj->ow == avi->ov) {
					awj->comp = 1, awj->link_id = avi->link_id;
					break;
				}
			}
			if (j == nw) {
				gfa_arc_t *old = g->arc, *na;
				na = add_arc(g, avi->w ^ 1, v ^ 1, avi->ow, avi->ov, (int64_t)avi->link_id, 1);
				if (old != g->arc) av = gfa_arc_a(g, v);
				na->rank = av[i].rank;
			}
		}
	}
	/* the reference re-sorts only if the number of VERTICES changed, i.e. never (gfa-base.c:298-301);
	 * the appended complement arcs are ordered by gfa_cleanup() below */
}

static void fix_arc_len(gfa_t *g) /* gfa-base.c:212-230 */
{
	uint64_t k;
	for (k = 0; k < g->n_arc; ++k) {
		gfa_arc_t *a = &g->arc[k];
		uint32_t v = (uint32_t)(a->v_lv >> 32), w = a->w;
		const gfa_seg_t *sv = &g->seg[v>>1];
		if (!sv->del && sv->len < a->ov) {
			if (mg_verbose >= 2) fprintf(stderr, "[W] overlap length longer than segment length for '%s': %d > %d\n", sv->name, a->ov, sv->len);
			a->ov = sv->len;
		}
		if (sv->del || g->seg[w>>1].del) a->del = 1;
		else a->v_lv |= (uint64_t)(uint32_t)(sv->len - a->ov);
	}
}

static void cleanup(gfa_t *g) /* gfa-base.c:305-335 */
{
	uint64_t e, n;
	for (e = n = 0; e < g->n_arc; ++e) {
		uint32_t u = (uint32_t)(g->arc[e].v_lv >> 32), v = g->arc[e].w;
		if (!g->arc[e].del && !g->seg[u>>1].del && !g->seg[v>>1].del) g->arc[n++] = g->arc[e];
	}
	if (n < g->n_arc) { free(g->idx); g->idx = 0; }
	g->n_arc = n;
	if (!arc_is_sorted(g)) { arc_sort(g); free(g->idx); g->idx = 0; }
	if (g->idx == 0) arc_index(g);
}

static void finalize(gfa_t *g)
{
	uint32_t i;
	for (i = 0; i < g->n_seg; ++i) /* gfa_fix_no_seg */
		if (g->seg[i].len == 0) {
			g->seg[i].del = 1;
			if (mg_verbose >= 2) fprintf(stderr, "[W] segment '%s' is used on an L-line but not defined on an S-line\n", g->seg[i].name);
		}
	arc_sort(g);
	arc_index(g);
	fix_semi_arc(g);
	fix_symm_add(g);
	fix_arc_len(g);
	cleanup(g);
}

/* ---- line reader over zlib ---- */
typedef struct { gzFile fp; char *buf; int beg, end, eof; } lr_t;

static int lr_getline(lr_t *r, char **line, size_t *m)
{
	size_t l = 0;
	int got = 0;
	for (;;) {
		if (r->beg >= r->end) {
			if (r->eof) break;
			r->end = gzread(r->fp, r->buf, 1 << 20), r->beg = 0;
			if (r->end <= 0) { r->eof = 1, r->end = 0; break; }
		}
		{
			char *s = r->buf + r->beg, *nl = (char*)memchr(s, '\n', (size_t)(r->end - r->beg));
			size_t n = nl ? (size_t)(nl - s) : (size_t)(r->end - r->beg);
			if (l + n + 1 > *m) { *m = (l + n + 1) * 2; *line = (char*)realloc(*line, *m); }
			memcpy(*line + l, s, n); l += n; got = 1;
			r->beg += (int)n + (nl ? 1 : 0);
			if (nl) break;
		}
	}
	if (!got) return -1;
	if (l > 0 && (*line)[l-1] == '\r') --l;
	(*line)[l] = 0;
	return (int)l;
}

gfa_t *gfa_read(const char *fn)
{
	lr_t r;
	gfa_t *g;
	char *line = 0, *fa_seq = 0;
	size_t m_line = 0, l_fa = 0, m_fa = 0;
	int l, is_fa = 0;
	gfa_seg_t *fa_seg = 0;
	uint64_t lineno = 0;

	mga_tables_init();
	memset(&r, 0, sizeof r);
	r.fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
	if (r.fp == 0) return 0;
	r.buf = (char*)malloc(1 << 20);
	g = MGA_CALLOC(gfa_t, 1);
	g->h_names = MGA_CALLOC(smap_t, 1);
	g->h_snames = MGA_CALLOC(smap_t, 1);
	while ((l = lr_getline(&r, &line, &m_line)) >= 0) {
		int ret = 0;
		++lineno;
		if (l > 0 && line[0] == '>') { /* FASTA record: one segment "s<N>" per sequence (gfa-io.c:266-288,311-317) */
			char nm[32], *p;
			if (fa_seg) { fa_seg->seq = dup_str(fa_seq ? fa_seq : "", l_fa), fa_seg->len = (int32_t)l_fa; update_sseq(g, fa_seg); }
			is_fa = 1;
			for (p = line; *p && !isspace((unsigned char)*p); ++p) {}
			*p = 0;
			snprintf(nm, sizeof nm, "s%u", g->n_seg + 1);
			{ int32_t sid = add_seg(g, nm); fa_seg = &g->seg[sid]; } /* NB: add_seg may move g->seg */
			fa_seg->snid = add_sseq(g, line + 1);
			fa_seg->soff = fa_seg->rank = 0;
			l_fa = 0;
			continue;
		} else if (is_fa) {
			if (l >= 3 && line[1] == '\t') { /* back to GFA lines */
				if (fa_seg) { fa_seg->seq = dup_str(fa_seq ? fa_seq : "", l_fa), fa_seg->len = (int32_t)l_fa; update_sseq(g, fa_seg); }
				fa_seg = 0, is_fa = 0;
			} else {
				if (l_fa + l + 1 > m_fa) { m_fa = (l_fa + l + 1) * 2; fa_seq = (char*)realloc(fa_seq, m_fa); }
				memcpy(fa_seq + l_fa, line, (size_t)l); l_fa += l;
				continue;
			}
		}
		if (l < 3 || line[1] != '\t') continue;
		if (line[0] == 'S') ret = parse_S(g, line);
		else if (line[0] == 'L') ret = parse_L(g, line);
		if (ret < 0 && mg_verbose >= 1) fprintf(stderr, "[E] invalid %c-line at line %ld (error code %d)\n", line[0], (long)lineno, ret);
	}
	if (is_fa && fa_seg) { fa_seg->seq = dup_str(fa_seq ? fa_seq : "", l_fa), fa_seg->len = (int32_t)l_fa; update_sseq(g, fa_seg); }
	free(line); free(fa_seq); free(r.buf);
	gzclose(r.fp);
	finalize(g);
	return g;
}

void gfa_destroy(gfa_t *g)
{
	uint32_t i;
	uint64_t k;
	if (g == 0) return;
	for (i = 0; i < g->n_seg; ++i) { free(g->seg[i].name); free(g->seg[i].seq); free(g->seg[i].aux.aux); }
	for (i = 0; i < g->n_sseq; ++i) free(g->sseq[i].name);
	if (g->link_aux) for (k = 0; k < g->n_arc; ++k) free(g->link_aux[k].aux);
	smap_free((smap_t*)g->h_names); smap_free((smap_t*)g->h_snames);
	free(g->idx); free(g->seg); free(g->arc); free(g->link_aux); free(g->sseq);
	free(g);
}
