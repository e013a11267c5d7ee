/*
 * gaf.c -- GAF/PAF text output of the graph chains of one read: mg_write_gaf (reference
 * format.c:121-291) with its integer/string formatter (mg_sprintf_lite, format.c:36-80) folded into
 * direct append helpers.  Byte-identical output is the parity surface of the whole path.
 */
#include <stdio.h>
#include <math.h>
#include <assert.h>
#include "mga_host.h"

/* A kstring_t whose capacity says MGA_KS_WINDOW is a WINDOW into somebody else's buffer (round 5: a chunk's GAF lines written straight to their place in the job's output,
 * mapper.c): it is never reallocated and never NUL-terminated -- the byte behind a piece belongs to the next piece, which another thread may be writing. */
#define KS_TERM(s) do { if ((s)->m != MGA_KS_WINDOW) (s)->s[(s)->l] = 0; } while (0)
static __thread size_t ks_win_limit = 0; /* bytes the calling thread's current window holds (mga_gaf_window_limit): a write past it lands in the next thread's piece */
void mga_gaf_window_limit(size_t bytes) { ks_win_limit = bytes; }
static void ks_win_overrun(const char *where, size_t at, size_t n)
{ /* the measuring pass and the writing pass run the same formatter: cannot happen, and must not go unnoticed */
	fprintf(stderr, "[E::%s] GAF window of %zu bytes overrun at %zu + %zu\n", where, ks_win_limit, at, n);
	abort();
}
/* window mode: the formatter reserves UPPER BOUNDS (ks_room) and then writes raw, so a window is checked where sizes are exact -- before the one large copy of a line (the
 * device's cg / ds text: KS_FITS) and right after every raw block (KS_WROTE: an overrun is at most one line header old when the process stops) */
#define KS_FITS(s, n) do { if ((s)->m == MGA_KS_WINDOW && (size_t)(s)->l + (size_t)(n) > ks_win_limit) ks_win_overrun(__func__, (s)->l, (n)); } while (0)
#define KS_WROTE(s) do { if ((s)->m == MGA_KS_WINDOW && (size_t)(s)->l > ks_win_limit) ks_win_overrun(__func__, (s)->l, 0); } while (0)
static inline void ks_room(kstring_t *s, size_t extra)
{
	if (s->m == MGA_KS_WINDOW) return; /* (never grown: see KS_FITS / KS_WROTE) */
	if (s->l + extra + 1 > s->m) {
		size_t m = s->l + extra + 1;
		if (m > 0xfffffff0u) { /* kstring_t (mgpriv.h:31-37) counts in 32 bits: fail loudly instead of wrapping.  A piece is one thread's share of a 16384-read chunk;
		                        * with the default -K 500M it stays far below this */
			fprintf(stderr, "[E::%s] more than 4 GB of GAF text in one output piece: use a smaller -K or more threads\n", __func__);
			abort();
		}
		m += m >> 1;
		if (m > 0xfffffff0u) m = 0xfffffff0u;
		s->m = (unsigned)(m < 64 ? 64 : m);
		s->s = (char*)realloc(s->s, s->m);
		if (s->s == 0) { fprintf(stderr, "[E::%s] out of memory (%u bytes of GAF text)\n", __func__, s->m); abort(); }
	}
}
static inline void ks_c(kstring_t *s, char c) { ks_room(s, 1); s->s[s->l++] = c; KS_TERM(s); }
static inline void ks_sn(kstring_t *s, const char *p, size_t n) { ks_room(s, n); memcpy(s->s + s->l, p, n); s->l += (unsigned)n; KS_TERM(s); }
static inline void ks_s(kstring_t *s, const char *p) { ks_sn(s, p, strlen(p)); }
static inline void ks_d(kstring_t *s, int32_t c)
{
	char buf[16];
	int l = 0;
	unsigned x = c >= 0 ? (unsigned)c : (unsigned)-c;
	do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
	if (c < 0) buf[l++] = '-';
	ks_room(s, (size_t)l);
	while (l > 0) s->s[s->l++] = buf[--l];
	KS_TERM(s);
}
/* mgpriv.h:118 / format.c:36-80: the reference's light formatter (%d %u %s %c only), exported because its consumers of mg_gchains_t
 * (asm-call.c:122-137, --call) print with it; appends to s like the original */
#include <stdarg.h>
void mg_sprintf_lite(kstring_t *s, const char *fmt, ...)
{
	va_list ap;
	const char *p;
	va_start(ap, fmt);
	for (p = fmt; *p; ++p) {
		if (*p != '%') { ks_c(s, *p); continue; }
		++p;
		if (*p == 'd') ks_d(s, va_arg(ap, int));
		else if (*p == 'u') {
			char buf[16];
			int l = 0;
			uint32_t x = va_arg(ap, uint32_t);
			do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
			while (l > 0) ks_c(s, buf[--l]);
		} else if (*p == 's') ks_s(s, va_arg(ap, const char*));
		else if (*p == 'c') ks_c(s, (char)va_arg(ap, int));
		else abort(); /* format.c:68 */
	}
	va_end(ap);
	ks_room(s, 0); KS_TERM(s);
}

void mg_write_gaf(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag, void *km)
{
	(void)km;
	s->l = 0;
	mga_write_gaf_append(s, g, gs, n_seg, qlens, qname, flag, 0);
}

/* ---- a GAF line in two phases ---------------------------------------------------------------------------------------------------------
 * The path column of a chain (format.c:140-199) is decided BEFORE anything is printed: the walk is folded into PIECES -- a vertex printed by name,
 * or a maximal run of vertices that continue each other on one stable sequence, printed as an interval of it -- and the line's layout (the
 * one-interval "compact" form with the strand in column 5, or the general form) follows from the piece list.  Printing is then a run of raw writes
 * into space reserved once per line.  mga_gaf_chain_rev(), which only needs the layout, shares the first phase. */
typedef struct { int32_t snid, rev, st, en; const char *name; } gaf_piece_t; /* snid < 0: a vertex by name (rev = its strand); else the interval [st, en) of stable sequence snid */
typedef struct { gaf_piece_t *a; int32_t n, m; gaf_piece_t fixed[16]; } gaf_walk_t;

static inline gaf_piece_t *walk_new_piece(gaf_walk_t *w)
{
	if (w->n == w->m) {
		w->m = w->m ? w->m * 2 : 16;
		if (w->a == w->fixed || w->a == 0) { gaf_piece_t *q = (gaf_piece_t*)malloc((size_t)w->m * sizeof *q); memcpy(q, w->fixed, (size_t)w->n * sizeof *q); w->a = q; }
		else w->a = (gaf_piece_t*)realloc(w->a, (size_t)w->m * sizeof *w->a);
	}
	return &w->a[w->n++];
}

/* phase 1: the pieces of the walk lc[0..cnt); returns 1 when the line takes the compact form (one piece, on a rank-0 stable sequence that starts at 0) */
static int walk_fold(const gfa_t *g, const mg_llchain_t *lc, int32_t cnt, uint64_t flag, gaf_walk_t *w)
{
	int32_t j;
	gaf_piece_t *cur = 0; /* the open interval, if the last vertex was on a stable sequence */
	w->a = w->fixed, w->n = 0, w->m = 16;
	for (j = 0; j < cnt; ++j) {
		const uint32_t v = lc[j].v;
		const gfa_seg_t *t = &g->seg[v >> 1];
		const int32_t rev = (int32_t)(v & 1);
		if ((flag & MG_M_VERTEX_COOR) || t->snid < 0) { /* by name */
			gaf_piece_t *q = walk_new_piece(w);
			q->snid = -1, q->rev = rev, q->st = q->en = 0, q->name = t->name;
			cur = 0;
			continue;
		}
		/* does this vertex continue the open interval?  forward: it starts where the interval ends; reverse: it ends where the interval starts */
		if (cur && cur->snid == t->snid && cur->rev == rev && (rev ? t->soff + t->len == cur->st : t->soff == cur->en)) {
			if (rev) cur->st = t->soff; else cur->en = t->soff + t->len;
			continue;
		}
		cur = walk_new_piece(w);
		cur->snid = t->snid, cur->rev = rev, cur->st = t->soff, cur->en = t->soff + t->len, cur->name = g->sseq[t->snid].name;
	}
	if (flag & (MG_M_VERTEX_COOR | MG_M_NO_COMP_PATH)) return 0;
	return w->n == 1 && w->a[0].snid >= 0 && g->sseq[w->a[0].snid].rank == 0 && g->sseq[w->a[0].snid].min == 0;
}
static inline void walk_free(gaf_walk_t *w) { if (w->a != w->fixed) free(w->a); }

/* does chain p print in the compact form on the reverse strand?  (that is what turns rev_sign on for the rest of the read's lines, format.c:123,183) */
int mga_gaf_chain_rev(const gfa_t *g, const mg_gchains_t *gs, const mg_gchain_t *p, uint64_t flag)
{
	gaf_walk_t w;
	const int compact = walk_fold(g, gs->lc + p->off, p->cnt, flag, &w);
	walk_free(&w);
	return compact && (gs->lc[p->off].v & 1);
}

/* raw writers into reserved space */
static inline char *put_u(char *w, uint32_t x)
{
	char buf[12];
	int l = 0;
	do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
	while (l > 0) *w++ = buf[--l];
	return w;
}
static inline char *put_d(char *w, int32_t c) { if (c < 0) { *w++ = '-'; return put_u(w, (uint32_t)-(int64_t)c); } return put_u(w, (uint32_t)c); }
static inline char *put_s(char *w, const char *p) { const size_t n = strlen(p); memcpy(w, p, n); return w + n; }
static inline char *put_tab_d(char *w, int32_t c) { *w++ = '\t'; return put_d(w, c); }
static inline char *put_tag_d(char *w, const char *tag, int32_t c) { w = put_s(w, tag); return put_d(w, c); }
static inline char *put_div(char *w, double div) /* "%.4f", "0" for exactly zero (format.c:200-203,259-262) */
{
	if (div == 0.0) { *w++ = '0'; return w; }
	return w + snprintf(w, 16, "%.4f", div);
}

static size_t name_bytes(const gaf_walk_t *w) { size_t n = 0; int32_t k; for (k = 0; k < w->n; ++k) n += strlen(w->a[k].name) + 26; return n; }

/* the body of mg_write_gaf, appending to s (the batch formatter writes the lines of many reads into one buffer) */
void mga_write_gaf_append(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag,
						  const mga_chain_text_t *txt) /* txt: per chain, statistics + cg/ds text produced by the device (k_text.hip), or NULL */
{
	int32_t i, j, qlen = 0, rev_sign = 0; /* rev_sign is deliberately NOT reset per chain (format.c:123) */
	size_t qn = strlen(qname);
	char *w;
	for (i = 0; i < n_seg; ++i) qlen += qlens[i];
	if ((flag & MG_M_FRAG_MERGE) && n_seg == 2 && qn > 2 && qname[qn - 1] == '1' && qname[qn - 2] == '/') qn -= 2; /* paired reads print without the "/1" (format.c:128,138) */
	if (gs == 0 || gs->n_gc == 0) {
		if (flag & MG_M_SHOW_UNMAP) {
			ks_room(s, qn + 64);
			w = s->s + s->l;
			memcpy(w, qname, qn); w += qn;
			w = put_tab_d(w, qlen);
			w = put_s(w, "\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0\n");
			s->l = (unsigned)(w - s->s); KS_WROTE(s); KS_TERM(s);
		}
		return;
	}
	for (i = 0; i < gs->n_gc; ++i) {
		const mg_gchain_t *p = &gs->gc[i];
		const mga_chain_text_t *tx = txt && txt[i].cg ? &txt[i] : 0;
		gaf_walk_t walk;
		int compact, first_rev;
		if ((p->id != p->parent && !(flag & MG_M_PRINT_2ND)) || p->cnt == 0) continue;
		compact = walk_fold(g, gs->lc + p->off, p->cnt, flag, &walk);
		first_rev = (int)(gs->lc[p->off].v & 1);
		if (compact && first_rev) rev_sign = 1;
		/* columns 1-12 and the fixed tags: one reservation */
		ks_room(s, qn + name_bytes(&walk) + 320 + (size_t)n_seg * 12);
		w = s->s + s->l;
		memcpy(w, qname, qn); w += qn;
		w = put_tab_d(w, qlen); w = put_tab_d(w, p->qs); w = put_tab_d(w, p->qe);
		*w++ = '\t', *w++ = compact && first_rev ? '-' : '+', *w++ = '\t';
		if (compact) { /* the stable sequence, its length, the interval of the alignment on it */
			const gfa_seg_t *t = &g->seg[gs->lc[first_rev ? p->off + p->cnt - 1 : p->off].v >> 1]; /* the segment the path's coordinates count from */
			const int32_t beg = first_rev ? p->plen - p->pe : p->ps, end = first_rev ? p->plen - p->ps : p->pe;
			w = put_s(w, walk.a[0].name);
			w = put_tab_d(w, g->sseq[walk.a[0].snid].max); w = put_tab_d(w, t->soff + beg); w = put_tab_d(w, t->soff + end);
		} else {
			for (j = 0; j < walk.n; ++j) {
				const gaf_piece_t *q = &walk.a[j];
				*w++ = "><"[q->rev];
				w = put_s(w, q->name);
				if (q->snid >= 0) { *w++ = ':'; w = put_d(w, q->st); *w++ = '-'; w = put_d(w, q->en); }
			}
			w = put_tab_d(w, p->plen); w = put_tab_d(w, p->ps); w = put_tab_d(w, p->pe);
		}
		walk_free(&walk);
		{
			const int32_t mlen = tx ? tx->mlen : p->p ? p->p->mlen : p->mlen, blen = tx ? tx->blen : p->p ? p->p->blen : p->blen;
			w = put_tab_d(w, mlen); w = put_tab_d(w, blen); w = put_tab_d(w, (int32_t)p->mapq);
			w = put_s(w, "\ttp:A:"); *w++ = p->id == p->parent ? 'P' : 'S';
			if (p->p || tx) w = put_tag_d(w, "\tNM:i:", blen - mlen);
		}
		w = put_tag_d(w, "\tcm:i:", p->n_anchor); w = put_tag_d(w, "\ts1:i:", p->score); w = put_tag_d(w, "\ts2:i:", p->subsc);
		if (p->div >= 0.0f && p->div <= 1.0f) { w = put_s(w, "\tdv:f:"); w = put_div(w, p->div); }
		if (n_seg > 1) {
			w = put_s(w, "\tql:B:i");
			for (j = 0; j < n_seg; ++j) { *w++ = ','; w = put_d(w, qlens[j]); }
		}
		s->l = (unsigned)(w - s->s);
		KS_WROTE(s);
		/* the alignment: text from the device (both strings already in print order), or from the chain's own CIGAR / difference string */
		if (tx) {
			ks_room(s, (size_t)tx->cg_len + (size_t)tx->ds_len + 16);
			KS_FITS(s, (size_t)tx->cg_len + (size_t)tx->ds_len + 12);
			w = s->s + s->l;
			w = put_s(w, "\tcg:Z:"); memcpy(w, tx->cg, (size_t)tx->cg_len); w += tx->cg_len;
			w = put_s(w, "\tds:Z:"); memcpy(w, tx->ds, (size_t)tx->ds_len); w += tx->ds_len;
			s->l = (unsigned)(w - s->s);
		} else {
			if (p->p) { /* cg:Z, last operator first on a reverse-strand line */
				const int32_t nc = p->p->n_cigar;
				ks_room(s, (size_t)nc * 12 + 8); /* <= 10 digits + operator per entry */
				w = put_s(s->s + s->l, "\tcg:Z:");
				for (j = 0; j < nc; ++j) {
					const uint64_t c = p->p->cigar[rev_sign ? nc - 1 - j : j];
					w = put_u(w, (uint32_t)(c >> 4));
					*w++ = "MIDNSHP=XB"[c & 0xf];
				}
				s->l = (unsigned)(w - s->s);
			}
			if (p->ds.ds) {
				ks_room(s, (size_t)p->ds.len + 8); /* the reversed string has the same length */
				w = put_s(s->s + s->l, "\tds:Z:");
				if (!rev_sign) { memcpy(w, p->ds.ds, (size_t)p->ds.len); w += p->ds.len; }
				else { /* the other strand: entries last to first; ':' counts stay, '*' pairs are complemented, inserted / deleted runs are reverse-complemented
				        * with their brackets swapped (format.c:217-241) */
					const char *ds = p->ds.ds;
					int32_t e;
					for (e = p->ds.n_off - 1; e >= 0; --e) {
						const int32_t b = p->ds.off[e], end = e + 1 < p->ds.n_off ? p->ds.off[e + 1] : p->ds.len;
						int32_t k;
						*w++ = ds[b];
						if (ds[b] == ':') { memcpy(w, ds + b + 1, (size_t)(end - b - 1)); w += end - b - 1; }
						else if (ds[b] == '*') for (k = b + 1; k < end; ++k) *w++ = (char)mga_comp_table[(uint8_t)ds[k]];
						else for (k = end - 1; k > b; --k) *w++ = ds[k] == '[' ? ']' : ds[k] == ']' ? '[' : (char)mga_comp_table[(uint8_t)ds[k]];
					}
				}
				s->l = (unsigned)(w - s->s);
			}
		}
		KS_FITS(s, 1);
		ks_c(s, '\n');
		if ((mg_dbg_flag & 0x8) || (flag & MG_M_WRITE_LCHAIN)) { /* one line per vertex of the walk, -S / --write-mz (format.c:252-289) */
			for (j = 0; j < p->cnt; ++j) {
				const mg_llchain_t *lc = &gs->lc[p->off + j];
				const gfa_seg_t *t = &g->seg[lc->v >> 1];
				ks_room(s, strlen(t->name) + 160 + ((flag & MG_M_WRITE_MZ) ? (size_t)lc->cnt * 24 : 0));
				w = s->s + s->l;
				*w++ = '*', *w++ = '\t', *w++ = "><"[lc->v & 1];
				w = put_s(w, t->name); w = put_tab_d(w, t->len); w = put_tab_d(w, lc->cnt);
				if (lc->cnt > 0) {
					const mg128_t *a = gs->a + lc->off, *z = a + lc->cnt - 1;
					const int32_t span = (int32_t)(a->y >> 32 & 0xff);
					const int32_t n_mz = (int32_t)(z->x >> 32) - (int32_t)(a->x >> 32) + 1; /* minimizers of the read between the first and the last anchor */
					const double div = n_mz == lc->cnt ? 0.0 : (n_mz > lc->cnt ? log((double)n_mz / lc->cnt) : log((double)lc->cnt / n_mz)) / span;
					*w++ = '\t'; w = put_div(w, div);
					w = put_tab_d(w, (int32_t)a->x + 1 - span); w = put_tab_d(w, (int32_t)z->x + 1);
					w = put_tab_d(w, (int32_t)a->y + 1 - span); w = put_tab_d(w, (int32_t)z->y + 1);
					if (flag & MG_M_WRITE_MZ) { /* minimizer-to-minimizer distances on the target, then on the query */
						int32_t k, axis;
						w = put_tab_d(w, span);
						for (axis = 0; axis < 2; ++axis) {
							*w++ = '\t';
							for (k = 1; k < lc->cnt; ++k) {
								const int32_t d = axis == 0 ? (int32_t)a[k].x - (int32_t)a[k - 1].x : (int32_t)a[k].y - (int32_t)a[k - 1].y;
								if (k > 1) *w++ = ',';
								w = put_d(w, d);
							}
						}
					}
				}
				*w++ = '\n';
				s->l = (unsigned)(w - s->s);
				KS_WROTE(s);
			}
			KS_TERM(s);
		}
	}
}
