/*
 * gaf.c -- GAF/PAF text output of the graph chains of one read: mg_write_gaf (reference
 * format.c:121-291) with its integer/string formatter (mg_sprintf_lite, format.c:36-80) folded into
 * direct append helpers.  Byte-identical output is the parity surface of the whole path.
 */
#include <stdio.h>
#include <math.h>
#include <assert.h>
#include "mga_host.h"

static inline void ks_room(kstring_t *s, size_t extra)
{
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
	}
}
static inline void ks_c(kstring_t *s, char c) { ks_room(s, 1); s->s[s->l++] = c; s->s[s->l] = 0; }
static inline void ks_sn(kstring_t *s, const char *p, size_t n) { ks_room(s, n); memcpy(s->s + s->l, p, n); s->l += (unsigned)n; s->s[s->l] = 0; }
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
	s->s[s->l] = 0;
}
static void ks_path_piece(kstring_t *s, int rev, const char *name, int32_t st, int32_t en) /* "%c%s:%d-%d" */
{
	ks_c(s, "><"[rev]); ks_s(s, name); ks_c(s, ':'); ks_d(s, st); ks_c(s, '-'); ks_d(s, en);
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
	ks_room(s, 0); s->s[s->l] = 0;
}

void mg_write_gaf(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag, void *km)
{
	(void)km;
	s->l = 0;
	mga_write_gaf_append(s, g, gs, n_seg, qlens, qname, flag, 0);
}

/* does the path of chain p print as ONE piece of a rank-0 stable sequence, on its reverse strand?  (format.c:150-199: the
 * "compact" form; that is what flips rev_sign).  Same decisions as the printing loop below, without printing. */
int mga_gaf_chain_rev(const gfa_t *g, const mg_gchains_t *gs, const mg_gchain_t *p, uint64_t flag)
{
	int32_t j, last_pnid = -1, st = -1, en = -1, rev = -1, compact;
	if (flag & MG_M_VERTEX_COOR) return 0;
	compact = flag & MG_M_NO_COMP_PATH ? 0 : 1;
	for (j = 0; j < p->cnt; ++j) {
		const mg_llchain_t *q = &gs->lc[p->off + j];
		const gfa_seg_t *t = &g->seg[q->v>>1];
		if (t->snid < 0) {
			compact = 0;
			last_pnid = -1, st = -1, en = -1, rev = -1;
		} else {
			int cont = 0;
			if (last_pnid >= 0 && t->snid == last_pnid && (int32_t)(q->v&1) == rev) {
				if (!(q->v&1)) { if (t->soff == en) en = t->soff + t->len, cont = 1; }
				else { if (t->soff + t->len == st) st = t->soff, cont = 1; }
			}
			if (cont == 0) {
				if (last_pnid >= 0) compact = 0;
				last_pnid = t->snid, rev = q->v&1, st = t->soff, en = st + t->len;
			}
		}
	}
	if (last_pnid >= 0) {
		if (g->sseq[last_pnid].rank != 0 || g->sseq[last_pnid].min != 0) compact = 0;
	} else compact = 0;
	return compact && (gs->lc[p->off].v&1);
}

/* the body of mg_write_gaf, appending to s (the batch formatter writes the lines of many reads into one buffer) */
void mga_write_gaf_append(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag,
						  const mga_chain_text_t *txt) /* txt: per chain, statistics + cg/ds text produced by the device (k_text.hip), or NULL */
{
	int32_t i, j, qlen, rev_sign = 0; /* rev_sign is deliberately NOT reset per chain (format.c:123) */
	const size_t l0 = s->l;
	for (i = 0, qlen = 0; i < n_seg; ++i) qlen += qlens[i];
	if ((gs == 0 || gs->n_gc == 0) && (flag & MG_M_SHOW_UNMAP)) {
		ks_s(s, qname);
		if ((flag & MG_M_FRAG_MERGE) && n_seg == 2 && s->l > l0 + 2 && s->s[s->l-1] == '1' && s->s[s->l-2] == '/') s->l -= 2;
		ks_c(s, '\t'); ks_d(s, qlen); ks_s(s, "\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0\n");
		return;
	}
	if (gs == 0) return;
	for (i = 0; i < gs->n_gc; ++i) {
		const mg_gchain_t *p = &gs->gc[i];
		int32_t sign_pos, compact;
		if (p->id != p->parent && !(flag & MG_M_PRINT_2ND)) continue;
		if (p->cnt == 0) continue;
		ks_s(s, qname);
		if ((flag & MG_M_FRAG_MERGE) && n_seg == 2 && s->l > l0 + 2 && s->s[s->l-1] == '1' && s->s[s->l-2] == '/') s->l -= 2;
		ks_c(s, '\t'); ks_d(s, qlen); ks_c(s, '\t'); ks_d(s, p->qs); ks_c(s, '\t'); ks_d(s, p->qe); ks_s(s, "\t+\t");
		sign_pos = (int32_t)s->l - 2;
		if (flag & MG_M_VERTEX_COOR) {
			compact = 0;
			for (j = 0; j < p->cnt; ++j) {
				const mg_llchain_t *q = &gs->lc[p->off + j];
				ks_c(s, "><"[q->v&1]); ks_s(s, g->seg[q->v>>1].name);
			}
		} else { /* stable coordinates: merge consecutive pieces of one stable sequence (format.c:150-199) */
			int32_t last_pnid = -1, st = -1, en = -1, rev = -1;
			compact = flag & MG_M_NO_COMP_PATH ? 0 : 1;
			for (j = 0; j < p->cnt; ++j) {
				const mg_llchain_t *q = &gs->lc[p->off + j];
				const gfa_seg_t *t = &g->seg[q->v>>1];
				if (t->snid < 0) {
					compact = 0;
					if (last_pnid >= 0) ks_path_piece(s, rev, g->sseq[last_pnid].name, st, en);
					last_pnid = -1, st = -1, en = -1, rev = -1;
					ks_c(s, "><"[q->v&1]); ks_s(s, t->name);
				} else {
					int cont = 0;
					if (last_pnid >= 0 && t->snid == last_pnid && (int32_t)(q->v&1) == rev) {
						if (!(q->v&1)) { if (t->soff == en) en = t->soff + t->len, cont = 1; }
						else { if (t->soff + t->len == st) st = t->soff, cont = 1; }
					}
					if (cont == 0) {
						if (last_pnid >= 0) compact = 0;
						if (last_pnid >= 0) ks_path_piece(s, rev, g->sseq[last_pnid].name, st, en);
						last_pnid = t->snid, rev = q->v&1, st = t->soff, en = st + t->len;
					}
				}
			}
			if (last_pnid >= 0) {
				if (g->sseq[last_pnid].rank != 0 || g->sseq[last_pnid].min != 0) compact = 0;
				if (!compact) ks_path_piece(s, rev, g->sseq[last_pnid].name, st, en);
			} else compact = 0;
		}
		if (compact) {
			int32_t rev = gs->lc[p->off].v&1;
			const gfa_seg_t *t = &g->seg[gs->lc[rev ? p->off + p->cnt - 1 : p->off].v>>1];
			const gfa_sseq_t *ps = &g->sseq[t->snid];
			ks_s(s, ps->name); ks_c(s, '\t'); ks_d(s, ps->max); ks_c(s, '\t');
			if (rev) {
				rev_sign = 1;
				s->s[sign_pos] = '-';
				ks_d(s, t->soff + (p->plen - p->pe)); ks_c(s, '\t'); ks_d(s, t->soff + (p->plen - p->ps));
			} else { ks_d(s, t->soff + p->ps); ks_c(s, '\t'); ks_d(s, t->soff + p->pe); }
		} else { ks_c(s, '\t'); ks_d(s, p->plen); ks_c(s, '\t'); ks_d(s, p->ps); ks_c(s, '\t'); ks_d(s, p->pe); }
		const mga_chain_text_t *tx = txt && txt[i].cg ? &txt[i] : 0;
		const int32_t a_mlen = tx ? tx->mlen : p->p ? p->p->mlen : p->mlen, a_blen = tx ? tx->blen : p->p ? p->p->blen : p->blen;
		ks_c(s, '\t'); ks_d(s, a_mlen); ks_c(s, '\t'); ks_d(s, a_blen); ks_c(s, '\t'); ks_d(s, (int32_t)p->mapq);
		ks_s(s, "\ttp:A:"); ks_c(s, p->id == p->parent ? 'P' : 'S');
		if (p->p || tx) { ks_s(s, "\tNM:i:"); ks_d(s, a_blen - a_mlen); }
		ks_s(s, "\tcm:i:"); ks_d(s, p->n_anchor); ks_s(s, "\ts1:i:"); ks_d(s, p->score); ks_s(s, "\ts2:i:"); ks_d(s, p->subsc);
		if (p->div >= 0.0f && p->div <= 1.0f) {
			char buf[16];
			if (p->div == 0.0f) buf[0] = '0', buf[1] = 0;
			else snprintf(buf, 16, "%.4f", p->div);
			ks_s(s, "\tdv:f:"); ks_s(s, buf);
		}
		if (n_seg > 1) {
			ks_s(s, "\tql:B:i");
			for (j = 0; j < n_seg; ++j) { ks_c(s, ','); ks_d(s, qlens[j]); }
		}
		if (tx) { /* both strings arrive in print order (already reversed when rev_sign is set) */
			ks_s(s, "\tcg:Z:"); ks_sn(s, tx->cg, (size_t)tx->cg_len);
			ks_s(s, "\tds:Z:"); ks_sn(s, tx->ds, (size_t)tx->ds_len);
		}
		if (p->p && !tx) {
			const int32_t nc = p->p->n_cigar;
			char *w;
			ks_s(s, "\tcg:Z:");
			ks_room(s, (size_t)nc * 12); /* <= 10 digits + operator per entry: one reservation, raw writes */
			w = s->s + s->l;
			for (j = 0; j < nc; ++j) {
				const uint64_t c = p->p->cigar[rev_sign ? nc - 1 - j : j];
				uint32_t x = (uint32_t)(c >> 4);
				if (x < 10) *w++ = (char)('0' + x);
				else if (x < 100) { *w++ = (char)('0' + x / 10); *w++ = (char)('0' + x % 10); }
				else {
					char buf[12];
					int l = 0;
					do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
					while (l > 0) *w++ = buf[--l];
				}
				*w++ = "MIDNSHP=XB"[c & 0xf];
			}
			s->l = (unsigned)(w - s->s);
			s->s[s->l] = 0;
		}
		if (p->ds.ds && !tx) {
			ks_s(s, "\tds:Z:");
			if (rev_sign) { /* reverse-complement the difference string entry by entry (format.c:217-241) */
				const char *ds = p->ds.ds;
				int32_t ii, jj;
				char *w;
				ks_room(s, (size_t)p->ds.len + 1); /* the reversed string has the same length */
				w = s->s + s->l;
				for (ii = p->ds.n_off - 1; ii >= 0; --ii) {
					int32_t off = p->ds.off[ii], en = ii < p->ds.n_off - 1 ? p->ds.off[ii+1] : p->ds.len;
					*w++ = ds[off];
					if (ds[off] == ':') { memcpy(w, ds + off + 1, (size_t)(en - off - 1)); w += en - off - 1; }
					else if (ds[off] == '*') { for (jj = off + 1; jj < en; ++jj) *w++ = (char)mga_comp_table[(uint8_t)ds[jj]]; }
					else {
						for (jj = en - 1; jj >= off + 1; --jj) {
							if (ds[jj] == '[') *w++ = ']';
							else if (ds[jj] == ']') *w++ = '[';
							else *w++ = (char)mga_comp_table[(uint8_t)ds[jj]];
						}
					}
				}
				s->l = (unsigned)(w - s->s);
				s->s[s->l] = 0;
			} else ks_sn(s, p->ds.ds, (size_t)p->ds.len);
		}
		ks_c(s, '\n');
		if ((mg_dbg_flag & 0x8) || (flag & MG_M_WRITE_LCHAIN)) { /* per-vertex lines, -S / --write-mz (format.c:252-289) */
			char buf[16];
			for (j = 0; j < p->cnt; ++j) {
				const mg_llchain_t *lc = &gs->lc[p->off + j];
				ks_s(s, "*\t"); ks_c(s, "><"[lc->v&1]); ks_s(s, g->seg[lc->v>>1].name); ks_c(s, '\t'); ks_d(s, g->seg[lc->v>>1].len); ks_c(s, '\t'); ks_d(s, lc->cnt);
				if (lc->cnt > 0) {
					double div;
					int32_t q_span = (int32_t)(gs->a[lc->off].y >> 32 & 0xff);
					int32_t n = (int32_t)(gs->a[lc->off + lc->cnt - 1].x >> 32) - (int32_t)(gs->a[lc->off].x >> 32) + 1;
					div = n == lc->cnt ? 0.0 : (n > lc->cnt ? log((double)n / lc->cnt) : log((double)lc->cnt / n)) / q_span;
					if (div == 0.0) buf[0] = '0', buf[1] = 0;
					else snprintf(buf, 16, "%.4f", div);
					ks_c(s, '\t'); ks_s(s, buf);
					ks_c(s, '\t'); ks_d(s, (int32_t)gs->a[lc->off].x + 1 - q_span); ks_c(s, '\t'); ks_d(s, (int32_t)gs->a[lc->off + lc->cnt - 1].x + 1);
					ks_c(s, '\t'); ks_d(s, (int32_t)gs->a[lc->off].y + 1 - q_span); ks_c(s, '\t'); ks_d(s, (int32_t)gs->a[lc->off + lc->cnt - 1].y + 1);
					if (flag & MG_M_WRITE_MZ) {
						int32_t ii, last = (int32_t)gs->a[lc->off].x + 1 - q_span;
						ks_c(s, '\t'); ks_d(s, q_span); ks_c(s, '\t');
						for (ii = 1; ii < lc->cnt; ++ii) {
							int32_t x = (int32_t)gs->a[lc->off + ii].x + 1 - q_span;
							if (ii > 1) ks_c(s, ',');
							ks_d(s, x - last); last = x;
						}
						last = (int32_t)gs->a[lc->off].y + 1 - q_span;
						ks_c(s, '\t');
						for (ii = 1; ii < lc->cnt; ++ii) {
							int32_t x = (int32_t)gs->a[lc->off + ii].y + 1 - q_span;
							if (ii > 1) ks_c(s, ',');
							ks_d(s, x - last); last = x;
						}
					}
				}
				ks_c(s, '\n');
			}
		}
	}
}
