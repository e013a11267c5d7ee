/*
 * ORACLE restatement of miniwfa's exact mode as minigraph calls it: mwf_wfa_auto -> mwf_wfa_exact
 * (step = 0, max_iter = 1e8) -> mwf_wfa_core with traceback (miniwfa.c:380-435,603-615,824-828).
 * 2-piece affine gap WFA, global alignment, penalties x / o1,e1 / o2,e2 (match = 0).
 *
 * The reference keeps a ring of max_pen+1 slices with 17-wide NEG_INF padding (miniwfa.c:79-118);
 * here every score keeps its own slice and reads outside [lo,hi] (or from a negative score) give
 * NEG_INF, which is what the padding / the initial all-NEG_INF ring provide.
 *
 *   H[s][d]  furthest target index k reached on diagonal d = q - t with penalty s, ending in a match/mismatch state
 *   E1,E2    ... ending in an insertion to the query (consumes query: comes from diagonal d-1, same k)
 *   F1,F2    ... ending in a deletion (consumes target: comes from diagonal d+1, k+1)
 * Recurrence + traceback byte: miniwfa.c:281-308.  Band bookkeeping wf->lo/hi: miniwfa.c:323-324,
 * 412-414; periodic trimming every 256 scores: miniwfa.c:139-169,420.  Traceback: miniwfa.c:329-377.
 */
#include <stdlib.h>
#include <string.h>
#include "mgo.h"

#define NEG_INF (-0x40000000)

typedef struct {
	int32_t lo, hi;
	int32_t *H, *E1, *F1, *E2, *F2; /* indexed [d - lo] */
	uint8_t *tb;
} slice_t;

typedef struct { slice_t *a; int32_t n, m; } slices_t;

static slice_t *slice_new(slices_t *S, int32_t lo, int32_t hi)
{
	slice_t *p;
	int32_t w = hi - lo + 1;
	if (S->n == S->m) { S->m = S->m ? S->m * 2 : 64; S->a = (slice_t*)realloc(S->a, S->m * sizeof(slice_t)); }
	p = &S->a[S->n++];
	p->lo = lo, p->hi = hi;
	p->H = (int32_t*)malloc(5 * (size_t)w * sizeof(int32_t));
	p->E1 = p->H + w, p->F1 = p->E1 + w, p->E2 = p->F1 + w, p->F2 = p->E2 + w;
	p->tb = (uint8_t*)calloc(w, 1);
	return p;
}

static inline int32_t at(const slices_t *S, int32_t s, int which, int32_t d)
{
	const slice_t *p;
	if (s < 0) return NEG_INF;
	p = &S->a[s];
	if (d < p->lo || d > p->hi) return NEG_INF;
	return (which == 0 ? p->H : which == 1 ? p->E1 : which == 2 ? p->F1 : which == 3 ? p->E2 : p->F2)[d - p->lo];
}

static inline int in_matrix(int32_t d, int32_t k, int32_t tl, int32_t ql) /* good_diag, miniwfa.c:134-137 */
{
	return k >= -1 && k < tl && d + k >= -1 && d + k < ql;
}

static int diag_alive(const slices_t *S, int32_t s_top, int32_t n_ring, int32_t d, int32_t tl, int32_t ql)
{
	int32_t j, w;
	for (j = 0; j < n_ring; ++j) {
		int32_t s = s_top - j;
		const slice_t *p;
		if (s < 0) break; /* initial slices hold only NEG_INF */
		p = &S->a[s];
		if (d < p->lo || d > p->hi) continue;
		for (w = 0; w < 5; ++w)
			if (in_matrix(d, at(S, s, w, d), tl, ql)) return 1;
	}
	return 0;
}

static void push_op(uint32_t *c, int32_t cap, int32_t *n, int32_t op, int32_t len)
{
	if (*n > 0 && *n <= cap && (c[*n - 1] & 0xf) == (uint32_t)op) { c[*n - 1] += (uint32_t)len << 4; return; }
	if (*n < cap) c[*n] = (uint32_t)len << 4 | (uint32_t)op;
	++*n;
}

int32_t mgo_wfa_exact(const mgo_wfa_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs,
					  uint32_t *cigar, int32_t cap, int32_t *n_cigar, int64_t *n_iter_)
{
	slices_t S = {0, 0, 0};
	slice_t *p;
	int32_t s = 0, wlo = 0, whi = 0, n_ring, last_state = 0, stopped = 0, d, ret;
	int32_t oe1 = opt->o1 + opt->e1, oe2 = opt->o2 + opt->e2;
	int64_t n_iter = 0;

	n_ring = opt->x;
	if (n_ring < oe1) n_ring = oe1;
	if (n_ring < oe2) n_ring = oe2;
	++n_ring; /* max_pen + 1 slices in the reference ring */

	p = slice_new(&S, 0, 0); /* score 0: H[0] = -1, everything else unreachable (miniwfa.c:103-119) */
	p->H[0] = -1, p->E1[0] = p->F1[0] = p->E2[0] = p->F2[0] = NEG_INF;
	*n_cigar = 0;

	for (;;) {
		int32_t lo, hi, found = 0;
		p = &S.a[s];
		for (d = p->lo; d <= p->hi; ++d) { /* extend along exact matches (miniwfa.c:399-411) */
			int32_t k0 = p->H[d - p->lo], k = k0;
			if (k < -1 || d + k < -1 || k >= tl || d + k >= ql) continue;
			while (k + 1 < tl && d + k + 1 < ql && ts[k + 1] == qs[d + k + 1]) ++k;
			if (k == tl - 1 && d + k == ql - 1) {
				if (k == k0) last_state = p->tb[d - p->lo] & 7;
				found = 1;
				break;
			}
			p->H[d - p->lo] = k;
		}
		if (found) break;
		lo = wlo > -tl ? wlo - 1 : -tl;
		hi = whi < ql ? whi + 1 : ql;
		++s;
		p = slice_new(&S, lo, hi);
		for (d = lo; d <= hi; ++d) { /* wf_next_tb (miniwfa.c:281-308) */
			int32_t ho1l = at(&S, s - oe1, 0, d - 1), e1l = at(&S, s - opt->e1, 1, d - 1);
			int32_t ho2l = at(&S, s - oe2, 0, d - 1), e2l = at(&S, s - opt->e2, 3, d - 1);
			int32_t ho1r = at(&S, s - oe1, 0, d + 1), f1r = at(&S, s - opt->e1, 2, d + 1);
			int32_t ho2r = at(&S, s - oe2, 0, d + 1), f2r = at(&S, s - opt->e2, 4, d + 1);
			int32_t hx = at(&S, s - opt->x, 0, d);
			int32_t E1, E2, F1, F2, e, f, h, H;
			uint8_t bits = 0, ze, zf, z;
			if (!(ho1l >= e1l)) bits |= 0x08;
			E1 = ho1l >= e1l ? ho1l : e1l;
			if (!(ho2l >= e2l)) bits |= 0x20;
			E2 = ho2l >= e2l ? ho2l : e2l;
			ze = E1 >= E2 ? 1 : 3;
			e = E1 >= E2 ? E1 : E2;
			if (!(ho1r >= f1r)) bits |= 0x10;
			F1 = (ho1r >= f1r ? ho1r : f1r) + 1;
			if (!(ho2r >= f2r)) bits |= 0x40;
			F2 = (ho2r >= f2r ? ho2r : f2r) + 1;
			zf = F1 >= F2 ? 2 : 4;
			f = F1 >= F2 ? F1 : F2;
			z = e >= f ? ze : zf;
			h = e >= f ? e : f;
			if (hx + 1 >= h) z = 0;
			H = hx + 1 >= h ? hx + 1 : h;
			p->H[d - lo] = H, p->E1[d - lo] = E1, p->F1[d - lo] = F1, p->E2[d - lo] = E2, p->F2[d - lo] = F2;
			p->tb[d - lo] = bits | z;
		}
		if (p->H[0] >= -1 || p->E1[0] >= -1 || p->F1[0] >= -1 || p->E2[0] >= -1 || p->F2[0] >= -1) wlo = lo;
		if (p->H[hi-lo] >= -1 || p->E1[hi-lo] >= -1 || p->F1[hi-lo] >= -1 || p->E2[hi-lo] >= -1 || p->F2[hi-lo] >= -1) whi = hi;
		if ((s & 0xff) == 0) { /* wf_stripe_shrink (miniwfa.c:139-169): drop dead diagonals at both ends */
			for (d = wlo; d <= whi; ++d) if (diag_alive(&S, s, n_ring, d, tl, ql)) break;
			wlo = d;
			for (d = whi; d >= wlo; --d) if (diag_alive(&S, s, n_ring, d, tl, ql)) break;
			whi = d;
		}
		n_iter += hi - lo + 1;
		if (opt->max_iter > 0 && n_iter > opt->max_iter) { stopped = 1; break; }
	}
	if (n_iter_) *n_iter_ = n_iter;

	if (!stopped) { /* wf_traceback (miniwfa.c:329-377) */
		int32_t i = ql - 1, k = tl - 1, sc = s, last = last_state, n = 0, a, b;
		while (i >= 0 && k >= 0) {
			int32_t k0 = k, state, ext;
			uint8_t x;
			if (last == 0) {
				while (i >= 0 && k >= 0 && qs[i] == ts[k]) --i, --k;
				if (k0 - k > 0) push_op(cigar, cap, &n, 7, k0 - k);
				if (i < 0 || k < 0) break;
			}
			x = S.a[sc].tb[(i - k) - S.a[sc].lo];
			state = last == 0 ? (x & 7) : last;
			ext = state > 0 ? (x >> (state + 2) & 1) : 0;
			if (state == 0) push_op(cigar, cap, &n, 8, 1), --i, --k, sc -= opt->x;
			else if (state == 1) push_op(cigar, cap, &n, 1, 1), --i, sc -= ext ? opt->e1 : oe1;
			else if (state == 3) push_op(cigar, cap, &n, 1, 1), --i, sc -= ext ? opt->e2 : oe2;
			else if (state == 2) push_op(cigar, cap, &n, 2, 1), --k, sc -= ext ? opt->e1 : oe1;
			else push_op(cigar, cap, &n, 2, 1), --k, sc -= ext ? opt->e2 : oe2;
			last = state > 0 && ext ? state : 0;
		}
		if (i >= 0) push_op(cigar, cap, &n, 1, i + 1);
		else if (k >= 0) push_op(cigar, cap, &n, 2, k + 1);
		if (n <= cap)
			for (a = 0, b = n - 1; a < b; ++a, --b) { uint32_t t = cigar[a]; cigar[a] = cigar[b]; cigar[b] = t; }
		*n_cigar = n;
	}
	ret = stopped ? -1 : s;
	for (d = 0; d < S.n; ++d) free(S.a[d].H), free(S.a[d].tb);
	free(S.a);
	return ret;
}
