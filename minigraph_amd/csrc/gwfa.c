/*
 * gwfa.c -- graph wavefront edit distance between two anchors on different segments, with walk
 * traceback: gfa_ed_init / gfa_ed_step / gfa_ed_destroy as mg_gchain_gen's bridge_gwfa uses them
 * (reference gfa-ed.c:44-617, caller gchain1.c:349-381).
 *
 * Unit-cost edit distance of a query slice against all walks from (v0,off0) that end at (v1,off1).
 * Wavefront cells are "diagonals" (vertex, d = i - k) holding the furthest target offset k; crossing
 * a vertex end fans out over its arcs in arc order.  ~1 call per read under -x lr, branchy and
 * allocation-heavy: host code.  Everything that decides ties follows the reference: the order cells
 * are appended to the next wavefront, the klib sort of the out-of-order subset and the stable merge
 * (gfa-ed.c:143-171), "keep the first of the furthest" dedup (:174-190), the forbidden-interval filter
 * (:192-202), pruning every 16 steps (:286-307) and the visited-(vertex,query) set (:456-462).
 */
#include <stdio.h>
#include "hchain.h"

#define DSHIFT 0x40000000

typedef struct { uint64_t vd; int32_t k, len; uint32_t xo; int32_t t; } diag_t;
typedef struct { uint64_t vd0, vd1; } intv_t;
typedef struct { int32_t v, pre; } trace_t;

typedef struct { diag_t *a; size_t n, m; } diag_v;
typedef struct { intv_t *a; size_t n, m; } intv_v;

#define VPUSHP(type, vec, ptr) do { \
		if ((vec).n == (vec).m) { (vec).m = (vec).m ? (vec).m << 1 : 16; (vec).a = MGA_REALLOC(type, (vec).a, (vec).m); } \
		(ptr) = &(vec).a[(vec).n++]; \
	} while (0)
#define VRESERVE(type, vec, cap) do { if ((vec).m < (size_t)(cap)) { (vec).m = (cap); (vec).a = MGA_REALLOC(type, (vec).a, (vec).m); } } while (0)

static inline uint64_t mk_vd(uint32_t v, int32_t d) { return (uint64_t)v << 32 | (uint32_t)(DSHIFT + d); }

/* ---- u64 set / map with open addressing ---- */
typedef struct { uint64_t *k; int32_t *v; uint32_t cap, cnt; } u64map_t;

static int32_t *u64map_put(u64map_t *h, uint64_t key, int *absent)
{
	uint32_t i;
	if (h->cnt * 2 >= h->cap) {
		uint32_t ocap = h->cap, j;
		uint64_t *ok = h->k; int32_t *ov = h->v;
		h->cap = ocap ? ocap * 2 : 64;
		h->k = MGA_MALLOC(uint64_t, h->cap); h->v = MGA_MALLOC(int32_t, h->cap);
		for (j = 0; j < h->cap; ++j) h->k[j] = ~0ULL;
		for (j = 0; j < ocap; ++j)
			if (ok[j] != ~0ULL) {
				uint32_t q = (uint32_t)((ok[j] ^ ok[j] >> 29) * 0x9E3779B97F4A7C15ULL >> 40) & (h->cap - 1);
				while (h->k[q] != ~0ULL) q = (q + 1) & (h->cap - 1);
				h->k[q] = ok[j], h->v[q] = ov[j];
			}
		free(ok); free(ov);
	}
	i = (uint32_t)((key ^ key >> 29) * 0x9E3779B97F4A7C15ULL >> 40) & (h->cap - 1);
	while (h->k[i] != ~0ULL && h->k[i] != key) i = (i + 1) & (h->cap - 1);
	*absent = h->k[i] == ~0ULL;
	if (*absent) h->k[i] = key, ++h->cnt;
	return &h->v[i];
}
static void u64map_clear(u64map_t *h) { uint32_t j; for (j = 0; j < h->cap; ++j) h->k[j] = ~0ULL; h->cnt = 0; }
static void u64map_free(u64map_t *h) { free(h->k); free(h->v); }

typedef struct {
	const gfa_t *g;
	const gfa_edseq_t *es;
	int32_t ql;
	const char *q;
	int32_t max_chk, bw_dyn, max_lag;
	int64_t i_term;
	u64map_t ha, ht;       /* visited (vertex, query pos) of the current step ; traceback node dedup */
	intv_v intv, tmp, swap;
	diag_v ooo;
	diag_v wf[2], head; /* the two wavefronts (current / next, swapped every step) and the cells that sit on a vertex or query end: kept across steps */
	int cur;
	uint64_t *sort_key; int64_t *sort_perm; diag_t *sort_tmp; int32_t m_sort; /* diag_sort scratch */
	trace_t *tr; size_t n_tr, m_tr;
	int32_t s, end_tb;
	uint32_t end_v; int32_t end_off;
} gw_t;

static int32_t trace_push(gw_t *z, int32_t v, int32_t pre) /* gfa-ed.c:213-227 */
{
	int absent;
	int32_t *val = u64map_put(&z->ht, (uint64_t)(uint32_t)v << 32 | (uint32_t)pre, &absent);
	if (absent) {
		if (z->n_tr == z->m_tr) { z->m_tr = z->m_tr ? z->m_tr << 1 : 16; z->tr = MGA_REALLOC(trace_t, z->tr, z->m_tr); }
		z->tr[z->n_tr].v = v, z->tr[z->n_tr].pre = pre;
		*val = (int32_t)z->n_tr++;
	}
	return *val;
}

static inline void diag_push(diag_v *a, uint32_t v, int32_t d, int32_t k, uint32_t x, uint32_t ooo, int32_t t)
{
	diag_t *p;
	VPUSHP(diag_t, *a, p);
	p->vd = mk_vd(v, d), p->k = k, p->xo = x << 1 | ooo, p->t = t, p->len = 0;
}

static inline int diag_update(diag_t *p, uint32_t v, int32_t d, int32_t k, uint32_t x, uint32_t ooo, int32_t t) /* gfa-ed.c:120-131 */
{
	if (p->vd == mk_vd(v, d)) {
		if (!(p->k > k)) p->xo = x << 1 | ooo, p->t = t, p->k = k;
		return 0;
	}
	return 1;
}

/* furthest target offset reachable by exact matches from k on diagonal d of a vertex of length vl (gfa-ed.c:305-329) */
static inline int32_t extend1(int32_t d, int32_t k, int32_t vl, const char *ts, int32_t ql, const char *qs)
{
	int32_t max_k = (ql - d < vl ? ql - d : vl) - 1;
	const char *t = ts + 1, *q = qs + d + 1;
	while (k + 8 <= max_k) { /* 8 bases per compare while a whole block is inside both sequences (gfa-ed.c:312-322 does the same on padded copies) */
		uint64_t x, y;
		memcpy(&x, t + k, 8); memcpy(&y, q + k, 8);
		if (x != y) return k + (__builtin_ctzll(x ^ y) >> 3);
		k += 8;
	}
	while (k < max_k && t[k] == q[k]) ++k;
	return k;
}

static size_t intv_merge_adj(size_t n, intv_t *a) /* gfa-ed.c:69-82 */
{
	size_t i, k;
	uint64_t st, en;
	if (n == 0) return 0;
	st = a[0].vd0, en = a[0].vd1;
	for (i = 1, k = 0; i < n; ++i) {
		if (a[i].vd0 > en) { a[k].vd0 = st, a[k++].vd1 = en; st = a[i].vd0, en = a[i].vd1; }
		else en = en > a[i].vd1 ? en : a[i].vd1;
	}
	a[k].vd0 = st, a[k++].vd1 = en;
	return k;
}

static int cmp_intv(const void *a, const void *b)
{
	uint64_t x = ((const intv_t*)a)->vd0, y = ((const intv_t*)b)->vd0;
	return x < y ? -1 : x > y;
}

/* sort a[] by vd: in-order cells stay in place, the flagged subset goes through the klib sort, stable merge (gfa-ed.c:143-171) */
static void diag_sort(gw_t *z, int32_t n_a, diag_t *a)
{
	int32_t i, j, k, n_b, n_c = 0;
	diag_t *b, *c;
	VRESERVE(diag_t, z->ooo, n_a);
	for (i = 0; i < n_a; ++i) if (a[i].xo & 1) ++n_c;
	n_b = n_a - n_c;
	b = z->ooo.a, c = b + n_b;
	for (i = j = k = 0; i < n_a; ++i) { if (a[i].xo & 1) c[k++] = a[i]; else b[j++] = a[i]; }
	if (n_c > 1) { /* scratch kept in z: one call per step */
		uint64_t *key;
		int64_t *perm;
		diag_t *tmp;
		if (z->m_sort < n_c) {
			z->m_sort = n_c + (n_c >> 1) + 16;
			z->sort_key = MGA_REALLOC(uint64_t, z->sort_key, z->m_sort);
			z->sort_perm = MGA_REALLOC(int64_t, z->sort_perm, z->m_sort);
			z->sort_tmp = MGA_REALLOC(diag_t, z->sort_tmp, z->m_sort);
		}
		key = z->sort_key, perm = z->sort_perm, tmp = z->sort_tmp;
		for (i = 0; i < n_c; ++i) key[i] = c[i].vd;
		mga_ksort_perm(n_c, key, 8, perm);
		for (i = 0; i < n_c; ++i) tmp[i] = c[perm[i]];
		memcpy(c, tmp, (size_t)n_c * sizeof(diag_t));
	}
	for (k = 0; k < n_c; ++k) c[k].xo &= 0xfffffffeU;
	i = j = k = 0;
	while (i < n_b && j < n_c) { if (b[i].vd <= c[j].vd) a[k++] = b[i++]; else a[k++] = c[j++]; }
	while (i < n_b) a[k++] = b[i++];
	while (j < n_c) a[k++] = c[j++];
}

static int32_t dedup(gw_t *z, int32_t n_a, diag_t *a) /* gwf_dedup, gfa-ed.c:258-271 */
{
	int32_t i, n, st, sorted = 1;
	if (z->intv.n + z->tmp.n > 0) {
		size_t ii = 0, jj = 0, kk = 0;
		int tmp_sorted = 1;
		for (i = 1; i < (int32_t)z->tmp.n; ++i) if (z->tmp.a[i-1].vd0 > z->tmp.a[i].vd0) { tmp_sorted = 0; break; }
		if (!tmp_sorted) qsort(z->tmp.a, z->tmp.n, sizeof(intv_t), cmp_intv); /* ties are merged away below: any sort */
		VRESERVE(intv_t, z->swap, z->intv.n + 1);
		memcpy(z->swap.a, z->intv.a, z->intv.n * sizeof(intv_t)); z->swap.n = z->intv.n;
		VRESERVE(intv_t, z->intv, z->intv.n + z->tmp.n + 1);
		while (ii < z->swap.n && jj < z->tmp.n) {
			if (z->swap.a[ii].vd0 <= z->tmp.a[jj].vd0) z->intv.a[kk++] = z->swap.a[ii++];
			else z->intv.a[kk++] = z->tmp.a[jj++];
		}
		while (ii < z->swap.n) z->intv.a[kk++] = z->swap.a[ii++];
		while (jj < z->tmp.n) z->intv.a[kk++] = z->tmp.a[jj++];
		z->intv.n = intv_merge_adj(kk, z->intv.a);
	}
	for (i = 1; i < n_a; ++i) if (a[i-1].vd > a[i].vd) { sorted = 0; break; }
	if (!sorted) diag_sort(z, n_a, a);
	for (i = 1, st = 0, n = 0; i <= n_a; ++i) { /* keep the furthest cell of every (vertex, diagonal): the first of equals */
		if (i == n_a || a[i].vd != a[st].vd) {
			int32_t j, max_j = st;
			for (j = st + 1; j < i; ++j) if (a[max_j].k < a[j].k) max_j = j;
			a[n++] = a[max_j];
			st = i;
		}
	}
	n_a = n;
	if (z->intv.n > 0) { /* drop cells inside finished diagonals (gfa-ed.c:192-202) */
		int32_t ii = 0, jj = 0, kk = 0, n_b = (int32_t)z->intv.n;
		const intv_t *b = z->intv.a;
		while (ii < n_a && jj < n_b) {
			if (a[ii].vd >= b[jj].vd0 && a[ii].vd < b[jj].vd1) ++ii;
			else if (a[ii].vd >= b[jj].vd1) ++jj;
			else a[kk++] = a[ii++];
		}
		while (ii < n_a) a[kk++] = a[ii++];
		n_a = kk;
	}
	return n_a;
}

static int32_t prune(int32_t n_a, diag_t *a, uint32_t max_lag, int32_t bw_dyn) /* gfa-ed.c:286-307 */
{
	int32_t i, j, iq, dq, max_i = -1;
	uint32_t max_x = 0;
	for (i = 0; i < n_a; ++i) if (a[i].xo >> 1 > max_x) max_x = a[i].xo >> 1, max_i = i;
	iq = (int32_t)a[max_i].vd - DSHIFT + a[max_i].k;
	dq = (int32_t)(a[max_i].xo >> 1) - iq - iq;
	for (i = j = 0; i < n_a; ++i) {
		diag_t *p = &a[i];
		int32_t ip = (int32_t)p->vd - DSHIFT + p->k;
		int32_t dp = (int32_t)(p->xo >> 1) - ip - ip;
		int32_t w = dp > dq ? dp - dq : dq - dp;
		if (bw_dyn >= 0 && w > bw_dyn) continue;
		if ((p->xo >> 1) + max_lag < max_x) continue;
		a[j++] = *p;
	}
	return j;
}

/* Landau-Vishkin over a run of adjacent diagonals on one vertex (gfa-ed.c:331-403) */
static void extend_batch(gw_t *z, int32_t n, diag_t *a, diag_v *B, diag_v *A)
{
	int32_t j, m, v = (int32_t)(a->vd >> 32), vl = z->es[v].len;
	const char *ts = z->es[v].seq;
	diag_t *b;
	for (j = 0; j < n; ++j) {
		int32_t k = extend1((int32_t)a[j].vd - DSHIFT, a[j].k, vl, ts, z->ql, z->q);
		a[j].len = k - a[j].k;
		a[j].xo += (uint32_t)a[j].len << 2;
		a[j].k = k;
	}
	VRESERVE(diag_t, *B, B->n + n + 2);
	b = &B->a[B->n];
	b[0].vd = a[0].vd - 1, b[0].xo = a[0].xo + 2, b[0].k = a[0].k + 1, b[0].t = a[0].t;
	b[1].vd = a[0].vd;
	b[1].xo = n == 1 || a[0].k > a[1].k ? a[0].xo + 4 : a[1].xo + 2;
	b[1].t  = n == 1 || a[0].k > a[1].k ? a[0].t : a[1].t;
	b[1].k  = (n == 1 || a[0].k > a[1].k ? a[0].k : a[1].k) + 1;
	for (j = 1; j < n - 1; ++j) {
		uint32_t x = a[j-1].xo + 2;
		int32_t k = a[j-1].k, t = a[j-1].t;
		x = k > a[j].k + 1 ? x : a[j].xo + 4;
		t = k > a[j].k + 1 ? t : a[j].t;
		k = k > a[j].k + 1 ? k : a[j].k + 1;
		x = k > a[j+1].k + 1 ? x : a[j+1].xo + 2;
		t = k > a[j+1].k + 1 ? t : a[j+1].t;
		k = k > a[j+1].k + 1 ? k : a[j+1].k + 1;
		b[j+1].vd = a[j].vd, b[j+1].k = k, b[j+1].xo = x, b[j+1].t = t;
	}
	if (n >= 2) {
		b[n].vd = a[n-1].vd;
		b[n].xo = a[n-2].k > a[n-1].k + 1 ? a[n-2].xo + 2 : a[n-1].xo + 4;
		b[n].t  = a[n-2].k > a[n-1].k + 1 ? a[n-2].t : a[n-1].t;
		b[n].k  = a[n-2].k > a[n-1].k + 1 ? a[n-2].k : a[n-1].k + 1;
	}
	b[n+1].vd = a[n-1].vd + 1, b[n+1].xo = a[n-1].xo + 2, b[n+1].t = a[n-1].t, b[n+1].k = a[n-1].k;
	for (j = 0; j < n; ++j) { /* cells at a vertex / query end are handled one by one by the caller */
		diag_t *p = &a[j];
		if (p->k == vl - 1 || (int32_t)p->vd - DSHIFT + p->k == z->ql - 1) {
			diag_t *qq;
			p->xo |= 1;
			VPUSHP(diag_t, *A, qq);
			*qq = *p;
		}
	}
	for (j = 0, m = 0; j < n + 2; ++j) {
		diag_t *p = &b[j];
		int32_t d = (int32_t)p->vd - DSHIFT;
		if (d + p->k < z->ql && p->k < vl) b[m++] = *p;
		else if (p->k == vl) {
			intv_t *iv;
			VPUSHP(intv_t, z->tmp, iv);
			iv->vd0 = mk_vd((uint32_t)v, d), iv->vd1 = iv->vd0 + 1;
		}
	}
	B->n += m;
}

/* one edit-distance step: reads the current wavefront a[] (= z->wf[z->cur]), builds the next one in the other buffer and returns it, or NULL when (v1,off1) is reached (gfa-ed.c:405-507) */
static diag_t *step(gw_t *z, uint32_t v1, int32_t off1, int32_t *n_a_, diag_t *a)
{
	int32_t i, x, n = *n_a_, do_dedup = 1;
	size_t head = 0;
	const gfa_t *g = z->g;
	const gfa_edseq_t *es = z->es;
	diag_v *Bp = &z->wf[z->cur ^ 1], *Ap = &z->head;
#define A (*Ap)
#define B (*Bp)
	A.n = B.n = 0;

	z->end_v = (uint32_t)-1, z->end_off = z->end_tb = -1;
	z->tmp.n = 0;
	u64map_clear(&z->ha);
	VRESERVE(diag_t, B, (size_t)n * 2 + 4);
	for (x = 0, i = 1; i <= n; ++i)
		if (i == n || a[i].vd != a[i-1].vd + 1) { extend_batch(z, i - x, &a[x], &B, &A); x = i; }
	if (A.n == 0) do_dedup = 0;

	while (head < A.n) {
		diag_t t = A.a[head++];
		uint32_t v = (uint32_t)(t.vd >> 32), x0;
		int32_t ooo = t.xo & 1, d = (int32_t)t.vd - DSHIFT, k, qi, vl = es[v].len;
		k = extend1(d, t.k, vl, es[v].seq, z->ql, z->q);
		qi = k + d;
		x0 = (t.xo >> 1) + ((uint32_t)(k - t.k) << 1);
		if (k + 1 < vl && qi + 1 < z->ql) { /* middle of a vertex */
			int32_t push1 = 1, push2 = 1;
			if (B.n >= 2) push1 = diag_update(&B.a[B.n - 2], v, d - 1, k + 1, x0 + 1, ooo, t.t);
			if (B.n >= 1) push2 = diag_update(&B.a[B.n - 1], v, d,     k + 1, x0 + 2, ooo, t.t);
			if (push1)          diag_push(&B, v, d - 1, k + 1, x0 + 1, 1, t.t);
			if (push2 || push1) diag_push(&B, v, d,     k + 1, x0 + 2, 1, t.t);
			diag_push(&B, v, d + 1, k, x0 + 1, ooo, t.t);
		} else if (qi + 1 < z->ql) { /* end of the vertex, query not finished: fan out over the arcs */
			int32_t nv = (int32_t)gfa_arc_n(g, v), j, n_ext = 0, tw;
			const gfa_arc_t *av = gfa_arc_a(g, v);
			intv_t *iv;
			VPUSHP(intv_t, z->tmp, iv);
			iv->vd0 = mk_vd(v, d), iv->vd1 = iv->vd0 + 1;
			tw = trace_push(z, (int32_t)v, t.t);
			for (j = 0; j < nv; ++j) {
				uint32_t w = av[j].w;
				int32_t ol = av[j].ow;
				int absent;
				u64map_put(&z->ha, (uint64_t)w << 32 | (uint32_t)(qi + 1), &absent);
				if (z->q[qi + 1] == es[w].seq[ol]) {
					++n_ext;
					if (absent) {
						diag_t *p;
						VPUSHP(diag_t, A, p);
						p->vd = mk_vd(w, qi + 1 - ol), p->k = ol, p->xo = (x0 + 2) << 1 | 1, p->t = tw, p->len = 0;
					}
				} else if (absent) {
					diag_push(&B, w, qi - ol,     ol, x0 + 1, 1, tw);
					diag_push(&B, w, qi + 1 - ol, ol, x0 + 2, 1, tw);
				}
			}
			if (nv == 0 || n_ext != nv) diag_push(&B, v, d + 1, k, x0 + 1, 1, t.t);
		} else if (v1 == (uint32_t)-1 || (v == v1 && k == off1)) { /* query finished at the requested end */
			z->end_v = v, z->end_off = k, z->end_tb = t.t, *n_a_ = 0;
			return 0;
		} else if (k + 1 < vl) { /* query finished inside a vertex: delete the next target base */
			diag_push(&B, v, d - 1, k + 1, x0 + 1, ooo, t.t);
		} else if (v != v1) { /* query and vertex both finished, not the last vertex */
			int32_t nv = (int32_t)gfa_arc_n(g, v), j, tw;
			const gfa_arc_t *av = gfa_arc_a(g, v);
			tw = trace_push(z, (int32_t)v, t.t);
			for (j = 0; j < nv; ++j) diag_push(&B, av[j].w, qi - av[j].ow, av[j].ow, x0 + 1, 1, tw);
		}
	}
	*n_a_ = n = (int32_t)B.n;
	if (do_dedup) *n_a_ = n = dedup(z, n, B.a);
	if (z->max_lag > 0 && n > z->max_chk && ((z->s + 1) & 0xf) == 0) *n_a_ = n = prune(n, B.a, (uint32_t)z->max_lag, z->bw_dyn);
	z->cur ^= 1;
	return B.a;
#undef A
#undef B
}

int32_t mga_gwfa_bridge(const gfa_t *g, const gfa_edseq_t *es, int32_t ql, const char *q, uint32_t v0, int32_t off0, uint32_t v1, int32_t off1,
						int32_t max_lag, int32_t s_term, int32_t **path, int32_t *nv)
{
	gw_t z;
	diag_t *a;
	int32_t n_a = 1, ret = -1;
	int64_t n_iter = 0;
	*path = 0, *nv = 0;
	memset(&z, 0, sizeof z);
	z.g = g, z.es = es, z.ql = ql, z.q = q;
	z.max_chk = 1000, z.bw_dyn = 1000, z.max_lag = max_lag, z.i_term = 500000000LL; /* gchain1.c:361-363 */
	VRESERVE(diag_t, z.wf[0], 16);
	a = z.wf[0].a; memset(a, 0, sizeof(diag_t));
	a[0].vd = mk_vd(v0, -off0), a[0].k = off0 - 1, a[0].xo = 0;
	z.m_tr = 16, z.tr = MGA_MALLOC(trace_t, z.m_tr);
	z.tr[0].v = -1, z.tr[0].pre = -1, z.n_tr = 1, a[0].t = 0; /* the root of the traceback forest (gfa-ed.c:568); real nodes have v >= 0 */
	z.end_v = (uint32_t)-1, z.end_off = -1;
	while (n_a > 0) { /* gfa_ed_step, gfa-ed.c:576-596 */
		a = step(&z, v1, off1, &n_a, a);
		n_iter += n_a;
		if (z.end_off >= 0 || n_a == 0) break;
		if (s_term >= 0 && z.s >= s_term) break;
		if (z.i_term > 0 && n_iter > z.i_term) break;
		++z.s;
	}
	if (z.end_off >= 0) { /* gwf_traceback, gfa-ed.c:509-522 */
		int32_t i = z.end_tb, n = 1, k, *p;
		while (i >= 0 && z.tr[i].v >= 0) ++n, i = z.tr[i].pre;
		p = MGA_MALLOC(int32_t, n);
		i = z.end_tb, n = 0;
		p[n++] = (int32_t)z.end_v;
		while (i >= 0 && z.tr[i].v >= 0) p[n++] = z.tr[i].v, i = z.tr[i].pre;
		for (i = 0; i < n >> 1; ++i) k = p[i], p[i] = p[n - 1 - i], p[n - 1 - i] = k;
		*path = p, *nv = n;
	}
	ret = z.end_v != (uint32_t)-1 ? z.s : -1;
	free(z.wf[0].a); free(z.wf[1].a); free(z.head.a); free(z.sort_key); free(z.sort_perm); free(z.sort_tmp);
	u64map_free(&z.ha); u64map_free(&z.ht);
	free(z.intv.a); free(z.tmp.a); free(z.swap.a); free(z.ooo.a); free(z.tr);
	return ret;
}
