/*
 * rmq.c -- RMQ-tree linear chaining on the host: mg_lchain_rmq (reference lchain.c:221-372).
 *
 * Under -x lr this is the long-join rescue (map-algo.c:407-417, fires on roughly half of 10 kb reads);
 * under -x asm it is the primary chainer.  The algorithm inserts/erases one tree node per anchor and
 * its result depends on the tie-breaking of the range-minimum query, which in turn depends on the AVL
 * shape and on WHEN each node's cached subtree minimum is refreshed (krmq.h:151-205).  It is therefore
 * kept sequential per read (reads run in parallel on host threads) and the tree below performs the same
 * rebalancing steps and the same minimum refreshes as the reference tree, on an index-based node pool.
 */
#include <assert.h>
#include "mga_host.h"
#include "hchain.h"

typedef struct {
	int32_t y;
	int64_t i;
	double pri;
	int32_t c[2];   /* children (pool indices, -1 = none) */
	int32_t s;      /* node holding the minimum pri of this subtree (with the reference's tie behaviour) */
	int32_t size;
	int8_t bal;
} rq_node_t;

typedef struct {
	rq_node_t *a;
	int32_t n, m, free_head;
} rq_pool_t;

#define RQ_MAXD 64
#define ND(t, x) (*(rq_node_t*)((char*)(t)->a + (x))) /* handles are BYTE offsets into the pool: base + handle is one addressing mode, no multiply on the pointer-chasing path */
#define RQ_H(idx) ((int32_t)((idx) * (int32_t)sizeof(rq_node_t)))

static int32_t rq_alloc(rq_pool_t *t)
{
	int32_t x;
	if (t->free_head >= 0) { x = t->free_head; t->free_head = ND(t, x).c[0]; return x; }
	if (t->n == t->m) { t->m = t->m ? t->m + (t->m >> 1) : 256; t->a = MGA_REALLOC(rq_node_t, t->a, t->m); }
	return RQ_H(t->n++);
}
static void rq_release(rq_pool_t *t, int32_t x) { ND(t, x).c[0] = t->free_head; t->free_head = x; }

static inline int rq_cmp(const rq_pool_t *t, int32_t y, int64_t i, int32_t p) /* lc_elem_cmp, lchain.c:228 */
{
	const rq_node_t *q = &ND(t, p);
	return y < q->y ? -1 : y > q->y ? 1 : (i > q->i) - (i < q->i);
}
#define RQ_LT(t, a_, b_) (ND(t, a_).pri < ND(t, b_).pri)
static inline int32_t rq_csize(const rq_pool_t *t, int32_t p, int d) { int32_t c = ND(t, p).c[d]; return c < 0 ? 0 : ND(t, c).size; }

static inline void rq_refresh(rq_pool_t *t, int32_t p, int32_t q, int32_t r) /* krmq_update_min, krmq.h:153-156 */
{
	ND(t, p).s = (q < 0 || RQ_LT(t, p, ND(t, q).s)) ? p : ND(t, q).s;
	ND(t, p).s = (r < 0 || RQ_LT(t, ND(t, p).s, ND(t, r).s)) ? ND(t, p).s : ND(t, r).s;
}

static int32_t rq_rot1(rq_pool_t *t, int32_t p, int dir) /* krmq.h:158-169 */
{
	int opp = 1 - dir;
	int32_t q = ND(t, p).c[opp], s = ND(t, p).s, size_p = ND(t, p).size;
	ND(t, p).size -= ND(t, q).size - rq_csize(t, q, dir);
	ND(t, q).size = size_p;
	rq_refresh(t, p, ND(t, p).c[dir], ND(t, q).c[dir]);
	ND(t, q).s = s;
	ND(t, p).c[opp] = ND(t, q).c[dir];
	ND(t, q).c[dir] = p;
	return q;
}

static int32_t rq_rot2(rq_pool_t *t, int32_t p, int dir) /* krmq.h:171-192 */
{
	int opp = 1 - dir, b1;
	int32_t q = ND(t, p).c[opp], r = ND(t, q).c[dir], s = ND(t, p).s;
	int32_t size_x_dir = rq_csize(t, r, dir);
	ND(t, r).size = ND(t, p).size;
	ND(t, p).size -= ND(t, q).size - size_x_dir;
	ND(t, q).size -= size_x_dir + 1;
	rq_refresh(t, p, ND(t, p).c[dir], ND(t, r).c[dir]);
	rq_refresh(t, q, ND(t, q).c[opp], ND(t, r).c[opp]);
	ND(t, r).s = s;
	ND(t, p).c[opp] = ND(t, r).c[dir];
	ND(t, r).c[dir] = p;
	ND(t, q).c[dir] = ND(t, r).c[opp];
	ND(t, r).c[opp] = q;
	b1 = dir == 0 ? +1 : -1;
	if (ND(t, r).bal == b1) ND(t, q).bal = 0, ND(t, p).bal = (int8_t)-b1;
	else if (ND(t, r).bal == 0) ND(t, q).bal = ND(t, p).bal = 0;
	else ND(t, q).bal = (int8_t)b1, ND(t, p).bal = 0;
	ND(t, r).bal = 0;
	return r;
}

static void rq_insert(rq_pool_t *t, int32_t *root, int32_t x) /* krmq_insert, krmq.h:194-243; keys are unique here */
{
	unsigned char stack[RQ_MAXD];
	int32_t path[RQ_MAXD], bp = *root, bq = -1, p, q, r = -1;
	int i, which = 0, top = 0, plen = 0, b1;
	for (p = bp, q = bq; p >= 0; q = p, p = ND(t, p).c[which]) {
		int cmp = rq_cmp(t, ND(t, x).y, ND(t, x).i, p);
		assert(cmp != 0);
		if (ND(t, p).bal != 0) bq = q, bp = p, top = 0;
		stack[top++] = which = (cmp > 0);
		path[plen++] = p;
	}
	ND(t, x).bal = 0, ND(t, x).size = 1, ND(t, x).c[0] = ND(t, x).c[1] = -1, ND(t, x).s = x;
	if (q < 0) *root = x; else ND(t, q).c[which] = x;
	if (bp < 0) return;
	for (i = 0; i < plen; ++i) ++ND(t, path[i]).size;
	for (i = plen - 1; i >= 0; --i) {
		rq_refresh(t, path[i], ND(t, path[i]).c[0], ND(t, path[i]).c[1]);
		if (ND(t, path[i]).s != x) break;
	}
	for (p = bp, top = 0; p != x; p = ND(t, p).c[stack[top]], ++top)
		if (stack[top] == 0) --ND(t, p).bal; else ++ND(t, p).bal;
	if (ND(t, bp).bal > -2 && ND(t, bp).bal < 2) return;
	which = ND(t, bp).bal < 0;
	b1 = which == 0 ? +1 : -1;
	q = ND(t, bp).c[1 - which];
	if (ND(t, q).bal == b1) { r = rq_rot1(t, bp, which); ND(t, q).bal = ND(t, bp).bal = 0; }
	else r = rq_rot2(t, bp, which);
	if (bq < 0) *root = r; else ND(t, bq).c[bp != ND(t, bq).c[0]] = r;
}

/* erase the node with key (y,i); returns its pool index or -1.  krmq_erase, krmq.h:245-323.
 * Pool slot t->n (never a real node) serves as the reference's stack-allocated "fake" super-root. */
static int32_t rq_erase(rq_pool_t *t, int32_t *root, int32_t y, int64_t i)
{
	int32_t path[RQ_MAXD], p, fake;
	unsigned char dir[RQ_MAXD];
	int k, d = 0, cmp;
	if (t->n == t->m) { t->m += (t->m >> 1) + 16; t->a = MGA_REALLOC(rq_node_t, t->a, t->m); }
	fake = RQ_H(t->n);
	ND(t, fake) = ND(t, *root);
	ND(t, fake).c[0] = *root, ND(t, fake).c[1] = -1;
	for (cmp = -1, p = fake; cmp; cmp = rq_cmp(t, y, i, p)) {
		int which = cmp > 0;
		dir[d] = (unsigned char)which;
		path[d++] = p;
		p = ND(t, p).c[which];
		if (p < 0) return -1;
	}
	for (k = 1; k < d; ++k) --ND(t, path[k]).size;
	if (ND(t, p).c[1] < 0) {
		ND(t, path[d-1]).c[dir[d-1]] = ND(t, p).c[0];
	} else {
		int32_t q = ND(t, p).c[1];
		if (ND(t, q).c[0] < 0) {
			ND(t, q).c[0] = ND(t, p).c[0];
			ND(t, q).bal = ND(t, p).bal;
			ND(t, path[d-1]).c[dir[d-1]] = q;
			path[d] = q, dir[d++] = 1;
			ND(t, q).size = ND(t, p).size - 1;
		} else {
			int32_t r;
			int e = d++;
			for (;;) {
				dir[d] = 0;
				path[d++] = q;
				r = ND(t, q).c[0];
				if (ND(t, r).c[0] < 0) break;
				q = r;
			}
			ND(t, r).c[0] = ND(t, p).c[0];
			ND(t, q).c[0] = ND(t, r).c[1];
			ND(t, r).c[1] = ND(t, p).c[1];
			ND(t, r).bal = ND(t, p).bal;
			ND(t, path[e-1]).c[dir[e-1]] = r;
			path[e] = r, dir[e] = 1;
			for (k = e + 1; k < d; ++k) --ND(t, path[k]).size;
			ND(t, r).size = ND(t, p).size - 1;
		}
	}
	for (k = d - 1; k >= 0; --k) rq_refresh(t, path[k], ND(t, path[k]).c[0], ND(t, path[k]).c[1]);
	while (--d > 0) {
		int32_t q = path[d];
		int which = dir[d], other = 1 - which, b1 = 1, b2 = 2;
		if (which) b1 = -b1, b2 = -b2;
		ND(t, q).bal = (int8_t)(ND(t, q).bal + b1);
		if (ND(t, q).bal == b1) break;
		else if (ND(t, q).bal == b2) {
			int32_t r = ND(t, q).c[other];
			if (ND(t, r).bal == -b1) {
				ND(t, path[d-1]).c[dir[d-1]] = rq_rot2(t, q, which);
			} else {
				ND(t, path[d-1]).c[dir[d-1]] = rq_rot1(t, q, which);
				if (ND(t, r).bal == 0) { ND(t, r).bal = (int8_t)-b1; ND(t, q).bal = (int8_t)b1; break; }
				else ND(t, r).bal = ND(t, q).bal = 0;
			}
		}
	}
	*root = ND(t, fake).c[0];
	return p;
}

/* minimum-pri node with key in the CLOSED interval [(lo_y,lo_i), (hi_y,hi_i)]; krmq_rmq, krmq.h:110-149 */
static int32_t rq_rmq(const rq_pool_t *t, int32_t root, int32_t lo_y, int64_t lo_i, int32_t hi_y, int64_t hi_i)
{
	int32_t p = root, path[2][RQ_MAXD], min;
	int plen[2] = {0, 0}, pcmp[2][RQ_MAXD], i, cmp, lca;
	if (root < 0) return -1;
	while (p >= 0) {
		cmp = rq_cmp(t, lo_y, lo_i, p);
		path[0][plen[0]] = p, pcmp[0][plen[0]++] = cmp;
		if (cmp < 0) p = ND(t, p).c[0]; else if (cmp > 0) p = ND(t, p).c[1]; else break;
	}
	p = root;
	while (p >= 0) {
		cmp = rq_cmp(t, hi_y, hi_i, p);
		path[1][plen[1]] = p, pcmp[1][plen[1]++] = cmp;
		if (cmp < 0) p = ND(t, p).c[0]; else if (cmp > 0) p = ND(t, p).c[1]; else break;
	}
	for (i = 0; i < plen[0] && i < plen[1]; ++i)
		if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
	if (i == plen[0] || i == plen[1]) return -1;
	lca = i, min = path[0][lca];
	for (i = lca + 1; i < plen[0]; ++i)
		if (pcmp[0][i] <= 0) {
			int32_t c;
			if (RQ_LT(t, path[0][i], min)) min = path[0][i];
			c = ND(t, path[0][i]).c[1];
			if (c >= 0 && RQ_LT(t, ND(t, c).s, min)) min = ND(t, c).s;
		}
	for (i = lca + 1; i < plen[1]; ++i)
		if (pcmp[1][i] >= 0) {
			int32_t c;
			if (RQ_LT(t, path[1][i], min)) min = path[1][i];
			c = ND(t, path[1][i]).c[0];
			if (c >= 0 && RQ_LT(t, ND(t, c).s, min)) min = ND(t, c).s;
		}
	return min;
}

/* iterator: a root-to-node stack.  krmq_itr_find + krmq_itr_prev (krmq.h:325-362) positioned on the
 * node `x` that krmq_interval returned as the lower neighbour (greatest key <= query). */
typedef struct { int32_t stack[RQ_MAXD]; int top; } rq_itr_t;

static void rq_itr_seek(const rq_pool_t *t, int32_t root, int32_t x, rq_itr_t *it)
{
	int32_t p = root;
	it->top = -1;
	while (p >= 0) {
		int cmp = rq_cmp(t, ND(t, x).y, ND(t, x).i, p);
		it->stack[++it->top] = p;
		if (cmp < 0) p = ND(t, p).c[0]; else if (cmp > 0) p = ND(t, p).c[1]; else break;
	}
}

static int rq_itr_prev(const rq_pool_t *t, rq_itr_t *it)
{
	int32_t p;
	if (it->top < 0) return 0;
	p = ND(t, it->stack[it->top]).c[0];
	if (p >= 0) {
		for (; p >= 0; p = ND(t, p).c[1]) it->stack[++it->top] = p;
		return 1;
	}
	do { p = it->stack[it->top--]; } while (it->top >= 0 && p == ND(t, it->stack[it->top]).c[0]);
	return it->top < 0 ? 0 : 1;
}

/* greatest node with key <= (y,i), or -1: the `lower` output of krmq_interval (krmq.h:95-108) */
static int32_t rq_lower(const rq_pool_t *t, int32_t root, int32_t y, int64_t i)
{
	int32_t p = root, l = -1;
	while (p >= 0) {
		int cmp = rq_cmp(t, y, i, p);
		if (cmp < 0) p = ND(t, p).c[0];
		else if (cmp > 0) l = p, p = ND(t, p).c[1];
		else { l = p; break; }
	}
	return l;
}

static inline int32_t score_simple(const mg128_t *ai, const mg128_t *aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width) /* lchain.c:234-250 */
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq ? dr - dq : dq - dr;
	dg = dr < dq ? dr : dq;
	q_span = (int32_t)(aj->y >> 32 & 0xff);
	sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
		float log_pen = dd >= 1 ? mga_log2f((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin_pen + .5f * log_pen);
	}
	return sc;
}

/* MGA_RQ_TIE_STATS=1 (measurement aid, round 5): how often is the range-minimum query's answer TIED -- i.e. decided by the AVL tree's shape, which no order-independent
 * formulation (a device arg-min over the window) can reproduce?  Brute force over the window per query; counters per process. */
static int g_tie_on = -1;
static int64_t g_tie[8]; /* runs, runs with a tie, anchors, anchors of runs with a tie, queries with an answer, tied queries, sum of window sizes, longest run */
void mga_rq_tie_stats(int64_t *out, int reset) { int k; for (k = 0; k < 8; ++k) { out[k] = __atomic_load_n(&g_tie[k], __ATOMIC_RELAXED); if (reset) __atomic_store_n(&g_tie[k], 0, __ATOMIC_RELAXED); } }

/* Forward pass of mg_lchain_rmq (lchain.c:275-353) over anchors [beg,end) of the x-sorted array a[]: fills f, p, v and the skip
 * marks t (zeroed by the caller) with ABSOLUTE indices.  beg must be 0 or the first anchor of a (segment, strand) group: there
 * the sequential run has just erased every node from both trees (the `a[i].x>>32 != a[st].x>>32` clause of lchain.c:294,304), st,
 * st_inner and i0 all equal beg, and nothing else is carried over -- so runs of whole groups are independent work items, which
 * is what lets a contig of 10^7 anchors use every host thread (mapper.c: rmq phases). */
void mga_lchain_rmq_fwd(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, float pen_gap, float pen_skip,
						int64_t beg, int64_t end, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t)
{
	int32_t root = -1;
	int64_t i, i0, st = beg, st_inner = beg;
	rq_pool_t T = {0, 0, 0, -1};
	int64_t ts_run_beg = beg, ts_run_ties = 0, ts_q = 0, ts_tq = 0, ts_win = 0;
	if (g_tie_on < 0) g_tie_on = getenv("MGA_RQ_TIE_STATS") && atoi(getenv("MGA_RQ_TIE_STATS")) > 0;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	/* ONE tree (round 4).  The reference keeps a second tree for the inner window (lchain.c:286-291,303-312) and only ever asks it ORDER questions -- the largest key below
	 * (y - 1, n), then its predecessors (lchain.c:326-345) -- whose answers do not depend on a tree's shape.  Its key set is { j : st_inner <= j < i0 } (anchors are inserted
	 * and erased in index order, and st_inner never passes i0: an anchor of [i0, i) has a[i]'s own x, and at st_inner == i0 the set is empty, so neither the distance nor
	 * the size clause holds there), a subset of the outer tree's { j : st <= j < i0 } (st <= st_inner: whatever moves st -- distance beyond max_dist, or more than
	 * cap_rmq_size keys -- moves st_inner too, the inner distance being the smaller one and the two sets being equal when st == st_inner).  So the inner walk runs over the
	 * OUTER tree and passes over the keys with j < st_inner: the same anchors in the same order, with half the insertions and erasures. */
	for (i = i0 = beg; i < end; ++i) {
		int64_t max_j = -1;
		int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff), max_f = q_span, q;
		if (i0 < i && a[i0].x != a[i].x) { /* add in-range anchors (lchain.c:279-293) */
			int64_t j;
			for (j = i0; j < i; ++j) {
				int32_t x = rq_alloc(&T);
				ND(&T, x).y = (int32_t)a[j].y, ND(&T, x).i = j;
				ND(&T, x).pri = -(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				rq_insert(&T, &root, x);
			}
			i0 = i;
		}
		/* drop active chains out of range (lchain.c:294-302) */
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist || (root >= 0 ? ND(&T, root).size : 0) > cap_rmq_size)) {
			if (root >= 0) { /* (krmq_find + krmq_erase in the reference: rq_erase leaves the tree untouched when the key is absent) */
				q = rq_erase(&T, &root, (int32_t)a[st].y, st);
				if (q >= 0) rq_release(&T, q);
			}
			++st;
		}
		if (max_dist_inner > 0) /* lchain.c:303-312: the inner window's set is [st_inner, i0), its size i0 - st_inner */
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + max_dist_inner || (st_inner < i0 ? i0 - st_inner : 0) > cap_rmq_size)) ++st_inner;
		/* RMQ (lchain.c:313-352) */
		q = rq_rmq(&T, root, (int32_t)a[i].y - max_dist, INT32_MAX, (int32_t)a[i].y - 1, 0);
		if (g_tie_on) {
			if (i > beg && a[i].x >> 32 != a[i - 1].x >> 32) { /* a (segment, strand) run ends: account for it */
				__atomic_fetch_add(&g_tie[0], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[2], i - ts_run_beg, __ATOMIC_RELAXED);
				if (ts_run_ties) { __atomic_fetch_add(&g_tie[1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[3], i - ts_run_beg, __ATOMIC_RELAXED); }
				{ int64_t mx = __atomic_load_n(&g_tie[7], __ATOMIC_RELAXED); while (i - ts_run_beg > mx && !__atomic_compare_exchange_n(&g_tie[7], &mx, i - ts_run_beg, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
				ts_run_beg = i, ts_run_ties = 0;
			}
			if (q >= 0) {
				int64_t j, n_eq = 0, n_in = 0;
				const double pq = ND(&T, q).pri;
				for (j = st; j < i0; ++j) {
					const int32_t yj = (int32_t)a[j].y;
					/* closed key range [(y - max_dist, INT32_MAX), (y - 1, 0)] over keys (y_j, j): y - max_dist < y_j < y - 1, and y_j == y - 1 only for j == 0 */
					if (yj > (int32_t)a[i].y - max_dist && (yj < (int32_t)a[i].y - 1 || (yj == (int32_t)a[i].y - 1 && j <= 0))) {
						++n_in;
						if (-(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y)) == pq) ++n_eq;
					}
				}
				++ts_q, ts_win += n_in;
				if (n_eq > 1) ++ts_tq, ++ts_run_ties;
			}
		}
		if (q >= 0) {
			int32_t sc, exact, width, n_skip = 0;
			int64_t j = ND(&T, q).i;
			sc = f[j] + score_simple(&a[i], &a[j], pen_gap, pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) max_f = sc, max_j = j;
			if (!exact && max_dist_inner > 0 && st_inner < i0 && (int32_t)a[i].y > 0) {
				int32_t lo = rq_lower(&T, root, (int32_t)a[i].y - 1, end); /* (the reference's key index n: above every index in the tree) */
				if (lo >= 0) {
					rq_itr_t itr;
					rq_itr_seek(&T, root, lo, &itr);
					while (itr.top >= 0) {
						int32_t qq = itr.stack[itr.top];
						if (ND(&T, qq).y < (int32_t)a[i].y - max_dist_inner) break;
						j = ND(&T, qq).i;
						if (j >= st_inner) { /* a key of the inner window */
							sc = f[j] + score_simple(&a[i], &a[j], pen_gap, pen_skip, 0, &width);
							if (width <= bw) {
								if (sc > max_f) {
									max_f = sc, max_j = j;
									if (n_skip > 0) --n_skip;
								} else if (t[j] == (int32_t)i) {
									if (++n_skip > max_chn_skip) break;
								}
								if (p[j] >= 0) t[p[j]] = (int32_t)i;
							}
						}
						if (!rq_itr_prev(&T, &itr)) break;
					}
				}
			}
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
	}
	if (g_tie_on && end > beg) {
		__atomic_fetch_add(&g_tie[0], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[2], end - ts_run_beg, __ATOMIC_RELAXED);
		if (ts_run_ties) { __atomic_fetch_add(&g_tie[1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[3], end - ts_run_beg, __ATOMIC_RELAXED); }
		__atomic_fetch_add(&g_tie[4], ts_q, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[5], ts_tq, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tie[6], ts_win, __ATOMIC_RELAXED);
	}
	free(T.a);
}

/* backtracking + compaction after the forward pass over all of a[0..n) (lchain.c:355-371); frees f, p, v, t */
mg128_t *mga_lchain_rmq_finish2(int bw, int min_cnt, int min_sc, int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t, int *n_u_, uint64_t **u_, int keep_arrays)
{
	int32_t n_u, n_v;
	uint64_t *u;
	mg128_t *ret;
	u = mga_chain_backtrack(n, f, p, v, t, min_cnt, min_sc, bw, 0, &n_u, &n_v);
	*n_u_ = n_u, *u_ = u;
	if (!keep_arrays) { free(p); free(f); free(t); }
	if (n_u == 0) { if (!keep_arrays) free(v); return 0; }
	ret = mga_compact_a(n_u, u, n_v, v, a);
	if (!keep_arrays) free(v);
	return ret;
}
mg128_t *mga_lchain_rmq_finish(int bw, int min_cnt, int min_sc, int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t, int *n_u_, uint64_t **u_)
{
	return mga_lchain_rmq_finish2(bw, min_cnt, min_sc, n, a, f, p, v, t, n_u_, u_, 0);
}

/* in: n x-sorted anchors a[]; out: chains in u[] (malloc'ed, *n_u_ entries) and the compacted anchor array
 * (malloc'ed, returned).  Same contract as the reference except that a[] is not freed. */
mg128_t *mga_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
						float pen_gap, float pen_skip, int64_t n, const mg128_t *a, int *n_u_, uint64_t **u_)
{
	int32_t *f, *t, *v;
	int64_t *p;
	*u_ = 0, *n_u_ = 0;
	if (n == 0 || a == 0) return 0;
	p = MGA_MALLOC(int64_t, n); f = MGA_MALLOC(int32_t, n); v = MGA_MALLOC(int32_t, n); t = MGA_CALLOC(int32_t, n);
	mga_lchain_rmq_fwd(max_dist, max_dist_inner, bw, max_chn_skip, cap_rmq_size, pen_gap, pen_skip, 0, n, a, f, p, v, t);
	return mga_lchain_rmq_finish(bw, min_cnt, min_sc, n, a, f, p, v, t, n_u_, u_);
}
