/*
 * hchain.c -- backtrack + compaction behind the host RMQ chainer (rmq.c; -x asm and the tied long-join rescues), and the first chaining pass of ULTRA-LONG -x lr reads.
 * Reference: lchain.c:9-219.  Everything after the linear chains -- chain records, clean-up, graph chaining -- lives in gc_core.h.
 */
#include <assert.h>
#include "hchain.h"

/* ---- mg_chain_bk_end + mg_chain_backtrack (lchain.c:9-77), single pass instead of count-then-fill ---- */
static int64_t chain_cut(int32_t max_drop, int32_t end_sc, int64_t end_i, const int32_t *f, const int64_t *p, int32_t *t)
{
	int64_t i = end_i, stop = -1, best_i = i;
	int32_t best = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		stop = i = p[i];
		s = i < 0 ? end_sc : end_sc - f[i];
		if (s > best) best = s, best_i = i;
		else if (best - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = end_i; i >= 0 && i != stop; i = p[i]) t[i] = 0;
	return best_i;
}

uint64_t *mga_chain_backtrack(int64_t n, const int32_t *f, const int64_t *p, int32_t *v, int32_t *t, int32_t min_cnt, int32_t min_sc,
							  int32_t max_drop, int32_t extra_u, int32_t *n_u_, int32_t *n_v_)
{
	mg128_t *z;
	uint64_t *u;
	int64_t i, k, n_z = 0, n_v = 0, m_u = 16;
	int32_t n_u = 0;
	*n_u_ = *n_v_ = 0;
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) return 0;
	z = MGA_MALLOC(mg128_t, n_z);
	for (i = 0, k = 0; i < n; ++i) if (f[i] >= min_sc) z[k].x = (uint64_t)(int64_t)f[i], z[k++].y = (uint64_t)i;
	mga_ksort_128x(n_z, z);
	u = MGA_MALLOC(uint64_t, m_u + extra_u);
	memset(t, 0, (size_t)n * 4);
	for (k = n_z - 1; k >= 0; --k) {
		int64_t e = (int64_t)z[k].y, n_v0 = n_v, cut;
		int32_t sc;
		if (t[e] != 0) continue;
		cut = chain_cut(max_drop, (int32_t)z[k].x, e, f, p, t);
		for (i = e; i != cut; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
		sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) {
			if (n_u == m_u) { m_u += m_u >> 1; u = MGA_REALLOC(uint64_t, u, m_u + extra_u); }
			u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
		} else n_v = n_v0;
	}
	free(z);
	*n_u_ = n_u, *n_v_ = (int32_t)n_v;
	return u;
}

mg128_t *mga_compact_a(int32_t n_u, uint64_t *u, int32_t n_v, const int32_t *v, const mg128_t *a)
{
	mg128_t *b = MGA_MALLOC(mg128_t, n_v > 0 ? n_v : 1), *w = MGA_MALLOC(mg128_t, n_u), *out = MGA_MALLOC(mg128_t, n_v > 0 ? n_v : 1);
	uint64_t *u2 = MGA_MALLOC(uint64_t, n_u);
	int64_t i, j, k;
	for (i = 0, k = 0; i < n_u; ++i) { /* each chain's anchors in increasing order */
		int32_t k0 = (int32_t)k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	for (i = k = 0; i < n_u; ++i) { /* chains by target position of their first anchor */
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	mga_ksort_128x(n_u, w);
	for (i = k = 0; i < n_u; ++i) {
		int32_t j2 = (int32_t)w[i].y, cnt = (int32_t)u[j2];
		u2[i] = u[j2];
		memcpy(&out[k], &b[w[i].y >> 32], (size_t)cnt * sizeof(mg128_t));
		k += cnt;
	}
	memcpy(u, u2, (size_t)n_u * 8);
	free(b); free(w); free(u2);
	return out;
}

/* ---- first chaining pass of an ultra-long read on a host thread (mg_lchain_dp's forward loop, lchain.c:168-207; single segment, not cDNA) ----
 * k_lchain gives a read one wavefront: ~10 k cycles per anchor are hidden by the other reads' wavefronts when a chunk holds thousands of 10 kb reads, but a 5 Mbp read
 * is 3 x 10^5 dependent anchor steps on ONE wavefront ([measured] 1.4 s, round 1-3: slower than the whole reference job).  A host core takes the same steps from its
 * caches in ~0.1 s, and reads of that length are few, so they are placed here -- like the RMQ chainer of -x asm -- while sketch, seeds, WFA and text stay on the device
 * (the long-query path of mapper.c).  Fills f, p, v (t: zeroed by the caller); the caller backtracks with mga_chain_backtrack / mga_compact_a. */
typedef struct { int32_t dist_x, dist_y, bw; float pen_gap, pen_skip; } lcdp_par_t;

static inline int lcdp_pair(const mg128_t *ai, const mg128_t *aj, const lcdp_par_t *P, int32_t *sc_) /* comput_sc (lchain.c:114-139) for one segment: 0 when aj cannot precede ai */
{
	const int32_t dq = (int32_t)ai->y - (int32_t)aj->y;
	int32_t dr, dd, dg, span, sc;
	if (dq <= 0 || dq > P->dist_x) return 0;
	dr = (int32_t)(ai->x - aj->x);
	if (dr == 0 || dq > P->dist_y) return 0;
	dd = dr > dq ? dr - dq : dq - dr;
	if (dd > P->bw) return 0;
	dg = dr < dq ? dr : dq;
	span = (int32_t)(aj->y >> 32 & 0xff);
	sc = span < dg ? span : dg;
	if (dd || dg > span) {
		const float lin = P->pen_gap * (float)dd + P->pen_skip * (float)dg;
		const float lg = dd >= 1 ? mga_log2f((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	*sc_ = sc;
	return 1;
}

void mga_lchain_dp_fwd(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float pen_gap, float pen_skip,
					   int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t)
{
	lcdp_par_t P;
	int64_t i, lo = 0, best_in_reach = -1; /* lo: first anchor that can still precede anchor i; best_in_reach: the reference's max_ii */
	P.dist_x = max_dist_x < bw ? bw : max_dist_x, P.dist_y = max_dist_y < bw ? bw : max_dist_y, P.bw = bw, P.pen_gap = pen_gap, P.pen_skip = pen_skip;
	for (i = 0; i < n; ++i) {
		const mg128_t *ai = &a[i];
		int64_t j, from = -1, scan_end;
		int32_t best = (int32_t)(ai->y >> 32 & 0xff), skipped = 0;
		while (lo < i && (ai->x >> 32 != a[lo].x >> 32 || ai->x > a[lo].x + (uint64_t)(int64_t)P.dist_x)) ++lo;
		if (i - lo > max_iter) lo = i - max_iter;
		for (j = i - 1; j >= lo; --j) { /* nearest first; the order decides which predecessors the skip heuristic still gets to see */
			int32_t sc;
			if (!lcdp_pair(ai, &a[j], &P, &sc)) continue;
			sc += f[j];
			if (sc > best) { best = sc, from = j; if (skipped > 0) --skipped; }
			else if (t[j] == (int32_t)i && ++skipped > max_skip) break;
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		scan_end = j;
		if (best_in_reach < 0 || ai->x - a[best_in_reach].x > (uint64_t)(int64_t)P.dist_x) { /* the best-scoring anchor in reach fell out of it: look again */
			int32_t top = INT32_MIN;
			best_in_reach = -1;
			for (j = i - 1; j >= lo; --j) if (top < f[j]) top = f[j], best_in_reach = j;
		}
		if (best_in_reach >= 0 && best_in_reach < scan_end) { /* the scan was cut before it got there */
			int32_t sc;
			if (lcdp_pair(ai, &a[best_in_reach], &P, &sc) && best < sc + f[best_in_reach]) best = sc + f[best_in_reach], from = best_in_reach;
		}
		f[i] = best, p[i] = from;
		v[i] = from >= 0 && v[from] > best ? v[from] : best;
		if (best_in_reach < 0 || (ai->x - a[best_in_reach].x <= (uint64_t)(int64_t)P.dist_x && f[best_in_reach] < best)) best_in_reach = i;
	}
}
