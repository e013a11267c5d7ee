/*
 * ORACLE restatement of linear chaining: comput_sc (lchain.c:114-139), mg_lchain_dp (:149-219),
 * mg_chain_bk_end/mg_chain_backtrack (:9-77) and compact_a (:79-112), for the long-read case the hot
 * path runs (is_cdna = 0, n_seg = 1, every anchor on query segment 0).
 *
 * Float semantics: chn_pen_* are float, mg_log2 is the bit-trick of mgpriv.h:63-71, products and
 * sums are evaluated in float WITHOUT fused multiply-add (reference is built -msse4, no FMA), and
 * the penalty is truncated toward zero by (int).  Compile with -ffp-contract=off.
 */
#include <stdlib.h>
#include <string.h>
#include "mgo.h"

static inline float fast_log2(float x) /* mgpriv.h:63-71; valid for x >= 2 */
{
	union { float f; uint32_t i; } z;
	float r;
	z.f = x;
	r = (float)((int32_t)(z.i >> 23 & 255) - 128);
	z.i &= ~(255U << 23);
	z.i += 127U << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

#define SC_NONE INT32_MIN

/* score of extending the chain ending at anchor j by anchor i (lchain.c:114-139) */
static int32_t pair_score(const mgo128_t *ai, const mgo128_t *aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, float pen_gap, float pen_skip)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, span, sc;
	if (dq <= 0 || dq > max_dist_x) return SC_NONE;
	dr = (int32_t)(ai->x - aj->x);
	if (dr == 0 || dq > max_dist_y) return SC_NONE;
	dd = dr > dq ? dr - dq : dq - dr;
	if (dd > bw) return SC_NONE;
	dg = dr < dq ? dr : dq;
	span = (int32_t)(aj->y >> 32 & 0xff);
	sc = span < dg ? span : dg;
	if (dd || dg > span) {
		float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		float lg = dd >= 1 ? fast_log2((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	return sc;
}

/* walk back from chain end z_k; stop at a used anchor or when the score drops by > max_drop from the
 * best seen; return the anchor where the chain is cut (-1: runs to the start).  lchain.c:9-25 */
static int64_t bk_end(int32_t max_drop, int32_t end_sc, int64_t end_i, const int32_t *f, const int64_t *p, int32_t *t)
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

int32_t mgo_lchain_dp(int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter,
					  int32_t min_cnt, int32_t min_sc, float pen_gap, float pen_skip,
					  int64_t n, mgo128_t *a, uint64_t *u, int64_t *n_a_out)
{
	int32_t *f, *t, *v, n_u = 0, max_drop = bw;
	int64_t *p, i, j, k, st = 0, best_in_range = -1, n_v = 0, n_z = 0;
	mgo128_t *z, *b, *w;

	*n_a_out = 0;
	if (n <= 0) return 0;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw) max_dist_y = bw;
	p = (int64_t*)malloc(n * 8);
	f = (int32_t*)malloc(n * 4);
	v = (int32_t*)malloc(n * 4);
	t = (int32_t*)calloc(n, 4);

	/* DP (lchain.c:168-207) */
	for (i = 0; i < n; ++i) {
		int64_t max_j = -1, end_j;
		int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist_x)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = pair_score(&a[i], &a[j], max_dist_x, max_dist_y, bw, pen_gap, pen_skip);
			if (sc == SC_NONE) continue;
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		end_j = j;
		if (best_in_range < 0 || a[i].x - a[best_in_range].x > (uint64_t)(int64_t)max_dist_x) { /* lchain.c:191-196 */
			int32_t mx = INT32_MIN;
			best_in_range = -1;
			for (j = i - 1; j >= st; --j)
				if (mx < f[j]) mx = f[j], best_in_range = j;
		}
		if (best_in_range >= 0 && best_in_range < end_j) { /* lchain.c:197-201 */
			int32_t sc = pair_score(&a[i], &a[best_in_range], max_dist_x, max_dist_y, bw, pen_gap, pen_skip);
			if (sc != SC_NONE && max_f < sc + f[best_in_range])
				max_f = sc + f[best_in_range], max_j = best_in_range;
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
		if (best_in_range < 0 || (a[i].x - a[best_in_range].x <= (uint64_t)(int64_t)max_dist_x && f[best_in_range] < f[i]))
			best_in_range = i;
	}

	/* backtrack (lchain.c:27-77): chain ends by ascending score through the klib sort, visited best first */
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) { free(p); free(f); free(v); free(t); return 0; }
	z = (mgo128_t*)malloc(n_z * sizeof(mgo128_t));
	for (i = 0, k = 0; i < n; ++i) if (f[i] >= min_sc) z[k].x = (uint64_t)(int64_t)f[i], z[k++].y = (uint64_t)i;
	mgo_sort128x(z, n_z);
	memset(t, 0, n * 4);
	for (k = n_z - 1; k >= 0; --k) {
		int64_t e = (int64_t)z[k].y, n_v0 = n_v, cut;
		int32_t sc;
		if (t[e] != 0) continue;
		cut = bk_end(max_drop, (int32_t)z[k].x, e, f, p, t);
		for (i = e; i != cut; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
		sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
		else n_v = n_v0;
	}
	free(z);
	if (n_u == 0) { free(p); free(f); free(v); free(t); return 0; }

	/* compact_a (lchain.c:79-112): anchors of each chain in increasing order, chains by target position */
	b = (mgo128_t*)malloc(n_v * sizeof(mgo128_t));
	for (i = 0, k = 0; i < n_u; ++i) {
		int64_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	w = (mgo128_t*)malloc(n_u * sizeof(mgo128_t));
	for (i = 0, k = 0; i < n_u; ++i) {
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	mgo_sort128x(w, n_u);
	{
		uint64_t *u2 = (uint64_t*)malloc(n_u * 8);
		for (i = 0, k = 0; i < n_u; ++i) {
			int32_t jj = (int32_t)w[i].y, cnt = (int32_t)u[jj];
			u2[i] = u[jj];
			memcpy(&a[k], &b[w[i].y >> 32], cnt * sizeof(mgo128_t));
			k += cnt;
		}
		memcpy(u, u2, n_u * 8);
		free(u2);
	}
	*n_a_out = n_v;
	free(b); free(w); free(p); free(f); free(v); free(t);
	return n_u;
}
