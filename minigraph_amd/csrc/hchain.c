/*
 * hchain.c -- host-side pieces of linear chaining that run after (or instead of) the GPU DP:
 * backtrack + compaction for the RMQ chainer, lchain records, bad-end / bad-seed clean-up and the
 * anchor rewrite.  Reference: lchain.c:9-112,374-441 and map-algo.c:194-330,423-447.
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

mg_lchain_t *mga_lchain_gen(uint32_t hash, int qlen, int n_u, const uint64_t *u, const mg128_t *a)
{
	mg128_t *z;
	mg_lchain_t *r;
	int i, k;
	(void)hash; (void)qlen;
	if (n_u == 0) return 0;
	r = MGA_CALLOC(mg_lchain_t, n_u);
	z = MGA_MALLOC(mg128_t, n_u);
	for (i = k = 0; i < n_u; ++i) { /* order by query start, then score: the klib sort on qs<<32|score */
		int32_t qs = (int32_t)a[k].y + 1 - (int32_t)(a[k].y >> 32 & 0xff);
		z[i].x = (uint64_t)qs << 32 | u[i] >> 32;
		z[i].y = (uint64_t)k << 32 | (uint64_t)(uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	mga_ksort_128x(n_u, z);
	for (i = 0; i < n_u; ++i) {
		mg_lchain_t *ri = &r[i];
		int32_t kk = (int32_t)(z[i].y >> 32), q_span = (int32_t)(a[kk].y >> 32 & 0xff);
		ri->off = kk;
		ri->cnt = (int32_t)z[i].y;
		ri->score = (int32_t)(uint32_t)z[i].x;
		ri->v = (uint32_t)(a[kk].x >> 32);
		ri->rs = (int32_t)a[kk].x + 1 > q_span ? (int32_t)a[kk].x + 1 - q_span : 0;
		ri->qs = (int32_t)(z[i].x >> 32);
		ri->re = (int32_t)a[kk + ri->cnt - 1].x + 1;
		ri->qe = (int32_t)a[kk + ri->cnt - 1].y + 1;
	}
	free(z);
	return r;
}

/* ---- noisy chain ends and seeds inside long indels (map-algo.c:194-330) ---- */

static void trim_high_occ_ends(const mg128_t *a, int32_t max_occ, int32_t max_trim, int32_t *as, int32_t *cnt) /* mm_fix_bad_ends */
{
	int32_t i, k, as0 = *as, cnt0 = *cnt;
	for (i = as0 + cnt0 - 1, k = 0; k < max_trim && k < cnt0; ++k, --i)
		if ((int32_t)(a[i].y >> MG_SEED_OCC_SHIFT) <= max_occ) break;
	*cnt -= k;
	for (i = as0, k = 0; k < *cnt && k < max_trim; ++i, ++k)
		if ((int32_t)(a[i].y >> MG_SEED_OCC_SHIFT) <= max_occ) break;
	*as += k, *cnt -= k;
}

static void trim_gappy_ends(const mg128_t *a, int32_t score, int bw, int min_match, int32_t *as, int32_t *cnt) /* mm_fix_bad_ends_alt */
{
	int32_t i, l, m, as0 = *as, cnt0 = *cnt;
	if (cnt0 < 3) return;
	m = l = (int32_t)(a[as0].y >> 32 & 0xff);
	for (i = as0 + 1; i < as0 + cnt0 - 1; ++i) {
		int32_t lq, lr, mn, mx, q_span = (int32_t)(a[i].y >> 32 & 0xff);
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *as = i;
		l += mn;
		m += mn < q_span ? mn : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= score >> 1) break;
	}
	*cnt = as0 + cnt0 - *as;
	m = l = (int32_t)(a[as0 + cnt0 - 1].y >> 32 & 0xff);
	for (i = as0 + cnt0 - 2; i > *as; --i) {
		int32_t lq, lr, mn, mx, q_span = (int32_t)(a[i+1].y >> 32 & 0xff);
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x;
		lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
		if (mx - mn > l >> 1) *cnt = i + 1 - *as;
		l += mn;
		m += mn < q_span ? mn : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= score >> 1) break;
	}
}

#define GAP_AT(a, as1, i) (((int32_t)(a)[(as1) + (i)].y - (int32_t)(a)[(as1) + (i) - 1].y) - ((int32_t)(a)[(as1) + (i)].x - (int32_t)(a)[(as1) + (i) - 1].x))

static int *long_gaps(int as1, int cnt1, const mg128_t *a, int min_gap, int *n_) /* collect_long_gaps */
{
	int i, n, *K;
	*n_ = 0;
	for (i = 1, n = 0; i < cnt1; ++i) { int gap = GAP_AT(a, as1, i); if (gap < -min_gap || gap > min_gap) ++n; }
	if (n <= 1) return 0;
	K = MGA_MALLOC(int, n);
	for (i = 1, n = 0; i < cnt1; ++i) { int gap = GAP_AT(a, as1, i); if (gap < -min_gap || gap > min_gap) K[n++] = i; }
	*n_ = n;
	return K;
}

static void ignore_seeds_between_opposite_gaps(int as1, int cnt1, mg128_t *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) /* mm_filter_bad_seeds */
{
	int max_st, max_en, n, i, k, max, *K;
	K = long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= MG_SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		qs = (int32_t)a[as1 + i - 1].y;
		rs = (int32_t)a[as1 + i - 1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
	free(K);
}

static void ignore_seeds_in_gap_runs(int as1, int cnt1, mg128_t *a, int min_gap, int max_ext) /* mm_filter_bad_seeds_alt */
{
	int n, k, *K;
	K = long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = GAP_AT(a, as1, i);
		int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
			q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
			rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
			qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1 + j].x, qe1 = (int32_t)a[as1 + j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int j, end = K[l - 1];
			for (j = K[k]; j < end; ++j) a[as1 + j].y |= MG_SEED_IGNORE;
			a[as1 + end].y |= MG_SEED_FIXED;
		}
		k = l;
	}
	free(K);
}

int32_t mga_lchain_cleanup(const mg_mapopt_t *opt, int32_t n_lc, mg_lchain_t *lc, mg128_t *a) /* map-algo.c:424-445 */
{
	int32_t i, n_new = 0;
	for (i = 0; i < n_lc; ++i) {
		mg_lchain_t *p = &lc[i];
		int32_t cnt = p->cnt, off = p->off;
		trim_high_occ_ends(a, opt->lc_max_occ, opt->lc_max_trim, &off, &cnt);
		trim_gappy_ends(a, p->score, opt->bw, 100, &off, &cnt);
		ignore_seeds_between_opposite_gaps(off, cnt, a, 10, 40, opt->max_gap >> 1, 10);
		ignore_seeds_in_gap_runs(off, cnt, a, 30, opt->max_gap >> 1);
		p->off = off, p->cnt = cnt;
		if (cnt >= opt->min_lc_cnt) {
			int32_t q_span = (int32_t)(a[p->off].y >> 32 & 0xff);
			p->rs = (int32_t)a[p->off].x + 1 - q_span;
			p->qs = (int32_t)a[p->off].y + 1 - q_span;
			p->re = (int32_t)a[p->off + p->cnt - 1].x + 1;
			p->qe = (int32_t)a[p->off + p->cnt - 1].y + 1;
			lc[n_new++] = *p;
		}
	}
	return n_new;
}

void mga_update_anchors(int32_t n_a, mg128_t *a, int32_t n, const int32_t *mini_pos) /* lchain.c:410-441 */
{
	int32_t st = -1, j, k, x, L = 0, R = n - 1;
	if (n_a <= 0) return;
	x = (int32_t)a[0].y;
	while (L <= R) {
		int32_t m = (int32_t)(((uint64_t)L + R) >> 1), y = mini_pos[m];
		if (y < x) L = m + 1; else if (y > x) R = m - 1; else { st = m; break; }
	}
	assert(st >= 0);
	for (k = 0, j = st; j < n && k < n_a; ++j)
		if ((int32_t)a[k].y == mini_pos[j])
			a[k].x = (uint64_t)j << 32 | (a[k].x & 0xffffffffU), ++k;
	assert(k == n_a);
}
