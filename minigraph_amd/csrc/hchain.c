/*
 * hchain.c -- backtrack + compaction behind the host RMQ chainer (rmq.c; -x asm and the tied long-join rescues).
 * Reference: lchain.c:9-112.  Everything after the linear chains -- chain records, clean-up, graph chaining -- lives in gc_core.h.
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
