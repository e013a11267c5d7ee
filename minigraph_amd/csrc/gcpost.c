/*
 * gcpost.c -- graph-chain post-processing on the host: ordering by score, primary/secondary
 * assignment, filtering and MAPQ (reference gcmisc.c:6-223).  A handful of chains per read.
 */
#include <math.h>
#include <assert.h>
#include "hchain.h"

/* make lc[] and a[] follow the order of gc[] (mg_gchain_restore_order, gcmisc.c:6-36) */
static void restore_order(mg_gchains_t *gcs)
{
	int32_t i, n_a = 0, n_lc = 0;
	mg_llchain_t *lc = MGA_MALLOC(mg_llchain_t, gcs->n_lc > 0 ? gcs->n_lc : 1);
	mg128_t *a = MGA_MALLOC(mg128_t, gcs->n_a > 0 ? gcs->n_a : 1);
	for (i = 0; i < gcs->n_gc; ++i) {
		mg_gchain_t *gc = &gcs->gc[i];
		assert(gc->cnt > 0);
		memcpy(&lc[n_lc], &gcs->lc[gc->off], (size_t)gc->cnt * sizeof(mg_llchain_t));
		memcpy(&a[n_a], &gcs->a[gcs->lc[gc->off].off], (size_t)gc->n_anchor * sizeof(mg128_t));
		n_lc += gc->cnt, n_a += gc->n_anchor;
	}
	memcpy(gcs->lc, lc, (size_t)gcs->n_lc * sizeof(mg_llchain_t));
	memcpy(gcs->a, a, (size_t)gcs->n_a * sizeof(mg128_t));
	free(lc); free(a);
	for (i = 0, n_lc = 0; i < gcs->n_gc; ++i) { gcs->gc[i].off = n_lc; n_lc += gcs->gc[i].cnt; }
	for (i = 0, n_a = 0; i < gcs->n_lc; ++i) { gcs->lc[i].off = n_a; n_a += gcs->lc[i].cnt; }
}

static void restore_offset(mg_gchains_t *gcs) /* gcmisc.c:38-54 */
{
	int32_t i, j, n_a = 0, n_lc = 0;
	for (i = 0; i < gcs->n_gc; ++i) {
		mg_gchain_t *gc = &gcs->gc[i];
		gc->off = n_lc;
		for (j = 0, gc->n_anchor = 0; j < gc->cnt; ++j) {
			mg_llchain_t *lc = &gcs->lc[n_lc + j];
			lc->off = n_a, n_a += lc->cnt, gc->n_anchor += lc->cnt;
		}
		n_lc += gc->cnt;
	}
	assert(n_lc == gcs->n_lc && n_a == gcs->n_a);
}

void mga_gchain_sort_by_score(mg_gchains_t *gcs) /* gcmisc.c:56-71: descending (score, hash) through the klib sort */
{
	mg128_t *z = MGA_MALLOC(mg128_t, gcs->n_gc > 0 ? gcs->n_gc : 1);
	mg_gchain_t *gc = MGA_MALLOC(mg_gchain_t, gcs->n_gc > 0 ? gcs->n_gc : 1);
	int32_t i;
	for (i = 0; i < gcs->n_gc; ++i) z[i].x = (uint64_t)gcs->gc[i].score << 32 | gcs->gc[i].hash, z[i].y = (uint64_t)i;
	mga_ksort_128x(gcs->n_gc, z);
	for (i = gcs->n_gc - 1; i >= 0; --i) gc[gcs->n_gc - 1 - i] = gcs->gc[z[i].y];
	memcpy(gcs->gc, gc, (size_t)gcs->n_gc * sizeof(mg_gchain_t));
	free(z); free(gc);
	restore_order(gcs);
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

void mga_gchain_set_parent(float mask_level, int n, mg_gchain_t *r, int sub_diff, int hard_mask_level) /* gcmisc.c:73-128 */
{
	int i, j, k, *w;
	uint64_t *cov;
	(void)sub_diff;
	if (n <= 0) return;
	for (i = 0; i < n; ++i) r[i].id = i;
	cov = MGA_MALLOC(uint64_t, n);
	w = MGA_MALLOC(int, n);
	w[0] = 0, r[0].parent = 0;
	for (i = 1, k = 1; i < n; ++i) {
		mg_gchain_t *ri = &r[i];
		int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
		if (!hard_mask_level) {
			for (j = 0; j < k; ++j) { /* overlaps with the primaries found so far */
				mg_gchain_t *rp = &r[w[j]];
				int sj = rp->qs, ej = rp->qe;
				if (ej <= si || sj >= ei) continue;
				if (sj < si) sj = si;
				if (ej > ei) ej = ei;
				cov[n_cov++] = (uint64_t)sj << 32 | (uint64_t)ej;
			}
			if (n_cov == 0) { j = k; goto set_parent_test; } /* no overlapping primary: i is a new primary */
			else {
				int jj, x = si;
				qsort(cov, (size_t)n_cov, 8, cmp_u64);
				for (jj = 0; jj < n_cov; ++jj) {
					if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
					x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
				}
				if (ei > x) uncov_len += ei - x;
			}
		}
		for (j = 0; j < k; ++j) {
			mg_gchain_t *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe, min, max, ol;
			if (ej <= si || sj >= ei) continue;
			min = ej - sj < ei - si ? ej - sj : ei - si;
			max = ej - sj > ei - si ? ej - sj : ei - si;
			ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
			if ((float)ol / min - (float)uncov_len / max > mask_level) {
				ri->parent = rp->parent;
				rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
				if (ri->cnt >= rp->cnt) ++rp->n_sub;
				break;
			}
		}
set_parent_test:
		if (j == k) w[k++] = i, ri->parent = i, ri->n_sub = 0;
	}
	free(cov); free(w);
}

int mga_gchain_flt_sub(float pri_ratio, int min_diff, int best_n, int n, mg_gchain_t *r) /* gcmisc.c:130-148 */
{
	if (pri_ratio > 0.0f && n > 0) {
		int i, k, n_2nd = 0;
		for (i = k = 0; i < n; ++i) {
			int p = r[i].parent;
			if (p == i) r[i].flt = 0, ++k;
			else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
				if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].ps == r[p].ps && r[i].pe == r[p].pe)) r[i].flt = 0, ++n_2nd, ++k;
				else r[i].flt = 1;
			} else r[i].flt = 1;
		}
		return k;
	}
	return n;
}

void mga_gchain_drop_flt(mg_gchains_t *gcs) /* gcmisc.c:150-188 */
{
	int32_t i, n_gc, n_lc, n_a, n_lc0, n_a0, *o2n;
	if (gcs->n_gc == 0) return;
	o2n = MGA_MALLOC(int32_t, gcs->n_gc);
	for (i = 0, n_gc = 0; i < gcs->n_gc; ++i) {
		mg_gchain_t *r = &gcs->gc[i];
		o2n[i] = -1;
		if (r->flt || r->cnt == 0) { free(r->p); r->p = 0; continue; }
		o2n[i] = n_gc++;
	}
	n_gc = n_lc = n_a = 0, n_lc0 = n_a0 = 0;
	for (i = 0; i < gcs->n_gc; ++i) {
		mg_gchain_t *r = &gcs->gc[i];
		if (o2n[i] >= 0) {
			memmove(&gcs->a[n_a], &gcs->a[n_a0], (size_t)r->n_anchor * sizeof(mg128_t));
			memmove(&gcs->lc[n_lc], &gcs->lc[n_lc0], (size_t)r->cnt * sizeof(mg_llchain_t));
			gcs->gc[n_gc] = *r;
			gcs->gc[n_gc].id = n_gc;
			gcs->gc[n_gc].parent = o2n[gcs->gc[n_gc].parent];
			++n_gc, n_lc += r->cnt, n_a += r->n_anchor;
		}
		n_lc0 += r->cnt, n_a0 += r->n_anchor;
	}
	assert(n_lc0 == gcs->n_lc && n_a0 == gcs->n_a);
	free(o2n);
	gcs->n_gc = n_gc, gcs->n_lc = n_lc, gcs->n_a = n_a;
	if (n_a != n_a0) {
		gcs->a = MGA_REALLOC(mg128_t, gcs->a, gcs->n_a > 0 ? gcs->n_a : 1);
		gcs->lc = MGA_REALLOC(mg_llchain_t, gcs->lc, gcs->n_lc > 0 ? gcs->n_lc : 1);
		gcs->gc = MGA_REALLOC(mg_gchain_t, gcs->gc, gcs->n_gc > 0 ? gcs->n_gc : 1);
	}
	restore_offset(gcs);
}

void mga_gchain_set_mapq(mg_gchains_t *gcs, int qlen, int max_mini, int min_gc_score) /* gcmisc.c:190-223 */
{
	static const float q_coef = 40.0f;
	int64_t sum_sc = 0;
	float uniq_ratio, r_sc, r_cnt;
	int i, t_sc, t_cnt;
	if (gcs == 0 || gcs->n_gc == 0) return;
	t_sc = qlen < 100 ? qlen : 100;
	t_cnt = max_mini < 10 ? max_mini : 10;
	if (t_cnt < 5) t_cnt = 5;
	r_sc = 1.0 / t_sc;
	r_cnt = 1.0 / t_cnt;
	for (i = 0; i < gcs->n_gc; ++i)
		if (gcs->gc[i].parent == gcs->gc[i].id) sum_sc += gcs->gc[i].score;
	uniq_ratio = (float)sum_sc / (sum_sc + gcs->rep_len);
	for (i = 0; i < gcs->n_gc; ++i) {
		mg_gchain_t *r = &gcs->gc[i];
		if (r->parent == r->id) {
			int mapq, subsc;
			float pen_s1 = (r->score > t_sc ? 1.0f : r->score * r_sc) * uniq_ratio;
			float x, pen_cm = r->n_anchor > t_cnt ? 1.0f : r->n_anchor * r_cnt;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			subsc = r->subsc > min_gc_score ? r->subsc : min_gc_score;
			x = (float)subsc / r->score;
			mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
			mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			mapq = mapq > 0 ? mapq : 0;
			if (r->score > subsc && mapq == 0) mapq = 1;
			r->mapq = mapq < 60 ? mapq : 60;
		} else r->mapq = 0;
	}
}
