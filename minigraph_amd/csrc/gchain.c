/*
 * gchain.c -- graph chaining: DP over linear chains with graph reachability (mg_gchain1_dp,
 * reference gchain1.c:16-240) and assembly of the final graph chains incl. bridging of anchor-free
 * vertices (mg_gchain_gen and helpers, gchain1.c:242-520).  Host code: ~2 linear chains per read.
 */
#include <math.h>
#include <stdio.h>
#include <assert.h>
#include "hchain.h"

typedef struct { uint32_t srt; int32_t i; } gc_frag_t;

static int32_t frag_find_max(int32_t n, const gc_frag_t *gf, uint32_t x) /* find_max, gchain1.c:16-30: last index with srt < x */
{
	int32_t s = 0, e = n;
	if (n == 0) return -1;
	if (gf[n-1].srt < x) return n - 1;
	if (gf[0].srt >= x) return -1;
	while (e > s) {
		int32_t m = s + (e - s) / 2;
		if (gf[m].srt >= x) e = m; else s = m + 1;
	}
	return s;
}

static inline int32_t target_dist(const gfa_t *g, const mg_lchain_t *l0, const mg_lchain_t *l1) /* gchain1.c:32-36 */
{
	return (l1->qs - l0->qe) - (g->seg[l0->v>>1].len - l0->re) + (g->seg[l1->v>>1].len - l1->rs);
}

static inline int32_t link_score(const mga_path_dst_t *dj, const mg_lchain_t *li, const mg_lchain_t *lc, const mg128_t *an, const gc_frag_t *a, const int32_t *f,
								 int bw, int ref_bonus, float chn_pen_gap) /* cal_sc, gchain1.c:38-60 */
{
	const mg_lchain_t *lj;
	int32_t gap, sc, segi, segj;
	float lin_pen, log_pen;
	if (dj->n_path == 0) return INT32_MIN;
	segi = (int32_t)((an[li->off].y & MG_SEED_SEG_MASK) >> MG_SEED_SEG_SHIFT);
	gap = dj->dist - dj->target_dist;
	lj = &lc[a[dj->meta].i];
	segj = (int32_t)((an[lj->off + lj->cnt - 1].y & MG_SEED_SEG_MASK) >> MG_SEED_SEG_SHIFT);
	if (gap < 0) gap = -gap;
	if (segi == segj && gap > bw) return INT32_MIN;
	if (lj->qe <= li->qs) sc = li->score;
	else sc = (int32_t)((double)(li->qe - lj->qe) / (li->qe - li->qs) * li->score + .499);
	if (dj->is_0) sc += ref_bonus;
	lin_pen = chn_pen_gap * (float)gap;
	log_pen = gap >= 2 ? mga_log2f((float)gap) : 0.0f;
	sc -= (int32_t)(lin_pen + log_pen);
	sc += f[dj->meta];
	return sc;
}

int32_t mga_gchain1_dp(const gfa_t *g, int32_t *n_lc_, mg_lchain_t *lc, int32_t qlen, int32_t max_dist_g, int32_t max_dist_q, int32_t bw, int32_t max_skip,
					   int32_t ref_bonus, float chn_pen_gap, float chn_pen_skip, float mask_level, const mg128_t *an, uint64_t **u_)
{
	int32_t i, j, k, m_dst = 0, n_dst = 0, n_ext, n_u, n_v, n_lc = *n_lc_;
	int32_t *f, *v, *t;
	int64_t *p;
	uint64_t *u;
	mga_path_dst_t *dst = 0;
	gc_frag_t *a;
	mg_lchain_t *swap;
	(void)chn_pen_skip;

	*u_ = 0;
	if (n_lc == 0) return 0;
	a = MGA_MALLOC(gc_frag_t, n_lc);
	for (i = n_ext = 0; i < n_lc; ++i) { /* gchain1.c:78-90: chains far from both segment ends cannot be linked */
		mg_lchain_t *r = &lc[i];
		int32_t isolated = 0, min_end = g->seg[r->v>>1].len - r->re;
		r->dist_pre = -1;
		if (r->rs < min_end) min_end = r->rs;
		if (min_end > max_dist_g) isolated = 1;
		else if (min_end >> 3 > r->score) isolated = 1;
		a[i].srt = (uint32_t)isolated << 31 | (uint32_t)r->qe;
		a[i].i = i;
		if (!isolated) ++n_ext;
	}
	if (n_ext < 2) { /* gchain1.c:91-98 */
		free(a);
		u = MGA_MALLOC(uint64_t, n_lc);
		for (i = 0; i < n_lc; ++i) u[i] = (uint64_t)lc[i].score << 32 | 1;
		*u_ = u;
		return n_lc;
	}
	{ /* radix_sort_gc: 4-byte key, exact klib permutation (gchain1.c:13-14,99) */
		uint64_t *key = MGA_MALLOC(uint64_t, n_lc);
		int64_t *perm = MGA_MALLOC(int64_t, n_lc);
		gc_frag_t *tmp = MGA_MALLOC(gc_frag_t, n_lc);
		for (i = 0; i < n_lc; ++i) key[i] = a[i].srt;
		mga_ksort_perm(n_lc, key, 4, perm);
		for (i = 0; i < n_lc; ++i) tmp[i] = a[perm[i]];
		memcpy(a, tmp, (size_t)n_lc * sizeof(gc_frag_t));
		free(key); free(perm); free(tmp);
	}
	v = MGA_MALLOC(int32_t, n_lc);
	f = MGA_MALLOC(int32_t, n_ext);
	p = MGA_MALLOC(int64_t, n_ext);
	t = MGA_CALLOC(int32_t, n_ext);

	for (i = 0; i < n_ext; ++i) { /* gchain1.c:108-208 */
		gc_frag_t *ai = &a[i];
		mg_lchain_t *li = &lc[ai->i];
		int32_t segi = (int32_t)((an[li->off].y & MG_SEED_SEG_MASK) >> MG_SEED_SEG_SHIFT);
		{ /* candidate predecessors: chains ending before li starts on the query, within the bands */
			int32_t x = li->qs + bw, n_skip = 0;
			if (x > qlen) x = qlen;
			x = frag_find_max(i, a, (uint32_t)x);
			n_dst = 0;
			for (j = x; j >= 0; --j) {
				gc_frag_t *aj = &a[j];
				mg_lchain_t *lj = &lc[aj->i];
				mga_path_dst_t *q;
				int32_t tdist, segj, dq;
				if (lj->qs >= li->qs) continue;
				if (lj->qe > li->qs) {
					int o = lj->qe - li->qs;
					if (o > (lj->qe - lj->qs) * mask_level || o > (li->qe - li->qs) * mask_level) continue;
				}
				dq = li->qs - lj->qe;
				segj = (int32_t)((an[lj->off + lj->cnt - 1].y & MG_SEED_SEG_MASK) >> MG_SEED_SEG_SHIFT);
				if (segi == segj) { if (dq > max_dist_q) break; }
				else { if (dq > max_dist_g && dq > max_dist_q) break; }
				if (li->v != lj->v) {
					int32_t min_dist = li->rs + (g->seg[lj->v>>1].len - lj->re);
					if (min_dist > max_dist_g) continue;
					if (segi == segj && min_dist - bw > li->qs - lj->qe) continue;
					tdist = target_dist(g, lj, li);
					if (tdist < 0) continue;
				} else if (lj->rs >= li->rs || lj->re >= li->re) {
					continue;
				} else {
					int32_t dr = li->rs - lj->re, w = dr > dq ? dr - dq : dq - dr;
					if (segi == segj && w > bw) continue;
					if (dr > max_dist_g || dr < -max_dist_g) continue;
					if (lj->re > li->rs) {
						int o = lj->re - li->rs;
						if (o > (lj->re - lj->rs) * mask_level || o > (li->re - li->rs) * mask_level) continue;
					}
					tdist = target_dist(g, lj, li);
				}
				MGA_GROW(mga_path_dst_t, dst, n_dst, m_dst);
				q = &dst[n_dst++];
				memset(q, 0, sizeof *q);
				q->inner = (li->v == lj->v);
				q->v = lj->v ^ 1;
				q->meta = (uint32_t)j;
				q->qlen = li->qs - lj->qe;
				q->target_dist = tdist;
				q->target_hash = 0, q->check_hash = 0;
				if (t[j] == i) { if (++n_skip > max_skip) break; }
				if (p[j] >= 0) t[p[j]] = i;
			}
		}
		{ /* reachability on the graph, then drop unreachable / out-of-band / hopeless candidates */
			int32_t kk;
			mga_pathv_t *unused = mga_shortest_k(g, li->v ^ 1, n_dst, dst, max_dist_g + (g->seg[li->v>>1].len - li->rs), MG_MAX_SHORT_K, 0);
			(void)unused;
			for (j = kk = 0; j < n_dst; ++j) {
				mga_path_dst_t *dj = &dst[j];
				int32_t sc;
				if (dj->n_path == 0) continue;
				sc = link_score(dj, li, lc, an, a, f, bw, ref_bonus, chn_pen_gap);
				if (sc == INT32_MIN) continue;
				if (sc + li->score < 0) continue;
				dst[kk++] = dst[j];
			}
			n_dst = kk;
		}
		{ /* DP */
			int32_t max_f = li->score, max_j = -1, max_d = -1, max_inner = 0;
			uint32_t max_hash = 0;
			for (j = 0; j < n_dst; ++j) {
				mga_path_dst_t *dj = &dst[j];
				int32_t sc = link_score(dj, li, lc, an, a, f, bw, ref_bonus, chn_pen_gap);
				if (sc == INT32_MIN) continue;
				if (sc > max_f) max_f = sc, max_j = (int32_t)dj->meta, max_d = dj->dist, max_hash = dj->hash, max_inner = dj->inner;
			}
			f[i] = max_f, p[i] = max_j;
			li->dist_pre = max_d, li->hash_pre = max_hash, li->inner_pre = max_inner;
			v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
		}
	}
	free(dst);

	u = mga_chain_backtrack(n_ext, f, p, v, t, 0, 0, INT32_MAX, n_lc - n_ext, &n_u, &n_v);
	free(f); free(p); free(t);
	if (u == 0) u = MGA_MALLOC(uint64_t, n_lc - n_ext + 1);
	for (i = 0; i < n_lc - n_ext; ++i) { /* isolated chains become singletons (gchain1.c:221-224) */
		u[n_u++] = (uint64_t)lc[a[n_ext + i].i].score << 32 | 1;
		v[n_v++] = n_ext + i;
	}
	swap = MGA_MALLOC(mg_lchain_t, n_v > 0 ? n_v : 1);
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) swap[k++] = lc[a[v[k0 + (ni - j - 1)]].i];
	}
	assert(k == n_v);
	memcpy(lc, swap, (size_t)n_v * sizeof(mg_lchain_t));
	*n_lc_ = n_v, *u_ = u;
	free(a); free(swap); free(v);
	return n_u;
}

/* blen / mlen / div / path coordinates of every graph chain (mg_gchain_extra, gchain1.c:242-297) */
static void gchain_extra(const gfa_t *g, mg_gchains_t *gs)
{
	int32_t i, j, k;
	for (i = 0; i < gs->n_gc; ++i) {
		mg_gchain_t *p = &gs->gc[i];
		const mg_llchain_t *q;
		const mg128_t *last_a;
		int32_t q_span, rest_pl, tmp, n_mini;
		p->qs = p->qe = p->ps = p->pe = -1, p->plen = p->blen = p->mlen = 0, p->div = -1.0f;
		if (p->cnt == 0) continue;
		q = &gs->lc[p->off];
		q_span = (int32_t)(gs->a[q->off].y >> 32 & 0xff);
		p->qs = (int32_t)gs->a[q->off].y + 1 - q_span;
		p->ps = (int32_t)gs->a[q->off].x + 1 - q_span;
		tmp = (int32_t)(gs->a[q->off].x >> 32);
		q = &gs->lc[p->off + p->cnt - 1];
		p->qe = (int32_t)gs->a[q->off + q->cnt - 1].y + 1;
		p->pe = g->seg[q->v>>1].len - (int32_t)gs->a[q->off + q->cnt - 1].x - 1;
		n_mini = (int32_t)(gs->a[q->off + q->cnt - 1].x >> 32) - tmp + 1;
		rest_pl = 0;
		last_a = &gs->a[gs->lc[p->off].off];
		for (j = 0; j < p->cnt; ++j) {
			const mg_llchain_t *ql = &gs->lc[p->off + j];
			int32_t vlen = g->seg[ql->v>>1].len;
			p->plen += vlen;
			for (k = 0; k < ql->cnt; ++k) {
				const mg128_t *r = &gs->a[ql->off + k];
				int32_t pl, qlv = (int32_t)r->y - (int32_t)last_a->y;
				int32_t span = (int32_t)(r->y >> 32 & 0xff);
				if (j == 0 && k == 0) pl = qlv = span;
				else if (j > 0 && k == 0) pl = (int32_t)r->x + 1 + rest_pl;
				else pl = (int32_t)r->x - (int32_t)last_a->x;
				if (qlv < 0) qlv = -qlv, n_mini += (int32_t)(last_a->x >> 32) - (int32_t)(r->x >> 32);
				p->blen += pl > qlv ? pl : qlv;
				p->mlen += pl > span && qlv > span ? span : pl < qlv ? pl : qlv;
				last_a = r;
			}
			if (ql->cnt == 0) rest_pl += vlen;
			else rest_pl = vlen - (int32_t)gs->a[ql->off + ql->cnt - 1].x - 1;
		}
		p->pe = p->plen - p->pe;
		p->div = n_mini >= p->n_anchor ? (float)(log((double)n_mini / p->n_anchor) / q_span) : (float)(log((double)p->n_anchor / n_mini) / q_span);
	}
}

typedef struct {
	const gfa_t *g;
	const gfa_edseq_t *es;
	const char *qseq;
	int32_t n_seg, n_llc, m_llc, n_a;
	mg_llchain_t *llc;
} bridge_t;

static inline void copy_lchain(mg_llchain_t *q, const mg_lchain_t *p, int32_t *n_a, mg128_t *a_new, const mg128_t *a_old, int32_t ed)
{
	q->cnt = p->cnt, q->v = p->v, q->score = p->score, q->ed = ed;
	memcpy(&a_new[*n_a], &a_old[p->off], (size_t)q->cnt * sizeof(mg128_t));
	q->off = *n_a;
	*n_a += q->cnt;
}

static void push_empty_vertex(bridge_t *b, uint32_t v)
{
	mg_llchain_t *q;
	MGA_GROW(mg_llchain_t, b->llc, b->n_llc, b->m_llc);
	q = &b->llc[b->n_llc++];
	q->off = q->cnt = q->score = 0, q->v = v, q->ed = -1;
}

static int32_t bridge_by_walk(bridge_t *b, const mg_lchain_t *l0, const mg_lchain_t *l1) /* bridge_shortk, gchain1.c:319-347 */
{
	int32_t s, n_pathv;
	mga_path_dst_t dst;
	mga_pathv_t *p;
	memset(&dst, 0, sizeof dst);
	dst.v = l0->v ^ 1;
	assert(l1->dist_pre >= 0);
	dst.target_dist = l1->dist_pre, dst.target_hash = l1->hash_pre, dst.check_hash = 1;
	p = mga_shortest_k(b->g, l1->v ^ 1, 1, &dst, dst.target_dist, MG_MAX_SHORT_K, &n_pathv);
	if (n_pathv == 0 || dst.target_hash != dst.hash) {
		fprintf(stderr, "[W::%s] %c%s[%d] -> %c%s[%d], dist=%d, target_dist=%d; chain skiped.\n", "bridge_shortk", "><"[(l1->v^1)&1], b->g->seg[l1->v>>1].name, l1->v^1,
				"><"[(l0->v^1)&1], b->g->seg[l0->v>>1].name, l0->v^1, dst.dist, dst.target_dist);
		free(p);
		return -1;
	}
	for (s = n_pathv - 2; s >= 1; --s) push_empty_vertex(b, p[s].v ^ 1); /* found backwards: reverse and flip */
	free(p);
	return 0;
}

static int32_t bridge_by_gwfa(bridge_t *b, int32_t kmer_size, int32_t gdp_max_ed, const mg_lchain_t *l0, const mg_lchain_t *l1, int32_t *ed) /* gchain1.c:349-381 */
{
	int32_t qs = l0->qe - kmer_size, qe = l1->qs + kmer_size, end0 = l0->re - kmer_size, end1 = l1->rs + kmer_size - 1, j, nv, *path, s;
	*ed = -1;
	s = mga_gwfa_bridge(b->g, b->es, qe - qs, &b->qseq[qs], l0->v, end0, l1->v, end1, gdp_max_ed / 2, gdp_max_ed, &path, &nv);
	if (s < 0) { free(path); return 0; }
	for (j = 1; j < nv - 1; ++j) push_empty_vertex(b, (uint32_t)path[j]);
	free(path);
	*ed = s;
	return 1;
}

static int32_t bridge_lchains(mg_gchains_t *gc, bridge_t *b, int32_t kmer_size, int32_t gdp_max_ed, const mg_lchain_t *l0, const mg_lchain_t *l1, const mg128_t *a) /* gchain1.c:383-407 */
{
	if (l1->v != l0->v) {
		int32_t ed = -1, ret = 0;
		if (b->n_seg > 1 || !bridge_by_gwfa(b, kmer_size, gdp_max_ed, l0, l1, &ed)) ret = bridge_by_walk(b, l0, l1);
		if (ret < 0) return -1;
		MGA_GROW(mg_llchain_t, b->llc, b->n_llc, b->m_llc);
		copy_lchain(&b->llc[b->n_llc++], l1, &b->n_a, gc->a, a, ed);
	} else {
		int32_t k;
		mg_llchain_t *t = &b->llc[b->n_llc - 1];
		for (k = 0; k < l1->cnt; ++k) {
			const mg128_t *ak = &a[l1->off + k];
			if ((int32_t)ak->x > l0->re && (int32_t)ak->y > l0->qe) break;
		}
		if (k < l1->cnt) {
			t->cnt += l1->cnt - k, t->score += l1->score;
			memcpy(&gc->a[b->n_a], &a[l1->off + k], (size_t)(l1->cnt - k) * sizeof(mg128_t));
			b->n_a += l1->cnt - k;
		}
	}
	return 0;
}

static void resolve_overlap(mg_lchain_t *l0, mg_lchain_t *l1, const mg128_t *a) /* gchain1.c:409-441 */
{
	int32_t j, x, y, shift0, shift1;
	x = (int32_t)a[l1->off].x, y = (int32_t)a[l1->off].y;
	for (j = l0->cnt - 1; j >= 0; --j)
		if ((int32_t)a[l0->off + j].y <= y && (l0->v != l1->v || (int32_t)a[l0->off + j].x <= x)) break;
	shift0 = l0->cnt - 1 - j;
	x = (int32_t)a[l0->off + l0->cnt - 1].x, y = (int32_t)a[l0->off + l0->cnt - 1].y;
	for (j = 0; j < l1->cnt; ++j)
		if ((int32_t)a[l1->off + j].y >= y && (l0->v != l1->v || (int32_t)a[l1->off + j].x >= x)) break;
	shift1 = j;
	assert(shift1 < l1->cnt);
	if (shift0 > 0) {
		l0->cnt -= shift0;
		if (l0->cnt) {
			l0->qe = (int32_t)a[l0->off + l0->cnt - 1].y + 1;
			l0->re = (int32_t)a[l0->off + l0->cnt - 1].x + 1;
		}
	}
	if (shift1 > 0) {
		l1->off += shift1, l1->cnt -= shift1;
		l1->qs = (int32_t)a[l1->off].y + 1 - (int32_t)(a[l1->off].y >> 32 & 0xff);
		l1->rs = (int32_t)a[l1->off].x + 1 - (int32_t)(a[l1->off].y >> 32 & 0xff);
	}
	if (l0->cnt == 0) l0->qs = l0->qe = l1->qs, l0->rs = l0->re = l1->rs;
}

mg_gchains_t *mga_gchain_gen(const gfa_t *g, const gfa_edseq_t *es, int32_t n_u, const uint64_t *u, mg_lchain_t *lc, const mg128_t *a, uint32_t hash,
							 int32_t min_gc_cnt, int32_t min_gc_score, int32_t gdp_max_ed, int32_t n_seg, const char *qseq) /* gchain1.c:443-520 */
{
	mg_gchains_t *gc = MGA_CALLOC(mg_gchains_t, 1);
	int32_t i, j, k, st, kmer_size;
	bridge_t b;
	for (i = 0, st = 0; i < n_u; ++i) {
		int32_t m = 0, nui = (int32_t)u[i];
		for (j = 0; j < nui; ++j) m += lc[st + j].cnt;
		if (m >= min_gc_cnt && (int64_t)(u[i] >> 32) >= min_gc_score) gc->n_gc++, gc->n_a += m;
		st += nui;
	}
	if (gc->n_gc == 0) return gc;
	gc->km = 0;
	gc->gc = MGA_CALLOC(mg_gchain_t, gc->n_gc);
	gc->a = MGA_MALLOC(mg128_t, gc->n_a);
	memset(&b, 0, sizeof b);
	b.g = g, b.es = es, b.n_seg = n_seg, b.qseq = qseq;
	kmer_size = (int32_t)(a[0].y >> 32 & 0xff);
	for (i = k = 0, st = 0, b.n_a = 0; i < n_u; ++i) {
		int32_t n_a0 = b.n_a, n_llc0 = b.n_llc, m = 0, nui = (int32_t)u[i];
		for (j = 0; j < nui; ++j) m += lc[st + j].cnt;
		if (m >= min_gc_cnt && (int64_t)(u[i] >> 32) >= min_gc_score) {
			uint32_t h = hash;
			int32_t j0;
			gc->gc[k].score = (int32_t)(u[i] >> 32);
			gc->gc[k].off = n_llc0;
			for (j = 0; j < nui; ++j) {
				const mg_lchain_t *p = &lc[st + j];
				h += mga_hash_u32((uint32_t)p->qs) + mga_hash_u32((uint32_t)p->re) + mga_hash_u32(p->v);
			}
			gc->gc[k].hash = mga_hash_u32(h);
			for (j = 1; j < nui; ++j) resolve_overlap(&lc[st + j - 1], &lc[st + j], a);
			MGA_GROW(mg_llchain_t, b.llc, b.n_llc, b.m_llc);
			copy_lchain(&b.llc[b.n_llc++], &lc[st], &b.n_a, gc->a, a, -1);
			for (j0 = 0, j = 1; j < nui; ++j) {
				const mg_lchain_t *l0 = &lc[st + j0], *l1 = &lc[st + j];
				if (l1->cnt > 0) {
					int32_t ret = bridge_lchains(gc, &b, kmer_size, gdp_max_ed, l0, l1, a), t;
					if (ret < 0)
						for (t = j0; t < j; ++t) {
							ret = bridge_lchains(gc, &b, kmer_size, gdp_max_ed, &lc[st + t], &lc[st + t + 1], a);
							assert(ret >= 0);
						}
					j0 = j;
				}
			}
			gc->gc[k].cnt = b.n_llc - n_llc0;
			gc->gc[k].n_anchor = b.n_a - n_a0;
			++k;
		}
		st += nui;
	}
	assert(b.n_a <= gc->n_a);
	gc->n_a = b.n_a;
	gc->n_lc = b.n_llc;
	gc->lc = MGA_MALLOC(mg_llchain_t, b.n_llc > 0 ? b.n_llc : 1);
	memcpy(gc->lc, b.llc, (size_t)b.n_llc * sizeof(mg_llchain_t));
	free(b.llc);
	gchain_extra(g, gc);
	mga_gchain_sort_by_score(gc);
	return gc;
}

void mg_gchain_free(mg_gchains_t *gs) /* gchain1.c:522-535 */
{
	int32_t i;
	if (gs == 0) return;
	for (i = 0; i < gs->n_gc; ++i) { free(gs->gc[i].p); free(gs->gc[i].ds.ds); free(gs->gc[i].ds.off); }
	free(gs->gc); free(gs->a); free(gs->lc);
	free(gs);
}
