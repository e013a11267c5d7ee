/* wfachain.c -- host half of miniwfa's k-mer chained fallback for anchor gaps whose exact WFA exceeds 1e8 wavefront cells.
 *
 * Reference: mwf_wfa_auto() (miniwfa.c:824-834) first runs the exact WFA with max_iter = 1e8; if that gives up it calls mwf_wfa_chain()
 * (miniwfa.c:776-822): 13-mers occurring <= 2 times in both sequences are matched, the longest co-linear subset is kept (LIS),
 * short co-diagonal runs (< 30 bp) are dropped, and the stretches between the surviving anchors are closed one by one -- '=' for adjacent
 * anchors on a diagonal, a D+I pair for long stretches without shared k-mers, a plain D or I for one-sided stretches, and an exact
 * (unbounded) WFA for everything else.
 *
 * Split of the work here: this file makes the PLAN (the list of literal CIGAR ops and of sub-problems, a few kilobases of integer work on
 * sequences that already are on the host side of the boundary) and stitches the final CIGAR; the sub-problems -- all of the DP -- go
 * back to the device ladder (k_wfa_sched.hip: wfs_fallback()).  Nothing here aligns bases.
 */
#include <stdlib.h>
#include <string.h>
#include "mga_host.h"
#include "wfachain.h"

/* all k-mers without ambiguous bases: (kmer<<1 | rid) << 32 | end position (mg_fc_kmer, miniwfa.c:644-656) */
static int32_t wc_kmers(int32_t len, const char *seq, int32_t rid, int32_t k, uint64_t *out)
{
	const uint64_t mask = (1ULL << 2 * k) - 1;
	uint64_t km = 0;
	int32_t n = 0, run = 0;
	for (int32_t i = 0; i < len; ++i) {
		const int c = mga_nt4_table[(uint8_t)seq[i]];
		if (c >= 4) { run = 0, km = 0; continue; }
		km = (km << 2 | (uint64_t)c) & mask;
		if (++run >= k) out[n++] = (km << 1 | (uint64_t)rid) << 32 | (uint32_t)i;
	}
	return n;
}

/* Longest strictly increasing subsequence of v[0..n) (mg_lis_64, miniwfa.c:620-639): tail[l] = index of the last element that ended a
 * subsequence of length l when it was visited; element i extends the longest one whose tail value is below v[i].  Which of several LIS
 * of equal length comes out is decided by this visiting order, so the order is kept.  Returns the length; idx[0..len) = the indices. */
static int32_t wc_lis(int32_t n, const uint64_t *v, int32_t *idx)
{
	if (n <= 0) return 0;
	int32_t *tail = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
	int32_t *prev = (int32_t*)malloc((size_t)n * sizeof(int32_t));
	int32_t len = 0;
	tail[0] = -1;
	for (int32_t i = 0; i < n; ++i) {
		int32_t a = 0, b = len; /* largest l in [0,len] with l == 0 or v[tail[l]] < v[i] (tails are increasing in l) */
		while (a < b) {
			const int32_t m = (a + b + 1) >> 1;
			if (v[tail[m]] < v[i]) a = m;
			else b = m - 1;
		}
		prev[i] = tail[a], tail[a + 1] = i;
		if (a + 1 > len) len = a + 1;
	}
	for (int32_t l = len - 1, i = tail[len]; l >= 0; --l, i = prev[i]) idx[l] = i;
	free(tail); free(prev);
	return len;
}

/* co-linear k-mer matches between s1 and s2 as (end1 << 32 | end2), increasing in both (mg_chain, miniwfa.c:658-710) */
static uint64_t *wc_anchors(int32_t l1, const char *s1, int32_t l2, const char *s2, int32_t k, int32_t max_occ, int32_t *n_out)
{
	*n_out = 0;
	if (l1 < k || l2 < k) return 0;
	uint64_t *km = (uint64_t*)malloc((size_t)(l1 + l2) * sizeof(uint64_t));
	int32_t n_km = wc_kmers(l1, s1, 0, k, km);
	n_km += wc_kmers(l2, s2, 1, k, km + n_km);
	mga_ksort_u64(n_km, km);
	/* groups of equal k-mers: s1's occurrences first (rid bit), each side by position */
	int64_t n_pair = 0, m_pair = 0;
	uint64_t *pair = 0;
	for (int32_t g0 = 0, i = 1; i <= n_km; ++i) {
		if (i < n_km && km[g0] >> 33 == km[i] >> 33) continue;
		int32_t mid = g0;
		while (mid < i && (km[mid] >> 32 & 1) == 0) ++mid;
		if (mid > g0 && mid < i && mid - g0 <= max_occ && i - mid <= max_occ)
			for (int32_t s = g0; s < mid; ++s)
				for (int32_t t = mid; t < i; ++t) {
					if (n_pair == m_pair) {
						m_pair = m_pair ? m_pair * 2 : 256;
						pair = (uint64_t*)realloc(pair, (size_t)m_pair * sizeof(uint64_t));
					}
					pair[n_pair++] = km[s] << 32 | (uint32_t)km[t];
				}
		g0 = i;
	}
	free(km);
	if (n_pair == 0) { free(pair); return 0; }
	if (n_pair > 0x7fffffff) { free(pair); return 0; } /* cannot happen: <= 4 pairs per k-mer of int32 sequences */
	mga_ksort_u64(n_pair, pair);                         /* by position in s1, then in s2 */
	for (int64_t i = 0; i < n_pair; ++i) pair[i] = pair[i] >> 32 | pair[i] << 32; /* LIS on (pos2, pos1) */
	int32_t *idx = (int32_t*)malloc((size_t)n_pair * sizeof(int32_t));
	const int32_t n_lis = wc_lis((int32_t)n_pair, pair, idx);
	uint64_t *out = (uint64_t*)malloc((size_t)(n_lis ? n_lis : 1) * sizeof(uint64_t));
	for (int32_t i = 0; i < n_lis; ++i) {
		const uint64_t v = pair[idx[i]];
		out[i] = v >> 32 | v << 32;
	}
	free(idx); free(pair);
	*n_out = n_lis;
	return out;
}

/* drop anchors of co-diagonal runs shorter than min_l bases (wf_anchor_filter, miniwfa.c:755-774).  The run [start,i) ends where the step
 * to anchor i (or to the end of both sequences) leaves the diagonal of anchor `start`; its length counts k for the first anchor. */
static int32_t wc_filter(int32_t n, uint64_t *a, int32_t tl, int32_t ql, int32_t k, int32_t min_l)
{
	int32_t px = 0, py = 0, lx = 0, start = -1, run = 0;
	for (int32_t i = 0; i <= n; ++i) {
		const int32_t x = i == n ? tl : (int32_t)(a[i] >> 32) + 1;
		const int32_t y = i == n ? ql : (int32_t)a[i] + 1;
		if (x - px != y - py) {
			if (run < min_l)
				for (int32_t j = start > 0 ? start : 0; j < i; ++j) a[j] = 0;
			px = x, py = y, start = i, run = k;
		} else run += x - lx;
		lx = x;
	}
	int32_t m = 0;
	for (int32_t i = 0; i < n; ++i)
		if (a[i] != 0) a[m++] = a[i];
	return m;
}

/* k-mer similarity of two stretches (mwf_ksim, miniwfa.c:712-738): shared k-mer occurrences over all, the larger of both sides */
static double wc_ksim(int32_t l1, const char *s1, int32_t l2, const char *s2, int32_t k)
{
	if (l1 < k || l2 < k) return 0;
	uint64_t *km = (uint64_t*)malloc((size_t)(l1 + l2) * sizeof(uint64_t));
	int32_t n_km = wc_kmers(l1, s1, 0, k, km), tot1 = 0, tot2 = 0, sh = 0;
	n_km += wc_kmers(l2, s2, 1, k, km + n_km);
	mga_ksort_u64(n_km, km);
	for (int32_t g0 = 0, i = 1; i <= n_km; ++i) {
		if (i < n_km && km[g0] >> 33 == km[i] >> 33) continue;
		int32_t mid = g0;
		while (mid < i && (km[mid] >> 32 & 1) == 0) ++mid;
		const int32_t c1 = mid - g0, c2 = i - mid;
		tot1 += c1, tot2 += c2;
		if (c1 > 0 && c2 > 0) sh += c1 < c2 ? c1 : c2;
		g0 = i;
	}
	free(km);
	const double p1 = (double)sh / tot1, p2 = (double)sh / tot2; /* 0/0 = NaN compares false both ways, like the reference */
	return p1 > p2 ? p1 : p2;
}

static int wc_push(mga_wc_plan_t *p, mga_wc_el_t e)
{
	if (p->n == p->m) {
		p->m = p->m ? p->m * 2 : 64;
		mga_wc_el_t *t = (mga_wc_el_t*)realloc(p->el, (size_t)p->m * sizeof(*t));
		if (t == 0) return -1;
		p->el = t;
	}
	p->el[p->n++] = e;
	return 0;
}

static inline int32_t wc_gap_cost(const mga_wc_par_t *o, int32_t l)
{
	const int32_t a = o->o1 + l * o->e1, b = o->o2 + l * o->e2;
	return b < a ? b : a;
}

void mga_wc_par_default(mga_wc_par_t *o) /* mwf_opt_init, miniwfa.c:12-22 */
{
	o->x = 4, o->o1 = 4, o->e1 = 2, o->o2 = 15, o->e2 = 1;
	o->kmer = 13, o->max_occ = 2, o->min_len = 30;
}

int mga_wfa_chain_plan(const mga_wc_par_t *o, int32_t tl, const char *ts, int32_t ql, const char *qs, mga_wc_plan_t *p)
{
	int32_t n_a, x0 = 0, y0 = 0;
	memset(p, 0, sizeof(*p));
	mga_tables_init();
	uint64_t *a = wc_anchors(tl, ts, ql, qs, o->kmer, o->max_occ, &n_a);
	n_a = wc_filter(n_a, a, tl, ql, o->kmer, o->min_len);
	for (int32_t i = 0; i <= n_a; ++i) {
		const int32_t x1 = i == n_a ? tl : (int32_t)(a[i] >> 32) + 1;
		const int32_t y1 = i == n_a ? ql : (int32_t)a[i] + 1;
		const int32_t dx = x1 - x0, dy = y1 - y0;
		mga_wc_el_t e;
		memset(&e, 0, sizeof(e));
		int rc = 0;
		if (i < n_a && dx == dy && dx <= o->kmer) {           /* the next anchor overlaps or abuts the previous one: all matches */
			e.op = 7, e.len = dx, rc = wc_push(p, e);
		} else if (dx > 0 && dy > 0) {
			if (dx >= 10000 && dy >= 10000 && wc_ksim(dx, ts + x0, dy, qs + y0, o->kmer) < 0.02) { /* unrelated: delete one, insert the other */
				e.op = 2, e.len = dx, rc = wc_push(p, e);
				e.op = 1, e.len = dy; if (rc == 0) rc = wc_push(p, e);
				p->score += o->o2 * 2 + o->e2 * (dx + dy);
			} else {
				e.sub = 1, e.x0 = x0, e.y0 = y0, e.tl = dx, e.ql = dy, rc = wc_push(p, e);
				++p->n_sub;
			}
		} else if (dx > 0) {
			e.op = 2, e.len = dx, rc = wc_push(p, e), p->score += wc_gap_cost(o, dx);
		} else if (dy > 0) {
			e.op = 1, e.len = dy, rc = wc_push(p, e), p->score += wc_gap_cost(o, dy);
		}
		if (rc < 0) { free(a); mga_wfa_chain_plan_free(p); return -1; }
		x0 = x1, y0 = y1;
	}
	free(a);
	return 0;
}

void mga_wfa_chain_plan_free(mga_wc_plan_t *p)
{
	free(p->el);
	memset(p, 0, sizeof(*p));
}

/* the ops of the plan in order, runs of one op merged the way wf_cigar_push1()/wf_cigar_push() (miniwfa.c:51-63,742-753) do it: a literal
 * op always merges with an equal op before it; of a sub-problem's CIGAR only the FIRST op may merge, the rest is appended as it is. */
int64_t mga_wfa_chain_stitch(const mga_wc_plan_t *p, const uint32_t *const *sub_cig, const int32_t *sub_n, uint32_t *out, int64_t cap)
{
	int64_t n = 0;
	int32_t si = 0;
#define WC_PUSH1(op_, len_) do { \
		if (n > 0 && (out[n - 1] & 0xf) == (uint32_t)(op_)) out[n - 1] += (uint32_t)(len_) << 4; \
		else { if (n == cap) return -1; out[n++] = (uint32_t)(len_) << 4 | (uint32_t)(op_); } \
	} while (0)
	for (int32_t i = 0; i < p->n; ++i) {
		const mga_wc_el_t *e = &p->el[i];
		if (!e->sub) { WC_PUSH1(e->op, e->len); continue; }
		const uint32_t *c = sub_cig[si];
		const int32_t nc = sub_n[si++];
		if (nc == 0) continue;
		WC_PUSH1(c[0] & 0xf, c[0] >> 4);
		if (n + nc - 1 > cap) return -1;
		memcpy(out + n, c + 1, (size_t)(nc - 1) * sizeof(uint32_t));
		n += nc - 1;
	}
#undef WC_PUSH1
	return n;
}
