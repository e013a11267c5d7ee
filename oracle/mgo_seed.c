/*
 * ORACLE restatement of the minimizer index lookup semantics (index.c:50-72,115-165,186-209) and of
 * seed collection (map-algo.c:58-91 collect_matches, :152-192 collect_seed_hits).
 *
 * The reference index is 2^b khashl buckets; what a lookup RETURNS is independent of that layout:
 * for a minimizer hash h, the list of y = seg<<32|lastPos<<1|strand of every occurrence in the graph,
 * ascending (singletons directly, index.c:56-58; lists sorted by radix_sort_gfa64, index.c:156).
 * The oracle keeps one sorted (hash, y) array and binary-searches it.
 */
#include <stdlib.h>
#include <string.h>
#include "mgo.h"

static int cmp128(const void *a_, const void *b_)
{
	const mgo128_t *a = (const mgo128_t*)a_, *b = (const mgo128_t*)b_;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	return a->y < b->y ? -1 : a->y > b->y;
}

mgo_idx_t *mgo_idx_build(int32_t n_seg, const char *const *seq, const int32_t *len, int32_t w, int32_t k)
{
	mgo_idx_t *idx = (mgo_idx_t*)calloc(1, sizeof(mgo_idx_t));
	mgo128_t *a = 0;
	int64_t n = 0, m = 0, i, nk;
	int32_t s;
	for (s = 0; s < n_seg; ++s) { /* index.c:200-205: one sketch per segment, rid = segment id */
		int64_t r;
		if (len[s] <= 0) continue;
		if (n + len[s] + 16 > m) { m = (n + len[s] + 16) * 2; a = (mgo128_t*)realloc(a, m * sizeof(mgo128_t)); }
		r = mgo_sketch(seq[s], len[s], w, k, (uint32_t)s, a + n, m - n);
		if (r < 0) { m = n - r + 16; a = (mgo128_t*)realloc(a, m * sizeof(mgo128_t)); r = mgo_sketch(seq[s], len[s], w, k, (uint32_t)s, a + n, m - n); }
		n += r;
	}
	for (i = 0; i < n; ++i) a[i].x >>= 8; /* the index key is the hash without the span byte (index.c:102,126) */
	qsort(a, n, sizeof(mgo128_t), cmp128);
	for (i = 0, nk = 0; i < n; ++i) if (i == 0 || a[i].x != a[i-1].x) ++nk;
	idx->n_keys = nk, idx->n_pos = n;
	idx->key = (uint64_t*)malloc((nk + 1) * 8);
	idx->off = (int64_t*)malloc((nk + 1) * 8);
	idx->pos = (uint64_t*)malloc((n + 1) * 8);
	for (i = 0, nk = 0; i < n; ++i) {
		if (i == 0 || a[i].x != a[i-1].x) idx->key[nk] = a[i].x, idx->off[nk++] = i;
		idx->pos[i] = a[i].y;
	}
	idx->off[nk] = n;
	free(a);
	return idx;
}

void mgo_idx_free(mgo_idx_t *idx)
{
	if (idx == 0) return;
	free(idx->key); free(idx->off); free(idx->pos); free(idx);
}

const uint64_t *mgo_idx_get(const mgo_idx_t *idx, uint64_t minier, int32_t *n)
{
	int64_t lo = 0, hi = idx->n_keys - 1;
	*n = 0;
	while (lo <= hi) {
		int64_t mid = (lo + hi) >> 1;
		if (idx->key[mid] < minier) lo = mid + 1;
		else if (idx->key[mid] > minier) hi = mid - 1;
		else { *n = (int32_t)(idx->off[mid + 1] - idx->off[mid]); return &idx->pos[idx->off[mid]]; }
	}
	return 0;
}

/*
 * collect_matches + collect_seed_hits for one single-segment query (n_segs==1, seg_id 0), without
 * the MG_M_NO_DIAG filter (map-algo.c:165-176 is off for lr).  Anchor encoding (map-algo.c:177-186):
 *   x = seg<<33 | rev<<32 | rpos              forward: rpos = lastPos on the segment
 *                                              reverse: rpos = seglen - (lastPos + 1 - span) - 1
 *   y = min(occ,255)<<56 | tandem<<42 | span<<32 | qpos
 * then radix_sort_128x by x.  rep_len: union length of query intervals covered by minimizers with
 * occ >= max_occ (map-algo.c:72-79,88).  mini_pos[]: qpos of every KEPT minimizer, even with 0 hits.
 */
int64_t mgo_collect_seed_hits(const mgo_idx_t *idx, const int32_t *seg_len, int32_t max_occ,
							  int64_t n_mz, const mgo128_t *mz, mgo128_t *a,
							  int32_t *rep_len, int32_t *n_mini_pos, int32_t *mini_pos)
{
	int64_t i, n_a = 0;
	int32_t rep_st = 0, rep_en = 0, rl = 0, nmp = 0;
	for (i = 0; i < n_mz; ++i) {
		uint64_t h = mz[i].x >> 8;
		uint32_t q_pos = (uint32_t)mz[i].y, q_span = (uint32_t)(mz[i].x & 0xff);
		int32_t t, k;
		const uint64_t *cr = mgo_idx_get(idx, h, &t);
		if (t >= max_occ) {
			int32_t en = (int32_t)(q_pos >> 1) + 1, st = en - (int32_t)q_span;
			if (st > rep_en) rl += rep_en - rep_st, rep_st = st, rep_en = en;
			else rep_en = en;
			continue;
		}
		if (mini_pos) mini_pos[nmp] = (int32_t)(q_pos >> 1);
		++nmp;
		if (a) {
			int tandem = (i > 0 && mz[i-1].x >> 8 == h) || (i + 1 < n_mz && mz[i+1].x >> 8 == h);
			for (k = 0; k < t; ++k) {
				uint64_t r = cr[k], seg = r >> 32;
				int32_t rpos = (int32_t)((uint32_t)r >> 1);
				mgo128_t *p = &a[n_a + k];
				if ((r & 1) == (q_pos & 1)) p->x = seg << 33 | (uint64_t)rpos;
				else p->x = seg << 33 | 1ULL << 32 | (uint64_t)(uint32_t)(seg_len[seg] - (rpos + 1 - (int32_t)q_span) - 1);
				p->y = (uint64_t)q_span << 32 | q_pos >> 1;
				if (tandem) p->y |= 1ULL << 42;
				p->y |= (uint64_t)(t < 255 ? t : 255) << 56;
			}
		}
		n_a += t;
	}
	rl += rep_en - rep_st;
	if (rep_len) *rep_len = rl;
	if (n_mini_pos) *n_mini_pos = nmp;
	if (a) mgo_sort128x(a, n_a);
	return n_a;
}
