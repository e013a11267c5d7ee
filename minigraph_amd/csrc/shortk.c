/*
 * shortk.c -- k-shortest walks from one vertex to a set of destination vertices:
 * mg_shortest_k (reference shortk.c:41-242), the reachability oracle of graph chaining and the
 * fallback bridge of mg_gchain_gen.
 *
 * Dijkstra-like search in which a vertex may be settled up to max_k (<= 15) times.  The frontier key
 * is dist<<32 | serial with a unique serial per pushed walk, so the settle order is fully determined
 * by arc order -- no dependence on the container.  The reference keeps the frontier in an AVL tree;
 * here it is an indexed binary min-heap (erase-by-handle for the "replace the longest walk" step).
 * The per-vertex list of walks is the reference's 15-slot max-heap with identical sift rules
 * (ksort.h:44-66): its slot 0 decides which walk gets replaced, also after settled walks had the low
 * half of their key rewritten (shortk.c:113), so the sift behaviour is reproduced rather than idealised.
 * Host code: irregular, allocation-heavy, ~1 call per read.
 */
#include <stdio.h>
#include <assert.h>
#include "hchain.h"

#define SK_EXT 1000 /* MG_SHORT_K_EXT, shortk.c:31 */

typedef struct {
	uint64_t di;     /* dist<<32 | serial (serial replaced by the settle rank once settled) */
	uint32_t v;
	int32_t pre;
	uint32_t hash;
	int32_t is_0;
	int32_t hpos;    /* position in the frontier heap, -1 when not queued */
} sk_node_t;

typedef struct { int32_t k; int32_t p[MG_MAX_SHORT_K]; } sk_topk_t; /* walks ending at one vertex (node indices) */

typedef struct {
	sk_node_t *nd; int32_t n_nd, m_nd;
	int32_t *heap; int32_t n_heap, m_heap;
	/* vertex -> topk, open addressing */
	uint32_t *hk; int32_t *hv; uint32_t hcap, hcnt;
	sk_topk_t *tk; int32_t n_tk, m_tk;
} sk_t;

/* ---- frontier: indexed min-heap on nd[].di ---- */
static void fh_up(sk_t *s, int32_t i)
{
	int32_t x = s->heap[i];
	while (i > 0) {
		int32_t par = (i - 1) >> 1;
		if (s->nd[s->heap[par]].di <= s->nd[x].di) break;
		s->heap[i] = s->heap[par], s->nd[s->heap[i]].hpos = i, i = par;
	}
	s->heap[i] = x, s->nd[x].hpos = i;
}
static void fh_down(sk_t *s, int32_t i)
{
	int32_t x = s->heap[i], n = s->n_heap;
	for (;;) {
		int32_t c = 2 * i + 1;
		if (c >= n) break;
		if (c + 1 < n && s->nd[s->heap[c + 1]].di < s->nd[s->heap[c]].di) ++c;
		if (s->nd[s->heap[c]].di >= s->nd[x].di) break;
		s->heap[i] = s->heap[c], s->nd[s->heap[i]].hpos = i, i = c;
	}
	s->heap[i] = x, s->nd[x].hpos = i;
}
static void fh_push(sk_t *s, int32_t x)
{
	if (s->n_heap == s->m_heap) { s->m_heap = s->m_heap ? s->m_heap * 2 : 64; s->heap = MGA_REALLOC(int32_t, s->heap, s->m_heap); }
	s->heap[s->n_heap] = x;
	fh_up(s, s->n_heap++);
}
static void fh_erase(sk_t *s, int32_t pos)
{
	int32_t x = s->heap[pos], last = s->heap[--s->n_heap];
	s->nd[x].hpos = -1;
	if (pos == s->n_heap) return;
	s->heap[pos] = last, s->nd[last].hpos = pos;
	fh_up(s, pos);
	fh_down(s, s->nd[last].hpos);
}

static int32_t new_node(sk_t *s, uint32_t v, int32_t d, uint32_t id)
{
	sk_node_t *p;
	if (s->n_nd == s->m_nd) { s->m_nd = s->m_nd ? s->m_nd * 2 : 64; s->nd = MGA_REALLOC(sk_node_t, s->nd, s->m_nd); }
	p = &s->nd[s->n_nd];
	p->v = v, p->di = (uint64_t)d << 32 | id, p->pre = -1, p->is_0 = 1, p->hpos = -1, p->hash = 0;
	return s->n_nd++;
}

/* ---- vertex -> topk ---- */
static sk_topk_t *vtx_put(sk_t *s, uint32_t v, int *absent)
{
	uint32_t i;
	if (s->hcnt * 2 >= s->hcap) {
		uint32_t ocap = s->hcap, j;
		uint32_t *ok = s->hk; int32_t *ov = s->hv;
		s->hcap = ocap ? ocap * 2 : 64;
		s->hk = MGA_MALLOC(uint32_t, s->hcap); s->hv = MGA_MALLOC(int32_t, s->hcap);
		for (j = 0; j < s->hcap; ++j) s->hv[j] = -1;
		for (j = 0; j < ocap; ++j)
			if (ov[j] >= 0) {
				uint32_t q = mga_hash_u32(ok[j]) & (s->hcap - 1);
				while (s->hv[q] >= 0) q = (q + 1) & (s->hcap - 1);
				s->hk[q] = ok[j], s->hv[q] = ov[j];
			}
		free(ok); free(ov);
	}
	i = mga_hash_u32(v) & (s->hcap - 1);
	while (s->hv[i] >= 0 && s->hk[i] != v) i = (i + 1) & (s->hcap - 1);
	*absent = s->hv[i] < 0;
	if (*absent) {
		if (s->n_tk == s->m_tk) { s->m_tk = s->m_tk ? s->m_tk * 2 : 32; s->tk = MGA_REALLOC(sk_topk_t, s->tk, s->m_tk); }
		s->hk[i] = v, s->hv[i] = s->n_tk++, ++s->hcnt;
	}
	return &s->tk[s->hv[i]];
}

/* the reference's max-heap sifts on q->p[] with "a < b" == nd[a].di < nd[b].di (ksort.h:44-66) */
#define TK_LT(s, a_, b_) ((s)->nd[(a_)].di < (s)->nd[(b_)].di)
static void tk_heapup(sk_t *s, int32_t n, int32_t *l)
{
	int32_t i, k = n - 1, tmp = l[k];
	while (k) {
		i = (k - 1) >> 1;
		if (TK_LT(s, tmp, l[i])) break;
		l[k] = l[i], k = i;
	}
	l[k] = tmp;
}
static void tk_heapdown(sk_t *s, int32_t i, int32_t n, int32_t *l)
{
	int32_t k = i, tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && TK_LT(s, l[k], l[k + 1])) ++k;
		if (TK_LT(s, l[k], tmp)) break;
		l[i] = l[k], i = k;
	}
	l[i] = tmp;
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

/* destinations grouped by vertex: first index of the group of vertex v in grp[] (sorted by v<<32|i), or -1 */
static int32_t grp_find(int32_t n, const uint64_t *grp, uint32_t v, int32_t *cnt)
{
	int32_t lo = 0, hi = n, e;
	while (lo < hi) { int32_t m = (lo + hi) >> 1; if ((uint32_t)(grp[m] >> 32) < v) lo = m + 1; else hi = m; }
	if (lo == n || (uint32_t)(grp[lo] >> 32) != v) return -1;
	for (e = lo; e < n && (uint32_t)(grp[e] >> 32) == v; ++e) {}
	*cnt = e - lo;
	return lo;
}

mga_pathv_t *mga_shortest_k(const gfa_t *g, uint32_t src, int32_t n_dst, mga_path_dst_t *dst, int32_t max_dist, int32_t max_k, int32_t *n_pathv)
{
	sk_t S;
	sk_topk_t *q;
	int32_t i, j, n_done = 0, n_found, x, *out = 0, n_out = 0, m_out = 0;
	uint32_t id = 0;
	int8_t *dst_done;
	uint64_t *grp;
	mga_pathv_t *ret = 0;
	int absent;

	if (n_pathv) *n_pathv = 0;
	if (n_dst <= 0) return 0;
	for (i = 0; i < n_dst; ++i) { /* shortk.c:61-67 */
		mga_path_dst_t *t = &dst[i];
		if (t->inner) t->dist = 0, t->n_path = 1, t->path_end = -1;
		else t->dist = -1, t->n_path = 0, t->path_end = -1;
	}
	if (max_k > MG_MAX_SHORT_K) max_k = MG_MAX_SHORT_K;
	memset(&S, 0, sizeof S);
	dst_done = MGA_CALLOC(int8_t, n_dst);
	grp = MGA_MALLOC(uint64_t, n_dst);
	for (i = 0; i < n_dst; ++i) grp[i] = (uint64_t)dst[i].v << 32 | (uint32_t)i;
	qsort(grp, (size_t)n_dst, 8, cmp_u64);

	x = new_node(&S, src, 0, id++);
	S.nd[x].hash = mga_hash_u32(src);
	fh_push(&S, x);
	q = vtx_put(&S, src, &absent);
	q->k = 1, q->p[0] = x;

	while (S.n_heap > 0) {
		int32_t r = S.heap[0], nv, off, cnt;
		const gfa_arc_t *av;
		fh_erase(&S, 0); /* the closest unsettled walk */
		if (n_out == m_out) { m_out = m_out ? m_out * 2 : 16; out = MGA_REALLOC(int32_t, out, m_out); }
		S.nd[r].di = S.nd[r].di >> 32 << 32 | (uint32_t)n_out;
		out[n_out++] = r;

		off = grp_find(n_dst, grp, S.nd[r].v, &cnt);
		if (off >= 0) { /* reached a destination vertex (shortk.c:116-153) */
			int32_t dist = (int32_t)(S.nd[r].di >> 32);
			for (j = 0; j < cnt; ++j) {
				mga_path_dst_t *t = &dst[(int32_t)grp[off + j]];
				int32_t done = 0;
				if (t->inner) done = 1;
				else {
					int32_t copy = 0;
					if (t->n_path == 0) copy = 1;
					else if (t->target_dist >= 0) {
						if (dist == t->target_dist && t->check_hash && S.nd[r].hash == t->target_hash) copy = 1, done = 1;
						else {
							int32_t d0 = t->dist, d1 = dist;
							d0 = d0 > t->target_dist ? d0 - t->target_dist : t->target_dist - d0;
							d1 = d1 > t->target_dist ? d1 - t->target_dist : t->target_dist - d1;
							if (d1 < d0) copy = 1;
						}
					}
					if (copy) {
						t->path_end = n_out - 1, t->dist = dist, t->hash = S.nd[r].hash, t->is_0 = S.nd[r].is_0;
						if (t->target_dist >= 0) {
							if (dist == t->target_dist && t->check_hash && S.nd[r].hash == t->target_hash) done = 1;
							else if (dist > t->target_dist + SK_EXT) done = 1;
						}
					}
					++t->n_path;
					if ((int32_t)t->n_path >= max_k) done = 1;
				}
				if (dst_done[off + j] == 0 && done) dst_done[off + j] = 1, ++n_done;
			}
			if (n_done == n_dst) break;
		}

		nv = (int32_t)gfa_arc_n(g, S.nd[r].v);
		av = gfa_arc_a(g, S.nd[r].v);
		for (i = 0; i < nv; ++i) { /* relax every arc, in arc order (shortk.c:157-188) */
			const gfa_arc_t *ai = &av[i];
			int32_t d = (int32_t)(S.nd[r].di >> 32) + (int32_t)(uint32_t)ai->v_lv;
			if (d > max_dist) continue;
			q = vtx_put(&S, ai->w, &absent);
			if (absent) q->k = 0;
			if (q->k < max_k) {
				x = new_node(&S, ai->w, d, id++);
				S.nd[x].pre = n_out - 1;
				S.nd[x].hash = S.nd[r].hash + mga_hash_u32(ai->w);
				S.nd[x].is_0 = S.nd[r].is_0;
				if (ai->rank > 0) S.nd[x].is_0 = 0;
				fh_push(&S, x);
				q = vtx_put(&S, ai->w, &absent); /* node pool may have moved nothing here, but tk pool could: refetch */
				q->p[q->k++] = x;
				tk_heapup(&S, q->k, q->p);
			} else if ((int32_t)(S.nd[q->p[0]].di >> 32) > d) { /* shorter than the longest kept walk: replace it */
				x = q->p[0];
				if (S.nd[x].hpos >= 0) {
					fh_erase(&S, S.nd[x].hpos);
					S.nd[x].di = (uint64_t)d << 32 | (id++);
					S.nd[x].pre = n_out - 1;
					S.nd[x].hash = S.nd[r].hash + mga_hash_u32(ai->w);
					S.nd[x].is_0 = S.nd[r].is_0;
					if (ai->rank > 0) S.nd[x].is_0 = 0;
					fh_push(&S, x);
					tk_heapdown(&S, 0, q->k, q->p);
				} else { /* shortk.c:182-186 */
					fprintf(stderr, "Warning: logical bug in gfa_shortest_k(): q->k=%d,q->p[0]->{d,i}={%d,%d},d=%d,src=%u,max_dist=%d,n_dst=%d\n", q->k,
							(int32_t)(S.nd[x].di >> 32), (int32_t)S.nd[x].di, d, src, max_dist, n_dst);
					free(S.nd); free(S.heap); free(S.hk); free(S.hv); free(S.tk); free(out); free(dst_done); free(grp);
					return 0;
				}
			}
		}
	}

	for (i = 0, n_found = 0; i < n_dst; ++i) if (dst[i].n_path > 0) ++n_found;
	if (n_found > 0 && n_pathv) { /* backtrack array (shortk.c:202-236) */
		int32_t n, *trans = MGA_CALLOC(int32_t, n_out > 0 ? n_out : 1);
		for (i = 0; i < n_dst; ++i) {
			mga_path_dst_t *t = &dst[i];
			if (t->n_path > 0 && t->target_dist >= 0 && t->path_end >= 0) trans[(int32_t)S.nd[out[t->path_end]].di] = 1;
		}
		for (i = 0; i < n_out; ++i) {
			int32_t off, cnt;
			off = grp_find(n_dst, grp, S.nd[out[i]].v, &cnt);
			if (off >= 0)
				for (j = off; j < off + cnt; ++j)
					if (dst[j].target_dist < 0) trans[i] = 1; /* NB: indexes dst[] by group position, as the reference does (shortk.c:215-217) */
		}
		for (i = n_out - 1; i >= 0; --i)
			if (trans[i] && S.nd[out[i]].pre >= 0) trans[S.nd[out[i]].pre] = 1;
		for (i = n = 0; i < n_out; ++i) trans[i] = trans[i] ? n++ : -1;
		*n_pathv = n;
		ret = MGA_MALLOC(mga_pathv_t, n > 0 ? n : 1);
		for (i = 0; i < n_out; ++i) {
			mga_pathv_t *p;
			if (trans[i] < 0) continue;
			p = &ret[trans[i]];
			p->v = S.nd[out[i]].v, p->d = (uint32_t)(S.nd[out[i]].di >> 32);
			p->pre = S.nd[out[i]].pre < 0 ? S.nd[out[i]].pre : trans[S.nd[out[i]].pre];
		}
		for (i = 0; i < n_dst; ++i)
			if (dst[i].path_end >= 0) dst[i].path_end = trans[dst[i].path_end];
		free(trans);
	}
	free(S.nd); free(S.heap); free(S.hk); free(S.hv); free(S.tk); free(out); free(dst_done); free(grp);
	return ret;
}
