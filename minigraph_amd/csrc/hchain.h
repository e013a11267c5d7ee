/* hchain.h -- host-side chaining stages shared by rmq.c, gchain.c and mapper.c */
#ifndef MGA_HCHAIN_H
#define MGA_HCHAIN_H
#include "mga_host.h"

/* mg_chain_backtrack (lchain.c:27-77): returns malloc'ed u[] of *n_u_ (+extra_u spare) entries; fills v[] */
uint64_t *mga_chain_backtrack(int64_t n, const int32_t *f, const int64_t *p, int32_t *v, int32_t *t, int32_t min_cnt, int32_t min_sc,
							  int32_t max_drop, int32_t extra_u, int32_t *n_u_, int32_t *n_v_);
/* compact_a (lchain.c:79-112): returns a malloc'ed anchor array of n_v entries; u[] is reordered in place */
mg128_t *mga_compact_a(int32_t n_u, uint64_t *u, int32_t n_v, const int32_t *v, const mg128_t *a);

mg128_t *mga_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
						float pen_gap, float pen_skip, int64_t n, const mg128_t *a, int *n_u_, uint64_t **u_);
/* the same in two steps: forward pass over anchors [beg,end) (beg = 0 or the start of a (segment,strand) group; f, p, v, t are arrays of the
 * whole read, t zeroed), then backtracking + compaction over all n anchors (frees f, p, v, t) */
void mga_lchain_rmq_fwd(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, float pen_gap, float pen_skip,
						int64_t beg, int64_t end, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t);
mg128_t *mga_lchain_rmq_finish(int bw, int min_cnt, int min_sc, int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t, int *n_u_, uint64_t **u_);

/* mg_lchain_gen (lchain.c:374-408) */
mg_lchain_t *mga_lchain_gen(uint32_t hash, int qlen, int n_u, const uint64_t *u, const mg128_t *a);
/* the n_lc > 1 clean-up block of mg_map_frag (map-algo.c:424-445); returns the new n_lc */
int32_t mga_lchain_cleanup(const mg_mapopt_t *opt, int32_t n_lc, mg_lchain_t *lc, mg128_t *a);
/* mg_update_anchors (lchain.c:431-441) */
void mga_update_anchors(int32_t n_a, mg128_t *a, int32_t n, const int32_t *mini_pos);

/* ---- graph chaining (gchain.c, shortk.c, gwfa.c) ---- */
typedef struct { /* mg_path_dst_t, mgpriv.h:40-52 */
	uint32_t v;
	int32_t target_dist;
	uint32_t target_hash;
	uint32_t meta:30, check_hash:1, inner:1;
	int32_t qlen;
	uint32_t n_path:31, is_0:1;
	int32_t path_end;
	int32_t dist;
	uint32_t hash;
} mga_path_dst_t;
typedef struct { uint32_t v, d; int32_t pre; } mga_pathv_t; /* mg_pathv_t, mgpriv.h:54-57 */

mga_pathv_t *mga_shortest_k(const gfa_t *g, uint32_t src, int32_t n_dst, mga_path_dst_t *dst, int32_t max_dist, int32_t max_k, int32_t *n_pathv);

/* GWFA between (v0,off0) and (v1,off1): returns edit distance or -1; *path = malloc'ed vertex walk of *nv vertices */
int32_t mga_gwfa_bridge(const gfa_t *g, const gfa_edseq_t *es, int32_t ql, const char *q, uint32_t v0, int32_t off0, uint32_t v1, int32_t off1,
						int32_t max_lag, int32_t s_term, int32_t **path, int32_t *nv);

int32_t mga_gchain1_dp(const gfa_t *g, int32_t *n_lc_, mg_lchain_t *lc, int32_t qlen, int32_t max_dist_g, int32_t max_dist_q, int32_t bw, int32_t max_skip,
					   int32_t ref_bonus, float chn_pen_gap, float chn_pen_skip, float mask_level, const mg128_t *an, uint64_t **u_);
mg_gchains_t *mga_gchain_gen(const gfa_t *g, const gfa_edseq_t *es, int32_t n_u, const uint64_t *u, mg_lchain_t *lc, const mg128_t *a, uint32_t hash,
							 int32_t min_gc_cnt, int32_t min_gc_score, int32_t gdp_max_ed, int32_t n_seg, const char *qseq);

/* ---- post-processing (gcpost.c) ---- */
void mga_gchain_sort_by_score(mg_gchains_t *gcs);
void mga_gchain_set_parent(float mask_level, int n, mg_gchain_t *r, int sub_diff, int hard_mask_level);
int mga_gchain_flt_sub(float pri_ratio, int min_diff, int best_n, int n, mg_gchain_t *r);
void mga_gchain_drop_flt(mg_gchains_t *gcs);
void mga_gchain_set_mapq(mg_gchains_t *gcs, int qlen, int max_mini, int min_gc_score);

#endif
