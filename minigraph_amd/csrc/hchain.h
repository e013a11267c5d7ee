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
mg128_t *mga_lchain_rmq_finish2(int bw, int min_cnt, int min_sc, int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t, int *n_u_, uint64_t **u_, int keep_arrays); /* keep_arrays: f, p, v, t belong to the caller (pinned staging of the device pass) */

/* forward pass of mg_lchain_dp (lchain.c:168-207) over the x-sorted anchors of ONE long read on a host thread (single segment, not cDNA); t zeroed by the caller */
void mga_lchain_dp_fwd(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float pen_gap, float pen_skip,
					   int64_t n, const mg128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t);

#endif
