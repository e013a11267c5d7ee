/* mapper.h -- host-side batch object shared by mapper.c and the stage tests */
#ifndef MGA_MAPPER_H
#define MGA_MAPPER_H
#include "mga_host.h"

typedef struct mga_batch_s mga_batch_t;

/* seqs/qnames/qlens/q_off are borrowed for the lifetime of the batch; q_off[i] = offset of read i in the device read buffer */
mga_batch_t *mga_batch_init(const mg_idx_t *gi, const mg_mapopt_t *opt, int n, const int *qlens, const char **seqs, const char **qnames,
							const int64_t *q_off, int n_threads);
void mga_batch_lchain_par(const mg_idx_t *gi, const mg_mapopt_t *opt, int qlen_max, mga_lchain_par_t *par);
/* host half 1 (map-algo.c:407-474 + the gap list of galign.c:53-125).  Inputs are what the GPU stages produce:
 * per read n_mz, rep_len, mini_pos; and either the DP chains (nu, nb, u, a laid out at a_off) or, with a_is_raw, the sorted anchors */
int mga_batch_chain(mga_batch_t *b, const int32_t *n_mz, const int32_t *rep_len, const int32_t *mini_pos, const int64_t *mini_off,
					const int32_t *nu, const int32_t *nb, const uint64_t *u, const mg128_t *a, const int64_t *a_off, int a_is_raw, const int32_t *rescue_flag);
int64_t mga_batch_n_wfa(const mga_batch_t *b);
int64_t mga_batch_wfa_target_bytes(const mga_batch_t *b);
void mga_batch_wfa_export(const mga_batch_t *b, mga_wfa_prob_t *prob, char *tseq);
/* host half 2: CIGAR stitching + ds from the WFA results */
int mga_batch_finish(mga_batch_t *b, const mga_wfa_res_t *res, const uint32_t *pool);
int mga_batch_finish_ordered(mga_batch_t *b, const int32_t *ncig, const int64_t *off, const uint32_t *ord);
mg_gchains_t **mga_batch_take_results(mga_batch_t *b);
void mga_batch_stats(const mga_batch_t *b, mga_stats_t *st);
void mga_batch_destroy(mga_batch_t *b);
#endif
