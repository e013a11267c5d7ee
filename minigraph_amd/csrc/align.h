/* align.h -- plan/apply interface of the base-alignment host half (align.c) */
#ifndef MGA_ALIGN_H
#define MGA_ALIGN_H
#include "mga_host.h"

/* mga_cigitem_t (mga_dev.h): op >= 0: ready operator (op, len = val); op == -1: WFA problem #val of this pool */

typedef struct { /* per host thread accumulators of one batch */
	char *tseq; int64_t n_t, m_t;                  /* spliced target sequences */
	mga_wfa_prob_t *prob; int64_t n_prob, m_prob;  /* t_off relative to this pool; q_off absolute in the device read buffer */
	mga_cigitem_t *item; int64_t n_item, m_item;
	int64_t wfa_t_bases, wfa_q_bases;
	/* text mode (the device writes cg:Z / ds:Z): printed chains and the vertices of their walks; offsets local to this pool */
	mga_txt_chain_t *chain; int64_t n_chain, m_chain;
	uint32_t *vert; int64_t n_vert, m_vert;
	/* round 5, text mode on a device that holds the graph's sequence: a problem's target is not spliced HERE (15 M small copies out of 6 GB of oriented segment
	 * sequence per 125 000 reads, then a gigabyte through pinned memory and PCIe) but DESCRIBED -- from base x0 + 1 of walk vertex vert[lc0] to base x1 of vert[lc0 + n_lc] --
	 * and spliced on the device from its own segment images (k_plan.hip: k_plan_target).  want_src: descriptors instead of bytes (n_t still counts the bytes: t_off) */
	int want_src;
	mga_plan_src_t *src; int64_t m_src;            /* one per problem of this pool; lc0 local to this pool's vert[] */
} mga_tpool_t;

/* vert_beg (want_src only): index in tp->vert of the chain's first walk vertex, pushed by the caller BEFORE this call */
void mga_plan_cigar(const gfa_t *g, const gfa_edseq_t *es, const mg_gchains_t *gt, int32_t gc_idx, int64_t q_base, mga_tpool_t *tp, int64_t vert_beg);
/* where the CIGAR of WFA problem j lives: either the kernels' raw output (res[j].cig_off into the pool, completion order)
 * or, when ord != NULL, the device-gathered copy in problem order (ncig[j] ops at ord + off[j]) */
typedef struct { const mga_wfa_res_t *res; const uint32_t *pool; const int32_t *ncig; const int64_t *off; const uint32_t *ord; } mga_cigsrc_t;
int mga_apply_cigar(mg_gchains_t *gt, int32_t gc_idx, const mga_cigitem_t *item, int64_t n_item, int64_t prob_base, const mga_cigsrc_t *src);
void mga_gen_ds(const gfa_edseq_t *es, const char *qseq, mg_gchains_t *gt);
#endif
