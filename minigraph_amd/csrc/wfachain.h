/* wfachain.h -- plan + stitch of miniwfa's chained fallback (mwf_wfa_chain, miniwfa.c:776-822); see wfachain.c */
#ifndef MGA_WFACHAIN_H
#define MGA_WFACHAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t x, o1, e1, o2, e2, kmer, max_occ, min_len; } mga_wc_par_t;
/* one element of the plan: a literal CIGAR op (sub == 0) or a sub-problem ts[x0,x0+tl) vs qs[y0,y0+ql) for the exact WFA (sub == 1) */
typedef struct { int32_t sub, op, len, x0, y0, tl, ql; } mga_wc_el_t;
typedef struct { int32_t n, m, n_sub, score; mga_wc_el_t *el; } mga_wc_plan_t; /* score: penalties of the literal ops only */

void mga_wc_par_default(mga_wc_par_t *o);
int mga_wfa_chain_plan(const mga_wc_par_t *o, int32_t tl, const char *ts, int32_t ql, const char *qs, mga_wc_plan_t *p);
void mga_wfa_chain_plan_free(mga_wc_plan_t *p);
/* final CIGAR from the plan and the CIGARs of its sub-problems (in plan order); returns the number of ops or -1 if cap is too small */
int64_t mga_wfa_chain_stitch(const mga_wc_plan_t *p, const uint32_t *const *sub_cig, const int32_t *sub_n, uint32_t *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
