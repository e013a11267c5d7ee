/*
 * mgo.h -- ORACLE (test infrastructure, never shipped, never timed as product).
 *
 * A plain-C CPU restatement of the reference minigraph hot path, one function per stage the HIP
 * kernels replace.  Every function cites the reference file:line it restates.  Parity of this
 * restatement is PINNED against the unmodified reference built by oracle/Makefile into
 * oracle/_ref/libmgref.so (tests/test_oracle_vs_ref.py) and against the committed golden
 * vectors under tests/golden/ (generated from that same library by tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code.
 */
#ifndef MGO_H
#define MGO_H

#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t x, y; } mgo128_t;

/* ---- klib-style in-place radix sort permutation (ksort.h:112-162) ---- */
void mgo_sort128x(mgo128_t *a, int64_t n);          /* key = x, 8 key bytes (misc.c:9-10) */
void mgo_sort64(uint64_t *a, int64_t n);            /* key = value, 8 key bytes (gfa-base.c:13-14) */

/* ---- sketch (sketch.c:28-109) ---- */
uint64_t mgo_hash64(uint64_t key, uint64_t mask);
/* appends to out (capacity cap); returns number of minimizers written, or -(needed) if cap too small */
int64_t mgo_sketch(const char *seq, int32_t len, int32_t w, int32_t k, uint32_t rid, mgo128_t *out, int64_t cap);

/* ---- index: sorted (hash -> positions) table, lookup semantics of index.c:50-72 ---- */
typedef struct {
	int64_t n_keys, n_pos;
	uint64_t *key;   /* sorted distinct minimizer hashes (x>>8 of sketch output) */
	int64_t *off;    /* n_keys+1 offsets into pos */
	uint64_t *pos;   /* per key: y values sorted ascending (index.c:147-158) */
} mgo_idx_t;
mgo_idx_t *mgo_idx_build(int32_t n_seg, const char *const *seq, const int32_t *len, int32_t w, int32_t k);
void mgo_idx_free(mgo_idx_t *idx);
const uint64_t *mgo_idx_get(const mgo_idx_t *idx, uint64_t minier, int32_t *n);

/* ---- seeds (map-algo.c:58-91,152-192) ---- */
/* returns n_a; a[] must hold the total hit count (call with a==NULL to get it); mini_pos holds n_mz */
int64_t mgo_collect_seed_hits(const mgo_idx_t *idx, const int32_t *seg_len, int32_t max_occ,
							  int64_t n_mz, const mgo128_t *mz, mgo128_t *a,
							  int32_t *rep_len, int32_t *n_mini_pos, int32_t *mini_pos);

/* ---- linear chaining (lchain.c:9-219) ---- */
/* a[] (n anchors, x-sorted) is compacted in place to the chained anchors; u[] (cap n) receives
 * score<<32|cnt per chain; returns n_u.  n_a_out = sum of cnt. */
int32_t mgo_lchain_dp(int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter,
					  int32_t min_cnt, int32_t min_sc, float chn_pen_gap, float chn_pen_skip,
					  int64_t n, mgo128_t *a, uint64_t *u, int64_t *n_a_out);

/* ---- miniwfa exact 2-piece affine WFA with traceback (miniwfa.c:11-435,603-615,824-834) ---- */
typedef struct { int32_t x, o1, e1, o2, e2; int64_t max_iter; } mgo_wfa_opt_t;
/* returns score (>=0) or -1 when max_iter is exceeded; cigar (cap ops) gets len<<4|op, *n_cigar set */
int32_t mgo_wfa_exact(const mgo_wfa_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs,
					  uint32_t *cigar, int32_t cap, int32_t *n_cigar, int64_t *n_iter);

#endif
