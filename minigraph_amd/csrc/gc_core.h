/*
 * gc_core.h -- graph chaining of ONE read, from the linear chains of k_lchain to the filtered graph chains, as ONE
 * allocation-free routine that runs on a GPU lane and on a host thread from the same source:
 *
 *   chain records (lchain.c:374-408) -> end / seed clean-up (map-algo.c:194-330,424-445) -> anchor rewrite (lchain.c:410-441)
 *   -> DP over the chains with graph reachability (gchain1.c:62-240 + shortk.c:41-242)
 *   -> walk assembly with GWFA / shortest-walk bridging (gchain1.c:242-520 + gfa-ed.c:44-617)
 *   -> ordering, primary / secondary assignment, filtering (gcmisc.c:6-188)
 *
 * Design (this file is NOT a transcription of those files; it restates what they compute):
 *   * every container lives in a bump ARENA owned by the executing lane (HBM scratch on the device, malloc'ed blocks on the
 *     host): vectors grow by re-allocation at the arena top, a whole call is released by resetting one offset, and running out
 *     of arena is a status the caller handles by re-running the read in a larger arena -- no malloc / free / kalloc anywhere;
 *   * the stages talk through index-based records (gc_chain_t, gc_frag_t, gc_cand_t, gc_walk_t ...) instead of pointers into
 *     each other's arrays, so that the same bytes are valid on both sides of PCIe;
 *   * everything that decides a tie in the reference is reproduced exactly: the klib radix-sort permutation (gc_ksort), the
 *     15-slot max-heap sifts of the walk lists, the settle order of the shortest-walk search, the append / merge / dedup order
 *     of the GWFA wavefronts;
 *   * arithmetic is integer or single IEEE operations in float / double (compiled with -ffp-contract=off); the two libm calls
 *     of the post-processing (log in div, logf in MAPQ) stay on the host, which gets their integer inputs.
 *
 * On the device the routine is executed by lane 0 of a wavefront per read (k_gchain.hip); the host instantiation serves
 * -x asm (where the chainer is host code anyway), the last capacity tier, and the CPU parity tests.
 */
#ifndef MGA_GC_CORE_H
#define MGA_GC_CORE_H

#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/minigraph_amd.h"

#if defined(__HIPCC__)
#define GC_HD __host__ __device__ inline
#ifdef GC_AB_NOINLINE   /* (A/B builds, tools/gchain_ab.py: the big routines as functions of their own -- half the code, [measured] 6-13 % slower) */
#define GC_HDN __host__ __device__ __attribute__((noinline)) inline
#else
#define GC_HDN __host__ __device__ inline
#endif
#else
#define GC_HD static inline
#define GC_HDN static inline
#endif

/* A routine's bookkeeping state (vector headers, counters, search state).  Every lane of the wavefront holds the SAME values in it (replicated execution, below), so on the
 * device ONE copy per wavefront in LDS serves all 64 lanes: a private copy per lane lives in scratch memory -- 64 times the footprint, a trip to L2 / HBM per access -- once
 * its address is taken, which is how the routines here hand state to each other.  (k_gchain runs one wavefront per workgroup; the routines using it are not re-entered.) */
/* Fields of such state sit on 16-byte boundaries (GC_F): the routines reach the state through plain pointers, the compiler merges neighbouring fields into one access, and a
 * 16-byte access through a generic pointer into LDS must be 16-byte aligned (8 is enough in HBM: the aperture fault of round 2).  With every field on its own 16-byte slot
 * there are no neighbours to merge, and whole-struct copies / clears start aligned. */
#define GC_F __attribute__((aligned(16)))
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GC_AB_NO_LDS_STATE)
#define GC_STATE(T, name) __shared__ T name##_lds_; T &name = name##_lds_
#else
#define GC_STATE(T, name) T name
#endif

#ifndef GC_PARSORT
#define GC_PARSORT 7   /* which parts of gc_diag_sort are split over the lanes (1 split, 2 sort fill, 4 merge): a debugging aid */
#endif
#define GC_OK        0
#define GC_E_ARENA   1   /* ran out of arena: re-run in a larger one */
#define GC_E_BUG     2   /* the "logical bug" exit of the shortest-walk search (shortk.c:182-186): the read gets no chains, as in the reference */

/* ------------------------------------------------------------------------------------------------ arena */

typedef struct gc_block_s { struct gc_block_s *prev; int64_t cap; } gc_block_t;
typedef struct gc_arena_s {
	GC_F char *base;          /* current block */
	GC_F int64_t top; GC_F int64_t cap;
	GC_F int32_t ovf;         /* set once an allocation failed (device: fixed capacity) */
	GC_F int32_t growable;    /* host: chain further malloc'ed blocks instead of failing */
	GC_F gc_block_t *blocks;  /* host: extra blocks, newest first */
	GC_F int64_t peak;
	GC_F unsigned long long *ticks; GC_F long long tick_last; /* profiling (device): cycles between consecutive GC_TICKs, summed per stage; NULL = off */
	GC_F char *fast_base; GC_F int64_t fast_cap; /* device: a small block of LDS for the scratch of ONE graph search / GWFA call at a time (a tenth of the latency
	                          * of HBM); a call that outgrows it is simply run again in the main arena.  NULL: everything in the main arena */
} gc_arena_t;

#if defined(__HIP_DEVICE_COMPILE__)
#define GC_TICK(A, id) do { if ((A)->ticks && (threadIdx.x & 63) == 0) { const long long now_ = (long long)clock64(); atomicAdd(&(A)->ticks[id], (unsigned long long)(now_ - (A)->tick_last)); (A)->tick_last = now_; } } while (0)
#define GC_COUNT(A, id, v) do { if ((A)->ticks && (threadIdx.x & 63) == 0) atomicAdd(&(A)->ticks[id], (unsigned long long)(v)); } while (0)
#elif defined(GC_HOST_PROF) /* (host profiling aid: cycles between ticks, summed per stage into gc_host_ticks[]) */
#include <x86intrin.h>
static unsigned long long gc_host_ticks[16];
static __thread unsigned long long gc_host_last;
#define GC_TICK(A, id) do { const unsigned long long now_ = __rdtsc(); if ((id) != 0) __sync_fetch_and_add(&gc_host_ticks[id], now_ - gc_host_last); gc_host_last = now_; } while (0)
#else
#define GC_TICK(A, id) ((void)0)
#endif
#ifdef GC_DD_PROF
#define GC_DDP(...) __VA_ARGS__   /* (profiling build: the phases of the device's dedup in count slots 1-4) */
#else
#define GC_DDP(...)
#endif
#ifndef GC_COUNT
#define GC_COUNT(A, id, v) ((void)0)   /* (device profiling of ONE bridge, k_gchain_p2 with MGA_GC_SPLIT_DEBUG=2: steps, cells, runs, head cells, output cells, sorts in ticks[1,2,3,4,6,10]) */
#endif

GC_HD void gc_arena_init(gc_arena_t *A, void *mem, int64_t cap, int growable) { A->base = (char*)mem, A->top = 0, A->cap = cap, A->ovf = 0, A->growable = growable, A->blocks = 0, A->peak = 0, A->ticks = 0, A->tick_last = 0, A->fast_base = 0, A->fast_cap = 0; }

GC_HD void *gc_alloc(gc_arena_t *A, int64_t bytes)
{
	const int64_t need = (bytes + 15) & ~(int64_t)15;
	if (A->top + need > A->cap) {
#if !defined(__HIP_DEVICE_COMPILE__)
		if (A->growable) { /* host: a new block, at least twice the request; the old block stays alive until gc_arena_free() */
			int64_t cap = A->cap * 2 > need * 2 ? A->cap * 2 : need * 2;
			gc_block_t *b = (gc_block_t*)malloc((size_t)cap + sizeof(gc_block_t));
			if (b) {
				b->prev = A->blocks, b->cap = cap, A->blocks = b;
				A->base = (char*)(b + 1), A->top = 0, A->cap = cap;
			} else { A->ovf = 1; return 0; }
		} else
#endif
		{ A->ovf = 1; return 0; }
	}
	void *p = A->base + A->top;
	A->top += need;
	if (A->top > A->peak) A->peak = A->top;
	return p;
}
static inline void gc_arena_free_blocks(gc_arena_t *A) { while (A->blocks) { gc_block_t *b = A->blocks; A->blocks = b->prev; free(b); } } /* host */

/* overlapping move in 4-byte units (every record moved here is a multiple of 4 bytes) */
GC_HD void gc_move(void *dst, const void *src, int64_t bytes)
{
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	const int64_t n = bytes >> 2;
	if (d == s || n <= 0) return;
	if (d < s) for (int64_t i = 0; i < n; ++i) d[i] = s[i];
	else for (int64_t i = n - 1; i >= 0; --i) d[i] = s[i];
}

/* ------------------------------------------------------------------------------------------------ execution model
 * On the device the routine runs REPLICATED on the 64 lanes of a wavefront: every lane executes the same control flow on the same
 * values (scalar state lives in identical private copies, memory is shared, stores of identical values to one address are benign),
 * so the hot per-element loops can be handed to the lanes -- 64 independent loads in flight instead of one dependent chain --
 * with a wavefront-scope fence between a loop and the code that reads what it wrote.  Order-preserving appends use ballots.
 * On the host the same code runs with one lane. */
#if defined(__HIP_DEVICE_COMPILE__)
#define GC_LANE ((int32_t)(threadIdx.x & 63))
#define GC_NLANE 64
GC_HD void gc_sync(void) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
GC_HD uint64_t gc_ballot(int pred) { return __ballot(pred); }
GC_HD int32_t gc_rank(uint64_t mask) { return (int32_t)__popcll(mask & ((1ULL << (threadIdx.x & 63)) - 1ULL)); } /* set bits below this lane */
GC_HD int32_t gc_popc(uint64_t mask) { return (int32_t)__popcll(mask); }
GC_HD int32_t gc_sum(int32_t v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; } /* sum over the lanes, the same in every lane */
#else
#define GC_LANE 0
#define GC_NLANE 1
GC_HD void gc_sync(void) {}
GC_HD uint64_t gc_ballot(int pred) { return pred ? 1ULL : 0ULL; }
GC_HD int32_t gc_rank(uint64_t mask) { (void)mask; return 0; }
GC_HD int32_t gc_popc(uint64_t mask) { return (int32_t)(mask & 1ULL); }
GC_HD int32_t gc_sum(int32_t v) { return v; }
#endif
#define GC_PAR_FOR(i, n) for (int32_t i = GC_LANE; i < (n); i += GC_NLANE)   /* no allocation, no gc_sync() inside */

/* non-overlapping copy in 4-byte units by all lanes */
GC_HD void gc_pcopy(void *dst, const void *src, int64_t bytes)
{
	if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes) & 15) == 0) { /* anchors, cells, intervals: 16-byte records at 16-byte addresses move as such -- a quarter of the
		                                                                      * instructions, and no later 8-byte update of a record by ANOTHER lane meets halves written by two lanes */
		typedef struct __attribute__((aligned(16))) { uint64_t a, b; } gc_w16_t;
		gc_w16_t *d = (gc_w16_t*)dst;
		const gc_w16_t *s = (const gc_w16_t*)src;
		for (int64_t i = GC_LANE, n = bytes >> 4; i < n; i += GC_NLANE) d[i] = s[i];
		gc_sync();
		return;
	}
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	for (int64_t i = GC_LANE, n = bytes >> 2; i < n; i += GC_NLANE) d[i] = s[i];
	gc_sync();
}
/* move to a LOWER address (dst <= src), regions may overlap: block by block, every block loaded by all lanes before it is stored */
GC_HD void gc_pmove_down(void *dst, const void *src, int64_t bytes)
{
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	const int64_t n = bytes >> 2;
	if (d == s) return;
	for (int64_t b = 0; b < n; b += GC_NLANE) {
		const int64_t i = b + GC_LANE;
		const uint32_t v = i < n ? s[i] : 0;
		gc_sync();
		if (i < n) d[i] = v;
		gc_sync();
	}
}

/* move to a HIGHER address (dst >= src), regions may overlap: blocks from the top down */
GC_HD void gc_pmove_up(void *dst, const void *src, int64_t bytes)
{
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	const int64_t n = bytes >> 2;
	if (d == s) return;
	for (int64_t e = n; e > 0; e -= GC_NLANE) {
		const int64_t i = e - 1 - GC_LANE;
		const uint32_t v = i >= 0 ? s[i] : 0;
		gc_sync();
		if (i >= 0) d[i] = v;
		gc_sync();
	}
}

/* growable array in the arena: {a, n, m}; growth re-allocates at the top (in place when the array is the last allocation) */
#define GC_VEC(T) struct { GC_F T *a; GC_F int32_t n; GC_F int32_t m; }
#define gc_vec_zero(v) ((v).a = 0, (v).n = (v).m = 0)
GC_HDN int gc_vec_grow_(gc_arena_t *A, void **pa, int32_t *pm, int32_t n_used, int32_t need, int32_t esz)
{
	if (need <= *pm) return GC_OK;
	int32_t m = *pm < 8 ? 8 : *pm + (*pm >> 1);
	if (m < need) m = need;
	char *old = (char*)*pa;
	if (old && old + (((int64_t)*pm * esz + 15) & ~(int64_t)15) == A->base + A->top && A->top + (((int64_t)m * esz + 15) & ~(int64_t)15) - (((int64_t)*pm * esz + 15) & ~(int64_t)15) <= A->cap) {
		A->top += (((int64_t)m * esz + 15) & ~(int64_t)15) - (((int64_t)*pm * esz + 15) & ~(int64_t)15); /* last allocation: extend in place */
		if (A->top > A->peak) A->peak = A->top;
		*pm = m;
		return GC_OK;
	}
	char *p = (char*)gc_alloc(A, (int64_t)m * esz);
	if (p == 0) return GC_E_ARENA;
	if (old && n_used > 0) gc_pcopy(p, old, (((int64_t)n_used * esz) + 3) & ~(int64_t)3); /* (allocations are 16-byte multiples: rounding up stays inside) */
	*pa = p, *pm = m;
	return GC_OK;
}
#define gc_vec_reserve(A, v, need) ((need) <= (v).m ? GC_OK : gc_vec_grow_((A), (void**)&(v).a, &(v).m, (v).n, (need), (int32_t)sizeof(*(v).a)))
#define GC_TRY(expr) do { int rc_ = (expr); if (rc_ != GC_OK) return rc_; } while (0)
#define GC_PUSH(A, v, ptr) do { GC_TRY(gc_vec_reserve((A), (v), (v).n + 1)); (ptr) = &(v).a[(v).n++]; } while (0)
#define GC_ALLOC(A, T, ptr, count) do { (ptr) = (T*)gc_alloc((A), (int64_t)(count) * (int64_t)sizeof(T)); if ((ptr) == 0) return GC_E_ARENA; } while (0)

/* ------------------------------------------------------------------------------------------------ small helpers */

GC_HD uint32_t gc_hash32(uint32_t key) /* kh_hash_uint32, khashl.h:321-327 */
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
	key ^= (key >> 6);   key += ~(key << 11); key ^= (key >> 16);
	return key;
}

GC_HD float gc_log2f(float x) /* mg_log2, mgpriv.h:63-71 (x >= 2) */
{
	union { float f; uint32_t i; } z;
	z.f = x;
	float r = (float)((int32_t)(z.i >> 23 & 255) - 128);
	z.i &= ~(255U << 23);
	z.i += 127U << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

/* anchor field accessors (minigraph.h:41, mgpriv.h:18-27) */
#define GC_AX(p) ((int32_t)(p).x)                       /* target end position */
#define GC_AY(p) ((int32_t)(p).y)                       /* query end position */
#define GC_ASPAN(p) ((int32_t)((p).y >> 32 & 0xff))
#define GC_ASEG(p) ((int32_t)(((p).y & MG_SEED_SEG_MASK) >> MG_SEED_SEG_SHIFT))

/* ---- exact permutation of klib's in-place MSD byte radix sort (ksort.h:112-162) on {key, val} records.  The sort is unstable
 * for n > 64 and results depend on where ties land, so the algorithm's moves are replayed: counting pass, displacement cycles from
 * the lowest bucket up, buckets of <= 64 records by (stable) insertion sort, larger ones recursively on the next byte.  Ranges
 * wait on an explicit stack in the arena; ranges are disjoint, so their processing order does not change the result. ---- */
typedef struct { uint64_t key, val; } gc_kv_t;

GC_HD void gc_isort(gc_kv_t *a, int32_t n)
{
	for (int32_t i = 1; i < n; ++i) {
		gc_kv_t t = a[i];
		int32_t j = i;
		for (; j > 0 && t.key < a[j - 1].key; --j) a[j] = a[j - 1];
		a[j] = t;
	}
}

GC_HDN int gc_ksort(gc_arena_t *A, gc_kv_t *a, int32_t n, int key_bytes)
{
	if (n <= 64) { gc_isort(a, n); return GC_OK; }
	const int64_t mark = A->top;
	typedef struct { int32_t b, e, sh; } rng_t;
	GC_VEC(rng_t) stk;
	gc_vec_zero(stk);
	int32_t *head, *tail, *cnt;
	GC_ALLOC(A, int32_t, head, 768);
	tail = head + 256, cnt = head + 512;
	rng_t *r0;
	GC_PUSH(A, stk, r0);
	r0->b = 0, r0->e = n, r0->sh = (key_bytes - 1) * 8;
	while (stk.n > 0) {
		const rng_t r = stk.a[--stk.n];
		for (int k = 0; k < 256; ++k) cnt[k] = 0;
		for (int32_t i = r.b; i < r.e; ++i) ++cnt[a[i].key >> r.sh & 0xff];
		for (int32_t k = 0, pos = r.b; k < 256; ++k) head[k] = pos, pos += cnt[k], tail[k] = pos;
		for (int k = 0; k < 256; ++k)
			while (head[k] != tail[k]) {
				gc_kv_t carry = a[head[k]];
				int l = (int)(carry.key >> r.sh & 0xff);
				if (l == k) { ++head[k]; continue; }
				do {
					gc_kv_t t = a[head[l]];
					a[head[l]++] = carry;
					carry = t;
					l = (int)(carry.key >> r.sh & 0xff);
				} while (l != k);
				a[head[k]++] = carry;
			}
		if (r.sh > 0) {
			const int nsh = r.sh > 8 ? r.sh - 8 : 0;
			for (int k = 0; k < 256; ++k) {
				const int32_t st = tail[k] - cnt[k];
				if (cnt[k] > 64) { rng_t *q; GC_PUSH(A, stk, q); q->b = st, q->e = tail[k], q->sh = nsh; }
				else if (cnt[k] > 1) gc_isort(a + st, cnt[k]);
			}
		}
	}
	A->top = mark;
	return GC_OK;
}

/* plain ascending sort of 64-bit values whose order among equals cannot matter (whole value is the key) */
GC_HDN void gc_sort_u64(uint64_t *a, int32_t n)
{
	for (int32_t i = n / 2 - 1; i >= 0; --i) { /* heap sort: O(n log n) without scratch */
		int32_t k = i; uint64_t t = a[k];
		for (;;) { int32_t c = 2 * k + 1; if (c >= n) break; if (c + 1 < n && a[c + 1] > a[c]) ++c; if (a[c] <= t) break; a[k] = a[c], k = c; }
		a[k] = t;
	}
	for (int32_t m = n - 1; m > 0; --m) {
		uint64_t t = a[m]; a[m] = a[0];
		int32_t k = 0;
		for (;;) { int32_t c = 2 * k + 1; if (c >= m) break; if (c + 1 < m && a[c + 1] > a[c]) ++c; if (a[c] <= t) break; a[k] = a[c], k = c; }
		a[k] = t;
	}
}

/* ------------------------------------------------------------------------------------------------ graph view */

typedef struct { uint64_t v_lv; uint32_t w; int32_t rank; int32_t ov, ow; uint64_t link_bits; } gc_arc_t; /* byte layout of gfa_arc_t (gfa.h:33-39) */

typedef struct {
	GC_F const gc_arc_t *arc;
	GC_F const uint64_t *idx;          /* per vertex: first arc << 32 | number of arcs (gfa.h:99-101) */
	GC_F const int32_t *seg_len;
	/* oriented vertex sequences: either one pointer per vertex (host: gfa_edseq_t[]) or two flat copies (device) */
	GC_F const gfa_edseq_t *es;
	GC_F const char *seq_fw; GC_F const char *seq_rc;
	GC_F const int64_t *seq_off;
} gc_graph_t;

GC_HD const char *gc_vseq(const gc_graph_t *G, uint32_t v) { return G->es ? G->es[v].seq : ((v & 1) ? G->seq_rc : G->seq_fw) + G->seq_off[v >> 1]; }
GC_HD int32_t gc_vlen(const gc_graph_t *G, uint32_t v) { return G->seg_len[v >> 1]; }
GC_HD int32_t gc_n_arc(const gc_graph_t *G, uint32_t v) { return (int32_t)(uint32_t)G->idx[v]; }
GC_HD const gc_arc_t *gc_arcs(const gc_graph_t *G, uint32_t v) { return G->arc + (G->idx[v] >> 32); }

/* ------------------------------------------------------------------------------------------------ parameters, records */

typedef struct {
	GC_F int32_t k;                                  /* minimizer length of the index */
	GC_F int32_t bw; GC_F int32_t bw_long; GC_F int32_t max_gap;               /* mg_mapopt_t */
	GC_F int32_t min_lc_cnt; GC_F int32_t lc_max_occ; GC_F int32_t lc_max_trim;
	GC_F int32_t max_gc_skip; GC_F int32_t ref_bonus; GC_F int32_t min_gc_cnt; GC_F int32_t min_gc_score; GC_F int32_t gdp_max_ed;
	GC_F int32_t best_n; GC_F int32_t sub_diff;
	GC_F float chn_pen_gap;                          /* already scaled by exp(-div * k) (map-algo.c:388-390) */
	GC_F float mask_level; GC_F float pri_ratio;
} gc_par_t;

typedef struct { /* one linear chain (what mg_lchain_t carries, mgpriv/minigraph.h:100-106) */
	int32_t off, cnt;          /* anchors */
	uint32_t v;
	int32_t rs, re, qs, qe, score;
	int32_t dist_pre; uint32_t hash_pre; int32_t inner_pre;   /* link to the predecessor chosen by the DP */
} gc_chain_t;

typedef struct { /* one graph chain, integer fields only; div and mapq are derived on the host (libm) */
	int32_t off, cnt, n_anchor, score;
	int32_t qs, qe, plen, ps, pe, blen, mlen;
	int32_t n_mini, q_span;    /* inputs of div (gchain1.c:299) */
	int32_t id, parent, subsc, n_sub, flt;
	uint32_t hash;
} gc_rec_t;

typedef struct {
	GC_F int32_t n_gc; GC_F int32_t n_lc; GC_F int32_t n_a;
	GC_F gc_rec_t *gc;              /* arena */
	GC_F mg_llchain_t *lc;          /* arena */
	GC_F mg128_t *a;                /* caller's buffer (capacity: the read's chained anchors) */
	GC_F int32_t n_gwfa; GC_F int32_t n_shortk; GC_F int32_t n_fast;  /* counters (n_fast: GWFA calls + graph searches that ran in the LDS scratch) */
} gc_result_t;

/* ------------------------------------------------------------------------------------------------ chain records + clean-up */

/* chain records in the order of the reference: by query start, then score, through the klib sort on qs<<32|score (lchain.c:374-408) */
GC_HD int gc_make_chains(gc_arena_t *A, int32_t n_u, const uint64_t *u, const mg128_t *a, gc_chain_t **out)
{
	gc_chain_t *c;
	gc_kv_t *z;
	*out = 0;
	if (n_u <= 0) return GC_OK;
	GC_ALLOC(A, gc_chain_t, c, n_u);
	const int64_t mark = A->top;
	GC_ALLOC(A, gc_kv_t, z, n_u);
	for (int32_t i = 0, k = 0; i < n_u; ++i) {
		const int32_t qs = GC_AY(a[k]) + 1 - GC_ASPAN(a[k]);
		z[i].key = (uint64_t)qs << 32 | u[i] >> 32;
		z[i].val = (uint64_t)k << 32 | (uint32_t)u[i];
		k += (int32_t)u[i];
	}
	GC_TRY(gc_ksort(A, z, n_u, 8));
	for (int32_t i = 0; i < n_u; ++i) {
		gc_chain_t *r = &c[i];
		const int32_t k = (int32_t)(z[i].val >> 32), span = GC_ASPAN(a[k]);
		r->off = k, r->cnt = (int32_t)(uint32_t)z[i].val, r->score = (int32_t)(uint32_t)z[i].key;
		r->v = (uint32_t)(a[k].x >> 32);
		r->rs = GC_AX(a[k]) + 1 > span ? GC_AX(a[k]) + 1 - span : 0;
		r->qs = (int32_t)(z[i].key >> 32);
		r->re = GC_AX(a[k + r->cnt - 1]) + 1, r->qe = GC_AY(a[k + r->cnt - 1]) + 1;
		r->dist_pre = -1, r->hash_pre = 0, r->inner_pre = 0;
	}
	A->top = mark;
	*out = c;
	return GC_OK;
}

/* The clean-up of a chain works on a window [s, s+n) of its anchors.  Four passes (map-algo.c:194-330):
 *   1. ends made of high-occurrence seeds are trimmed (at most max_trim per side);
 *   2. ends that sit behind a gap larger than half of what has been matched so far are trimmed;
 *   3. seeds between an insertion-like and a deletion-like long gap that nearly cancel are flagged IGNORE;
 *   4. seeds in runs of long gaps that follow each other closely are flagged IGNORE, the last one FIXED. */
GC_HD int32_t gc_gap_at(const mg128_t *a, int32_t i) { return (GC_AY(a[i]) - GC_AY(a[i - 1])) - (GC_AX(a[i]) - GC_AX(a[i - 1])); }

GC_HD void gc_trim_repetitive_ends(const mg128_t *a, int32_t max_occ, int32_t max_trim, int32_t *s, int32_t *n)
{
	int32_t cut = 0;
	while (cut < max_trim && cut < *n && (int32_t)(a[*s + *n - 1 - cut].y >> MG_SEED_OCC_SHIFT) > max_occ) ++cut;
	*n -= cut;
	cut = 0;
	while (cut < *n && cut < max_trim && (int32_t)(a[*s + cut].y >> MG_SEED_OCC_SHIFT) > max_occ) ++cut;
	*s += cut, *n -= cut;
}

/* one direction of pass 2: walk inwards from an end, remember the innermost anchor that follows a disproportionate gap */
GC_HD void gc_trim_gapped_ends(const mg128_t *a, int32_t score, int32_t bw, int32_t min_match, int32_t *s, int32_t *n)
{
	const int32_t s0 = *s, e0 = *s + *n; /* [s0, e0) */
	if (*n < 3) return;
	int32_t len = GC_ASPAN(a[s0]), mat = len;
	for (int32_t i = s0 + 1; i < e0 - 1; ++i) {
		const int32_t dr = GC_AX(a[i]) - GC_AX(a[i - 1]), dq = GC_AY(a[i]) - GC_AY(a[i - 1]);
		const int32_t lo = dr < dq ? dr : dq, hi = dr > dq ? dr : dq, span = GC_ASPAN(a[i]);
		if (hi - lo > len >> 1) *s = i;
		len += lo, mat += lo < span ? lo : span;
		if (len >= bw << 1 || (mat >= min_match && mat >= bw) || mat >= score >> 1) break;
	}
	*n = e0 - *s;
	len = mat = GC_ASPAN(a[e0 - 1]);
	for (int32_t i = e0 - 2; i > *s; --i) {
		const int32_t dr = GC_AX(a[i + 1]) - GC_AX(a[i]), dq = GC_AY(a[i + 1]) - GC_AY(a[i]);
		const int32_t lo = dr < dq ? dr : dq, hi = dr > dq ? dr : dq, span = GC_ASPAN(a[i + 1]);
		if (hi - lo > len >> 1) *n = i + 1 - *s;
		len += lo, mat += lo < span ? lo : span;
		if (len >= bw << 1 || (mat >= min_match && mat >= bw) || mat >= score >> 1) break;
	}
}

/* positions (relative to s) of the gaps longer than min_gap; returns the count, 0 when there are fewer than two */
GC_HD int gc_long_gaps(gc_arena_t *A, const mg128_t *a, int32_t s, int32_t n, int32_t min_gap, int32_t **pos, int32_t *n_pos)
{
	int32_t m = 0, *K;
	*pos = 0, *n_pos = 0;
	for (int32_t base = 1; base < n; base += GC_NLANE) { /* (lanes: one anchor each; the count is the same in every lane) */
		const int32_t i = base + GC_LANE;
		int big = 0;
		if (i < n) { const int32_t g = gc_gap_at(a, s + i); big = g < -min_gap || g > min_gap; }
		m += gc_popc(gc_ballot(big));
	}
	if (m <= 1) return GC_OK;
	GC_ALLOC(A, int32_t, K, m);
	m = 0;
	for (int32_t base = 1; base < n; base += GC_NLANE) {
		const int32_t i = base + GC_LANE;
		int big = 0;
		if (i < n) { const int32_t g = gc_gap_at(a, s + i); big = g < -min_gap || g > min_gap; }
		const uint64_t mk = gc_ballot(big);
		if (big) K[m + gc_rank(mk)] = i;
		m += gc_popc(mk);
	}
	gc_sync();
	*pos = K, *n_pos = m;
	return GC_OK;
}

GC_HD int gc_flag_cancelling_gaps(gc_arena_t *A, mg128_t *a, int32_t s, int32_t n, int32_t min_gap, int32_t diff_thres, int32_t max_ext_len, int32_t max_ext_cnt)
{
	const int64_t mark = A->top;
	int32_t *K, m;
	GC_TRY(gc_long_gaps(A, a, s, n, min_gap, &K, &m));
	if (K == 0) return GC_OK;
	int32_t best = 0, best_st = -1, best_en = -1;
	for (int32_t k = 0;; ++k) {
		if (k == m || k >= best_en) { /* the best window found so far ends here: flag what lies inside it */
			if (best_en > 0) for (int32_t i = K[best_st]; i < K[best_en]; ++i) a[s + i].y |= MG_SEED_IGNORE;
			best = 0, best_st = best_en = -1;
			if (k == m) break;
		}
		int32_t ins = 0, del = 0, top_diff = 0, top_l = -1;
		int32_t g = gc_gap_at(a, s + K[k]);
		if (g > 0) ins += g; else del -= g;
		const int32_t q0 = GC_AY(a[s + K[k] - 1]), r0 = GC_AX(a[s + K[k] - 1]);
		for (int32_t l = k + 1; l < m && l <= k + max_ext_cnt; ++l) {
			const int32_t j = K[l];
			if (GC_AY(a[s + j]) - q0 > max_ext_len || GC_AX(a[s + j]) - r0 > max_ext_len) break;
			g = gc_gap_at(a, s + j);
			if (g > 0) ins += g; else del -= g;
			const int32_t diff = ins + del - (ins > del ? ins - del : del - ins);
			if (top_diff < diff) top_diff = diff, top_l = l;
		}
		if (top_diff > diff_thres && top_diff > best) best = top_diff, best_st = k, best_en = top_l;
	}
	A->top = mark;
	return GC_OK;
}

GC_HD int gc_flag_gap_runs(gc_arena_t *A, mg128_t *a, int32_t s, int32_t n, int32_t min_gap, int32_t max_ext)
{
	const int64_t mark = A->top;
	int32_t *K, m;
	GC_TRY(gc_long_gaps(A, a, s, n, min_gap, &K, &m));
	if (K == 0) return GC_OK;
	for (int32_t k = 0; k < m;) {
		int32_t g1 = gc_gap_at(a, s + K[k]), l;
		int32_t re = GC_AX(a[s + K[k]]), qe = GC_AY(a[s + K[k]]);
		if (g1 < 0) g1 = -g1;
		for (l = k + 1; l < m; ++l) {
			const int32_t j = K[l];
			if (GC_AY(a[s + j]) - qe > max_ext || GC_AX(a[s + j]) - re > max_ext) break;
			int32_t g2 = gc_gap_at(a, s + j);
			const int32_t span = GC_ASPAN(a[s + j - 1]);
			const int32_t r2 = GC_AX(a[s + j - 1]) + span, q2 = GC_AY(a[s + j - 1]) + span;
			const int32_t between = r2 - re < q2 - qe ? r2 - re : q2 - qe;
			if (g2 < 0) g2 = -g2;
			if (between > g1 + g2) break;
			re = GC_AX(a[s + j]), qe = GC_AY(a[s + j]), g1 = g2;
		}
		if (l > k + 1) {
			const int32_t end = K[l - 1];
			for (int32_t j = K[k]; j < end; ++j) a[s + j].y |= MG_SEED_IGNORE;
			a[s + end].y |= MG_SEED_FIXED;
		}
		k = l;
	}
	A->top = mark;
	return GC_OK;
}

/* only reads with several chains are cleaned (map-algo.c:424); chains left with fewer than min_lc_cnt anchors are dropped */
GC_HD int gc_clean_chains(gc_arena_t *A, const gc_par_t *P, mg128_t *a, gc_chain_t *c, int32_t *n_c)
{
	int32_t kept = 0;
	for (int32_t i = 0; i < *n_c; ++i) {
		gc_chain_t r = c[i];
		int32_t s = r.off, n = r.cnt;
		gc_trim_repetitive_ends(a, P->lc_max_occ, P->lc_max_trim, &s, &n);
		gc_trim_gapped_ends(a, r.score, P->bw, 100, &s, &n);
		GC_TRY(gc_flag_cancelling_gaps(A, a, s, n, 10, 40, P->max_gap >> 1, 10));
		GC_TRY(gc_flag_gap_runs(A, a, s, n, 30, P->max_gap >> 1));
		if (n < P->min_lc_cnt) continue;
		const int32_t span = GC_ASPAN(a[s]);
		r.off = s, r.cnt = n;
		r.rs = GC_AX(a[s]) + 1 - span, r.qs = GC_AY(a[s]) + 1 - span;
		r.re = GC_AX(a[s + n - 1]) + 1, r.qe = GC_AY(a[s + n - 1]) + 1;
		c[kept++] = r;
	}
	*n_c = kept;
	return GC_OK;
}

/* anchor.x high word := rank of the anchor's minimizer among the read's kept minimizers (lchain.c:410-441).  The reference walks the two
 * ascending lists side by side; query positions are strictly ascending in both, so a binary search per anchor finds the same rank. */
GC_HD int gc_index_anchors(mg128_t *a, int32_t n_a, const int32_t *mini_pos, int32_t n_mini)
{
	int bad = 0;
#if !defined(__HIP_DEVICE_COMPILE__)
	/* one lane: the walk the reference takes (lchain.c:431-441) -- a chain's anchors ascend on the query, so behind the first one's minimizer the two lists are walked side by
	 * side.  A chain the walk does not get through is left to the searches below. */
	if (n_a > 0) {
		int32_t lo = 0, hi = n_mini - 1, st = -1;
		const int32_t x0 = GC_AY(a[0]);
		while (lo <= hi) { const int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1), y = mini_pos[mid]; if (y < x0) lo = mid + 1; else if (y > x0) hi = mid - 1; else { st = mid; break; } }
		if (st >= 0) {
			/* Ranks are written as the walk goes.  When the walk does not get through, the state the searches below leave is the one they would have left alone: the walk
			 * writes a[k].x only where y == mini_pos[j], mini_pos ascends strictly, so the search of that anchor finds the same j and writes the same value again; an anchor the
			 * searches do not find (the GC_E_BUG case) was never written by the walk either. */
			int32_t k = 0, j = st;
			while (k < n_a && j < n_mini) {
				const int32_t y = GC_AY(a[k]), m = mini_pos[j];
				if (y == m) a[k].x = (uint64_t)j << 32 | (a[k].x & 0xffffffffU), ++k, ++j;
				else if (y > m) ++j;
				else break; /* not ascending, or not a minimizer position: the searches decide */
			}
			if (k == n_a) return GC_OK;
		}
	}
#endif
	GC_PAR_FOR(k, n_a) {
		const int32_t x = GC_AY(a[k]);
		int32_t lo = 0, hi = n_mini - 1, at = -1;
		while (lo <= hi) {
			const int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1), y = mini_pos[mid];
			if (y < x) lo = mid + 1; else if (y > x) hi = mid - 1; else { at = mid; break; }
		}
		if (at < 0) bad = 1;
		else a[k].x = (uint64_t)at << 32 | (a[k].x & 0xffffffffU);
	}
	gc_sync();
	return gc_ballot(bad) ? GC_E_BUG : GC_OK;
}

/* ------------------------------------------------------------------------------------------------ shortest walks */

#define GC_SK_EXT 1000 /* MG_SHORT_K_EXT, shortk.c:31 */

typedef struct { /* a destination of the search: in/out (what mg_path_dst_t carries, mgpriv.h:40-52) */
	uint32_t v;
	int32_t target_dist;
	uint32_t target_hash;
	int32_t meta, check_hash, inner;
	int32_t n_path, is_0, path_end, dist;
	uint32_t hash;
} gc_dst_t;

typedef struct { uint32_t v, d; int32_t pre; } gc_walkv_t;

typedef struct { uint64_t di; uint32_t v; int32_t pre; uint32_t hash; int32_t is_0, hpos; } gc_sknode_t; /* di = dist<<32 | serial (settle rank once settled) */
typedef struct { int32_t k; int32_t p[MG_MAX_SHORT_K]; } gc_sklist_t;                                   /* walks ending at one vertex, a max-heap on di */

typedef struct {
	GC_VEC(gc_sknode_t) nd;
	GC_VEC(int32_t) heap;
	GC_F uint32_t *hk; GC_F int32_t *hv; GC_F uint32_t hcap; GC_F uint32_t hcnt;   /* vertex -> index into tk, open addressing */
	GC_VEC(gc_sklist_t) tk;
} gc_sk_t;

GC_HD void gc_fh_up(gc_sk_t *s, int32_t i)
{
	const int32_t x = s->heap.a[i];
	while (i > 0) {
		const int32_t par = (i - 1) >> 1;
		if (s->nd.a[s->heap.a[par]].di <= s->nd.a[x].di) break;
		s->heap.a[i] = s->heap.a[par], s->nd.a[s->heap.a[i]].hpos = i, i = par;
	}
	s->heap.a[i] = x, s->nd.a[x].hpos = i;
}
GC_HD void gc_fh_down(gc_sk_t *s, int32_t i)
{
	const int32_t x = s->heap.a[i], n = s->heap.n;
	for (;;) {
		int32_t c = 2 * i + 1;
		if (c >= n) break;
		if (c + 1 < n && s->nd.a[s->heap.a[c + 1]].di < s->nd.a[s->heap.a[c]].di) ++c;
		if (s->nd.a[s->heap.a[c]].di >= s->nd.a[x].di) break;
		s->heap.a[i] = s->heap.a[c], s->nd.a[s->heap.a[i]].hpos = i, i = c;
	}
	s->heap.a[i] = x, s->nd.a[x].hpos = i;
}
GC_HD int gc_fh_push(gc_arena_t *A, gc_sk_t *s, int32_t x)
{
	GC_TRY(gc_vec_reserve(A, s->heap, s->heap.n + 1));
	s->heap.a[s->heap.n] = x;
	gc_fh_up(s, s->heap.n++);
	return GC_OK;
}
GC_HD void gc_fh_erase(gc_sk_t *s, int32_t pos)
{
	const int32_t x = s->heap.a[pos], last = s->heap.a[--s->heap.n];
	s->nd.a[x].hpos = -1;
	if (pos == s->heap.n) return;
	s->heap.a[pos] = last, s->nd.a[last].hpos = pos;
	gc_fh_up(s, pos);
	gc_fh_down(s, s->nd.a[last].hpos);
}
GC_HD int gc_sk_node(gc_arena_t *A, gc_sk_t *s, uint32_t v, int32_t d, uint32_t id, int32_t *out)
{
	gc_sknode_t *p;
	GC_PUSH(A, s->nd, p);
	p->v = v, p->di = (uint64_t)d << 32 | id, p->pre = -1, p->is_0 = 1, p->hpos = -1, p->hash = 0;
	*out = s->nd.n - 1;
	return GC_OK;
}
GC_HD int gc_sk_vertex(gc_arena_t *A, gc_sk_t *s, uint32_t v, int32_t *slot, int *absent) /* slot = index into tk */
{
	if (s->hcnt * 2 >= s->hcap) {
		const uint32_t ocap = s->hcap, ncap = ocap ? ocap * 2 : 64;
		const uint32_t *ok = s->hk; const int32_t *ov = s->hv;
		uint32_t *nk; int32_t *nv;
		GC_ALLOC(A, uint32_t, nk, ncap); GC_ALLOC(A, int32_t, nv, ncap);
		for (uint32_t j = 0; j < ncap; ++j) nv[j] = -1;
		for (uint32_t j = 0; j < ocap; ++j)
			if (ov[j] >= 0) {
				uint32_t q = gc_hash32(ok[j]) & (ncap - 1);
				while (nv[q] >= 0) q = (q + 1) & (ncap - 1);
				nk[q] = ok[j], nv[q] = ov[j];
			}
		s->hk = nk, s->hv = nv, s->hcap = ncap;
	}
	uint32_t i = gc_hash32(v) & (s->hcap - 1);
	while (s->hv[i] >= 0 && s->hk[i] != v) i = (i + 1) & (s->hcap - 1);
	*absent = s->hv[i] < 0;
	if (*absent) {
		gc_sklist_t *t;
		GC_PUSH(A, s->tk, t);
		t->k = 0;
		s->hk[i] = v, s->hv[i] = s->tk.n - 1, ++s->hcnt;
	}
	*slot = s->hv[i];
	return GC_OK;
}
/* the reference's max-heap sifts (ksort.h:44-66) on node indices compared by di */
GC_HD void gc_tk_up(const gc_sk_t *s, int32_t n, int32_t *l)
{
	int32_t k = n - 1;
	const int32_t tmp = l[k];
	while (k) {
		const int32_t i = (k - 1) >> 1;
		if (s->nd.a[tmp].di < s->nd.a[l[i]].di) break;
		l[k] = l[i], k = i;
	}
	l[k] = tmp;
}
GC_HD void gc_tk_down(const gc_sk_t *s, int32_t i, int32_t n, int32_t *l)
{
	int32_t k = i;
	const int32_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && s->nd.a[l[k]].di < s->nd.a[l[k + 1]].di) ++k;
		if (s->nd.a[l[k]].di < s->nd.a[tmp].di) break;
		l[i] = l[k], i = k;
	}
	l[i] = tmp;
}
GC_HD int32_t gc_grp_find(int32_t n, const uint64_t *grp, uint32_t v, int32_t *cnt) /* destinations grouped by vertex (sorted v<<32|index) */
{
	int32_t lo = 0, hi = n, e;
	while (lo < hi) { const int32_t m = (lo + hi) >> 1; if ((uint32_t)(grp[m] >> 32) < v) lo = m + 1; else hi = m; }
	if (lo == n || (uint32_t)(grp[lo] >> 32) != v) return -1;
	for (e = lo; e < n && (uint32_t)(grp[e] >> 32) == v; ++e) {}
	*cnt = e - lo;
	return lo;
}

/* Up to max_k shortest walks from src to every destination vertex within max_dist (shortk.c:41-242): a Dijkstra search in which a
 * vertex may be settled max_k times.  The frontier key dist<<32|serial is unique, so the settle order is a function of the arc
 * order alone.  walk != NULL: also return the settled walks needed to backtrack to the destinations (gc_walkv_t[], *n_walk). */
GC_HDN int gc_shortest_k(gc_arena_t *A, const gc_graph_t *G, uint32_t src, int32_t n_dst, gc_dst_t *dst, int32_t max_dist, int32_t max_k,
						gc_walkv_t **walk, int32_t *n_walk)
{
	if (walk) *walk = 0, *n_walk = 0;
	if (n_dst <= 0) return GC_OK;
	for (int32_t i = 0; i < n_dst; ++i) {
		gc_dst_t *t = &dst[i];
		if (t->inner) t->dist = 0, t->n_path = 1, t->path_end = -1;
		else t->dist = -1, t->n_path = 0, t->path_end = -1;
	}
	if (max_k > MG_MAX_SHORT_K) max_k = MG_MAX_SHORT_K;
	GC_STATE(gc_sk_t, S);
	memset(&S, 0, sizeof S);
	int8_t *dst_done;
	uint64_t *grp;
	GC_VEC(int32_t) out;
	gc_vec_zero(out);
	GC_ALLOC(A, int8_t, dst_done, n_dst);
	GC_ALLOC(A, uint64_t, grp, n_dst);
	for (int32_t i = 0; i < n_dst; ++i) dst_done[i] = 0, grp[i] = (uint64_t)dst[i].v << 32 | (uint32_t)i;
	gc_sort_u64(grp, n_dst);
	uint32_t id = 0;
	int32_t x, slot, n_done = 0;
	int absent;
	GC_TRY(gc_sk_node(A, &S, src, 0, id++, &x));
	S.nd.a[x].hash = gc_hash32(src);
	GC_TRY(gc_fh_push(A, &S, x));
	GC_TRY(gc_sk_vertex(A, &S, src, &slot, &absent));
	S.tk.a[slot].k = 1, S.tk.a[slot].p[0] = x;

	while (S.heap.n > 0) {
		const int32_t r = S.heap.a[0];
		int32_t cnt;
		gc_fh_erase(&S, 0); /* the closest unsettled walk */
		GC_TRY(gc_vec_reserve(A, out, out.n + 1));
		S.nd.a[r].di = S.nd.a[r].di >> 32 << 32 | (uint32_t)out.n;
		out.a[out.n++] = r;
		const uint32_t rv = S.nd.a[r].v, rhash = S.nd.a[r].hash;
		const int32_t rdist = (int32_t)(S.nd.a[r].di >> 32), ris0 = S.nd.a[r].is_0;

		const int32_t off = gc_grp_find(n_dst, grp, rv, &cnt);
		if (off >= 0) { /* a destination vertex was reached (shortk.c:116-153) */
			for (int32_t j = 0; j < cnt; ++j) {
				gc_dst_t *t = &dst[(int32_t)(uint32_t)grp[off + j]];
				int done = 0;
				if (t->inner) done = 1;
				else {
					int copy = 0;
					const int exact = rdist == t->target_dist && t->check_hash && rhash == t->target_hash;
					if (t->n_path == 0) copy = 1;
					else if (t->target_dist >= 0) {
						if (exact) copy = 1, done = 1;
						else {
							const int32_t d0 = t->dist > t->target_dist ? t->dist - t->target_dist : t->target_dist - t->dist;
							const int32_t d1 = rdist > t->target_dist ? rdist - t->target_dist : t->target_dist - rdist;
							if (d1 < d0) copy = 1;
						}
					}
					if (copy) {
						t->path_end = out.n - 1, t->dist = rdist, t->hash = rhash, t->is_0 = ris0;
						if (t->target_dist >= 0) {
							if (exact) done = 1;
							else if (rdist > t->target_dist + GC_SK_EXT) done = 1;
						}
					}
					if (++t->n_path >= max_k) done = 1;
				}
				if (dst_done[off + j] == 0 && done) dst_done[off + j] = 1, ++n_done;
			}
			if (n_done == n_dst) break;
		}

		const int32_t nv = gc_n_arc(G, rv);
		const gc_arc_t *av = gc_arcs(G, rv);
		for (int32_t i = 0; i < nv; ++i) { /* relax every arc, in arc order (shortk.c:157-188) */
			const uint32_t w = av[i].w;
			const int32_t d = rdist + (int32_t)(uint32_t)av[i].v_lv;
			if (d > max_dist) continue;
			GC_TRY(gc_sk_vertex(A, &S, w, &slot, &absent));
			gc_sklist_t *q = &S.tk.a[slot];
			if (q->k < max_k) {
				GC_TRY(gc_sk_node(A, &S, w, d, id++, &x));
				gc_sknode_t *nx = &S.nd.a[x];
				nx->pre = out.n - 1, nx->hash = rhash + gc_hash32(w), nx->is_0 = av[i].rank > 0 ? 0 : ris0;
				GC_TRY(gc_fh_push(A, &S, x));
				q->p[q->k++] = x;
				gc_tk_up(&S, q->k, q->p);
			} else if ((int32_t)(S.nd.a[q->p[0]].di >> 32) > d) { /* shorter than the longest kept walk to w: that one is replaced */
				x = q->p[0];
				if (S.nd.a[x].hpos < 0) return GC_E_BUG; /* shortk.c:182-186 */
				gc_fh_erase(&S, S.nd.a[x].hpos);
				gc_sknode_t *nx = &S.nd.a[x];
				nx->di = (uint64_t)d << 32 | (id++);
				nx->pre = out.n - 1, nx->hash = rhash + gc_hash32(w), nx->is_0 = av[i].rank > 0 ? 0 : ris0;
				GC_TRY(gc_fh_push(A, &S, x));
				gc_tk_down(&S, 0, q->k, q->p);
			}
		}
	}

	if (walk) { /* the settled walks that lead to a found destination, re-numbered (shortk.c:202-236) */
		int32_t n_found = 0;
		for (int32_t i = 0; i < n_dst; ++i) n_found += dst[i].n_path > 0;
		if (n_found > 0) {
			int32_t *trans, n = 0, cnt;
			GC_ALLOC(A, int32_t, trans, out.n > 0 ? out.n : 1);
			for (int32_t i = 0; i < out.n; ++i) trans[i] = 0;
			for (int32_t i = 0; i < n_dst; ++i) {
				const gc_dst_t *t = &dst[i];
				if (t->n_path > 0 && t->target_dist >= 0 && t->path_end >= 0) trans[(int32_t)(uint32_t)S.nd.a[out.a[t->path_end]].di] = 1;
			}
			for (int32_t i = 0; i < out.n; ++i) {
				const int32_t off = gc_grp_find(n_dst, grp, S.nd.a[out.a[i]].v, &cnt);
				if (off >= 0)
					for (int32_t j = off; j < off + cnt; ++j)
						if (dst[j].target_dist < 0) trans[i] = 1; /* NB: dst[] indexed by group position, as the reference does (shortk.c:215-217) */
			}
			for (int32_t i = out.n - 1; i >= 0; --i)
				if (trans[i] && S.nd.a[out.a[i]].pre >= 0) trans[S.nd.a[out.a[i]].pre] = 1;
			for (int32_t i = 0; i < out.n; ++i) trans[i] = trans[i] ? n++ : -1;
			gc_walkv_t *ret;
			GC_ALLOC(A, gc_walkv_t, ret, n > 0 ? n : 1);
			for (int32_t i = 0; i < out.n; ++i) {
				if (trans[i] < 0) continue;
				const gc_sknode_t *p = &S.nd.a[out.a[i]];
				ret[trans[i]].v = p->v, ret[trans[i]].d = (uint32_t)(p->di >> 32), ret[trans[i]].pre = p->pre < 0 ? p->pre : trans[p->pre];
			}
			for (int32_t i = 0; i < n_dst; ++i) if (dst[i].path_end >= 0) dst[i].path_end = trans[dst[i].path_end];
			*walk = ret, *n_walk = n;
		}
	}
	return GC_OK;
}

/* ------------------------------------------------------------------------------------------------ DP over the chains */

typedef struct { uint32_t srt; int32_t i; } gc_frag_t; /* srt = isolated<<31 | query end; i = chain index */

/* where the scan for predecessors of fragment n starts (find_max, gchain1.c:16-30): the last of f[0..n) when all query ends are below x,
 * none (-1) when none is, otherwise the FIRST fragment whose query end is >= x -- one past the last one below x, which the scan then
 * rejects or accepts on its own tests; the choice is the reference's and decides which candidates are seen */
GC_HD int32_t gc_scan_start(int32_t n, const gc_frag_t *f, uint32_t x)
{
	if (n == 0) return -1;
	if (f[n - 1].srt < x) return n - 1;
	if (f[0].srt >= x) return -1;
	int32_t lo = 0, hi = n;
	while (lo < hi) { const int32_t m = lo + ((hi - lo) >> 1); if (f[m].srt >= x) hi = m; else lo = m + 1; }
	return lo;
}

/* score of appending chain i behind destination dj (gchain1.c:39-60) */
GC_HD int32_t gc_link_score(const gc_dst_t *dj, const gc_chain_t *ci, const gc_chain_t *c, const mg128_t *an, const gc_frag_t *fr, const int32_t *f,
							int32_t bw, int32_t ref_bonus, float pen_gap, int32_t *ok)
{
	*ok = 0;
	if (dj->n_path == 0) return 0;
	const gc_chain_t *cj = &c[fr[dj->meta].i];
	int32_t gap = dj->dist - dj->target_dist, sc;
	if (gap < 0) gap = -gap;
	if (GC_ASEG(an[ci->off]) == GC_ASEG(an[cj->off + cj->cnt - 1]) && gap > bw) return 0;
	if (cj->qe <= ci->qs) sc = ci->score;
	else sc = (int32_t)((double)(ci->qe - cj->qe) / (ci->qe - ci->qs) * ci->score + .499); /* the part of chain i beyond the query overlap */
	if (dj->is_0) sc += ref_bonus;
	{
		const float lin = pen_gap * (float)gap, lg = gap >= 2 ? gc_log2f((float)gap) : 0.0f;
		sc -= (int32_t)(lin + lg);
	}
	sc += f[dj->meta];
	*ok = 1;
	return sc;
}

/* backtrack of a chaining DP (lchain.c:9-77) as gchain1.c:216 uses it (min_cnt = min_sc = 0, no drop limit): chain ends are taken in
 * descending f through the klib order; from an end the predecessors are followed up to a fragment already used, and the chain is CUT
 * where the score gained since that point is largest (first such point).  u[] gets score<<32|count, v[] the fragments, end first. */
GC_HD int gc_backtrack_all(gc_arena_t *A, int32_t n, const int32_t *f, const int64_t *p, int32_t *t, uint64_t *u, int32_t *v, int32_t *n_u_, int32_t *n_v_)
{
	const int64_t mark = A->top;
	gc_kv_t *z;
	int32_t n_z = 0, n_u = 0, n_v = 0;
	*n_u_ = *n_v_ = 0;
	for (int32_t i = 0; i < n; ++i) n_z += f[i] >= 0;
	if (n_z == 0) return GC_OK;
	GC_ALLOC(A, gc_kv_t, z, n_z);
	for (int32_t i = 0, k = 0; i < n; ++i) if (f[i] >= 0) z[k].key = (uint64_t)(int64_t)f[i], z[k++].val = (uint64_t)i;
	GC_TRY(gc_ksort(A, z, n_z, 8));
	for (int32_t i = 0; i < n; ++i) t[i] = 0;
	for (int32_t k = n_z - 1; k >= 0; --k) {
		const int32_t e = (int32_t)z[k].val, end_sc = (int32_t)z[k].key, n_v0 = n_v;
		if (t[e] != 0) continue;
		int64_t i = e, cut = e;
		int32_t best = 0;
		do { /* where to cut: the predecessor position with the largest partial score */
			i = p[i];
			const int32_t sc = i < 0 ? end_sc : end_sc - f[i];
			if (sc > best) best = sc, cut = i;
		} while (i >= 0 && t[i] == 0);
		for (i = e; i != cut; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
		const int32_t sc = i < 0 ? end_sc : end_sc - f[i];
		if (sc >= 0 && n_v > n_v0) u[n_u++] = (uint64_t)sc << 32 | (uint32_t)(n_v - n_v0);
		else n_v = n_v0;
	}
	A->top = mark;
	*n_u_ = n_u, *n_v_ = n_v;
	return GC_OK;
}

/* DP over the linear chains of a read in the order of their query ends; a chain may follow another one on the same segment
 * (colinear, within the band) or on a different one when the graph offers a walk of a fitting length (gchain1.c:62-240).
 * c[] is re-ordered chain by chain (each graph chain's members in query order); u[] gets score<<32|count per graph chain. */
GC_HD int gc_chain_dp(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, int32_t qlen, const mg128_t *an, gc_chain_t *c, int32_t *n_c_, uint64_t **u_, int32_t *n_u_, int32_t *n_shortk)
{
	const int32_t n_c = *n_c_, max_dist_g = P->bw_long, max_dist_q = P->bw_long, bw = P->bw_long;
	*u_ = 0, *n_u_ = 0;
	if (n_c == 0) return GC_OK;
	uint64_t *u;
	GC_ALLOC(A, uint64_t, u, n_c);
	gc_frag_t *fr;
	gc_kv_t *z;
	int32_t n_ext = 0;
	GC_ALLOC(A, gc_frag_t, fr, n_c);
	/* a chain far from both ends of its segment (or short relative to that distance) cannot be linked through the graph */
	for (int32_t i = 0; i < n_c; ++i) {
		gc_chain_t *r = &c[i];
		int32_t to_end = gc_vlen(G, r->v) - r->re;
		r->dist_pre = -1;
		if (r->rs < to_end) to_end = r->rs;
		const int isolated = to_end > max_dist_g || (to_end >> 3) > r->score;
		fr[i].srt = (uint32_t)isolated << 31 | (uint32_t)r->qe, fr[i].i = i;
		n_ext += !isolated;
	}
	if (n_ext < 2) { /* nothing to link: every chain is a graph chain of its own */
		for (int32_t i = 0; i < n_c; ++i) u[i] = (uint64_t)c[i].score << 32 | 1;
		*u_ = u, *n_u_ = n_c;
		return GC_OK;
	}
	{ /* radix_sort_gc: klib sort on the 4-byte key (gchain1.c:13-14,100) */
		const int64_t mark = A->top;
		GC_ALLOC(A, gc_kv_t, z, n_c);
		for (int32_t i = 0; i < n_c; ++i) z[i].key = fr[i].srt, z[i].val = (uint64_t)fr[i].i;
		GC_TRY(gc_ksort(A, z, n_c, 4));
		for (int32_t i = 0; i < n_c; ++i) fr[i].srt = (uint32_t)z[i].key, fr[i].i = (int32_t)z[i].val;
		A->top = mark;
	}
	int32_t *f, *v, *t;
	int64_t *p;
	GC_ALLOC(A, int32_t, v, n_c); GC_ALLOC(A, int32_t, f, n_ext); GC_ALLOC(A, int64_t, p, n_ext); GC_ALLOC(A, int32_t, t, n_ext);
	for (int32_t i = 0; i < n_ext; ++i) t[i] = 0;
	GC_VEC(gc_dst_t) cand;
	gc_vec_zero(cand);
	for (int32_t i = 0; i < n_ext; ++i) {
		gc_chain_t *ci = &c[fr[i].i];
		const int32_t segi = GC_ASEG(an[ci->off]);
		cand.n = 0;
		{ /* candidate predecessors, nearest query end first (gchain1.c:113-175) */
			int32_t x = ci->qs + bw, n_skip = 0;
			if (x > qlen) x = qlen;
			for (int32_t j = gc_scan_start(i, fr, (uint32_t)x); j >= 0; --j) {
				const gc_chain_t *cj = &c[fr[j].i];
				int32_t target;
				if (cj->qs >= ci->qs) continue; /* contained on the query */
				if (cj->qe > ci->qs) { /* query overlap */
					const int32_t o = cj->qe - ci->qs;
					if ((float)o > (float)(cj->qe - cj->qs) * P->mask_level || (float)o > (float)(ci->qe - ci->qs) * P->mask_level) continue;
				}
				const int32_t dq = ci->qs - cj->qe, segj = GC_ASEG(an[cj->off + cj->cnt - 1]);
				if (segi == segj) { if (dq > max_dist_q) break; }
				else if (dq > max_dist_g && dq > max_dist_q) break;
				if (ci->v != cj->v) { /* different segments: the graph gap is at least what is left of both segments */
					const int32_t min_dist = ci->rs + (gc_vlen(G, cj->v) - cj->re);
					if (min_dist > max_dist_g) continue;
					if (segi == segj && min_dist - bw > ci->qs - cj->qe) continue;
					target = (ci->qs - cj->qe) - (gc_vlen(G, cj->v) - cj->re) + (gc_vlen(G, ci->v) - ci->rs); /* mg_target_dist, gchain1.c:32-37 */
					if (target < 0) continue;
				} else {
					if (cj->rs >= ci->rs || cj->re >= ci->re) continue; /* not colinear */
					const int32_t dr = ci->rs - cj->re, w = dr > dq ? dr - dq : dq - dr;
					if (segi == segj && w > bw) continue;
					if (dr > max_dist_g || dr < -max_dist_g) continue;
					if (cj->re > ci->rs) {
						const int32_t o = cj->re - ci->rs;
						if ((float)o > (float)(cj->re - cj->rs) * P->mask_level || (float)o > (float)(ci->re - ci->rs) * P->mask_level) continue;
					}
					target = (ci->qs - cj->qe) - (gc_vlen(G, cj->v) - cj->re) + (gc_vlen(G, ci->v) - ci->rs);
				}
				gc_dst_t *q;
				GC_PUSH(A, cand, q);
				memset(q, 0, sizeof *q);
				q->inner = ci->v == cj->v, q->v = cj->v ^ 1, q->meta = j, q->target_dist = target;
				if (t[j] == i && ++n_skip > P->max_gc_skip) break;
				if (p[j] >= 0) t[p[j]] = i;
			}
		}
		{ /* reachability + distance through the graph, no sequences involved */
			const int64_t mark = A->top;
			int rc = GC_E_ARENA;
			if (A->fast_base) { gc_arena_t F; gc_arena_init(&F, A->fast_base, A->fast_cap, 0); rc = gc_shortest_k(&F, G, ci->v ^ 1, cand.n, cand.a, max_dist_g + (gc_vlen(G, ci->v) - ci->rs), MG_MAX_SHORT_K, 0, 0); }
			if (rc == GC_E_ARENA) rc = gc_shortest_k(A, G, ci->v ^ 1, cand.n, cand.a, max_dist_g + (gc_vlen(G, ci->v) - ci->rs), MG_MAX_SHORT_K, 0, 0);
			if (rc == GC_E_ARENA) return rc; /* (GC_E_BUG: the search stopped early, what it had found stands -- the reference tears its allocator down there) */
			A->top = mark; /* the search's scratch; cand was allocated before it */
			++*n_shortk;
		}
		int32_t best_f = ci->score, best_j = -1, best_d = -1, best_inner = 0;
		uint32_t best_hash = 0;
		for (int32_t j = 0; j < cand.n; ++j) {
			const gc_dst_t *dj = &cand.a[j];
			int32_t ok;
			const int32_t sc = gc_link_score(dj, ci, c, an, fr, f, bw, P->ref_bonus, P->chn_pen_gap, &ok);
			if (!ok || sc + ci->score < 0) continue;
			if (sc > best_f) best_f = sc, best_j = dj->meta, best_d = dj->dist, best_hash = dj->hash, best_inner = dj->inner;
		}
		f[i] = best_f, p[i] = best_j;
		ci->dist_pre = best_d, ci->hash_pre = best_hash, ci->inner_pre = best_inner;
		v[i] = best_j >= 0 && v[best_j] > best_f ? v[best_j] : best_f;
	}
	int32_t n_u, n_v;
	GC_TRY(gc_backtrack_all(A, n_ext, f, p, t, u, v, &n_u, &n_v));
	for (int32_t i = 0; i < n_c - n_ext; ++i) { /* the isolated chains behind the linked ones */
		u[n_u++] = (uint64_t)c[fr[n_ext + i].i].score << 32 | 1;
		v[n_v++] = n_ext + i;
	}
	gc_chain_t *tmp;
	GC_ALLOC(A, gc_chain_t, tmp, n_v > 0 ? n_v : 1);
	for (int32_t i = 0, k = 0; i < n_u; ++i) {
		const int32_t k0 = k, ni = (int32_t)(uint32_t)u[i];
		for (int32_t j = 0; j < ni; ++j) tmp[k++] = c[fr[v[k0 + (ni - j - 1)]].i];
	}
	memcpy(c, tmp, (size_t)n_v * sizeof(gc_chain_t));
	*n_c_ = n_v, *u_ = u, *n_u_ = n_u;
	return GC_OK;
}

/* ------------------------------------------------------------------------------------------------ GWFA */

#if defined(GC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
/* host-only statistics of the GWFA steps (a profiling aid: -DGC_STATS, printed by gc_stats_dump()) */
typedef struct { long long calls, steps, cells, runs, heads0, heads, out, single_nohead, single, nohead, le64, sum_ql, reached, steps_hist[8], n_hist[8], simple_cells, dedup_steps, long_runs, arena_hist[8], arena_sum, fit16, fit32, fit64, dedup_calls, dd_n[8], dd_done[8], dd_fresh[8], dd_nc[8], dd_sum_n, dd_sum_done, dd_sum_fresh, dd_sum_nc, dd_unsorted, hd_arc, hd_in, hd_other; } gc_stats_t;
static gc_stats_t gc_stats;
static inline int gc_stats_bin(long long v) { int b = 0; while (v > 1 && b < 7) v >>= 2, ++b; return b; }
static void gc_stats_dump(void)
{
	const gc_stats_t *S = &gc_stats;
	fprintf(stderr, "[gc_stats] calls %lld reached %lld mean_ql %.1f steps %lld (%.1f/call) cells %lld (%.1f/step) runs %.2f/step heads0 %.2f/step heads %.2f/step out %.1f/step\n", S->calls, S->reached, (double)S->sum_ql / (S->calls ? S->calls : 1),
			S->steps, (double)S->steps / (S->calls ? S->calls : 1), S->cells, (double)S->cells / (S->steps ? S->steps : 1), (double)S->runs / (S->steps ? S->steps : 1), (double)S->heads0 / (S->steps ? S->steps : 1), (double)S->heads / (S->steps ? S->steps : 1), (double)S->out / (S->steps ? S->steps : 1));
	fprintf(stderr, "[gc_stats] steps with one run and no head cell %.3f (cells in them %.3f); one run %.3f; no head %.3f; all runs <= 64 %.3f; dedup %.3f\n", (double)S->single_nohead / S->steps, (double)S->simple_cells / S->cells, (double)S->single / S->steps, (double)S->nohead / S->steps, (double)S->le64 / S->steps, (double)S->dedup_steps / S->steps);
	fprintf(stderr, "[gc_stats] steps per call (bins 1,4,16,64,256,1k,4k,16k+):"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->steps_hist[i]); fprintf(stderr, "\n");
	fprintf(stderr, "[gc_stats] arena bytes per call: mean %.0f; <=16K %.3f <=32K %.3f <=64K %.3f; hist (0.5K,2K,8K,32K,128K,512K,2M,8M+):", (double)S->arena_sum / S->calls, (double)S->fit16 / S->calls, (double)S->fit32 / S->calls, (double)S->fit64 / S->calls); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->arena_hist[i]); fprintf(stderr, "\n");
	fprintf(stderr, "[gc_stats] dedup: mean cells %.1f done %.1f fresh %.2f flagged %.1f unsorted %.3f\n", (double)S->dd_sum_n / S->dedup_calls, (double)S->dd_sum_done / S->dedup_calls, (double)S->dd_sum_fresh / S->dedup_calls, (double)S->dd_sum_nc / S->dedup_calls, (double)S->dd_unsorted / S->dedup_calls);
	fprintf(stderr, "[gc_stats] dedup cells hist (<=1,<8,<32,<128,<512,..):"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->dd_n[i]); fprintf(stderr, "; done hist:"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->dd_done[i]);
	fprintf(stderr, "; fresh hist:"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->dd_fresh[i]); fprintf(stderr, "; flagged hist:"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->dd_nc[i]); fprintf(stderr, "\n");
	fprintf(stderr, "[gc_stats] head cells: inside a vertex %lld, arc fan-out %lld, other %lld\n", S->hd_in, S->hd_arc, S->hd_other);
	fprintf(stderr, "[gc_stats] cells per step (same bins):"); for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", S->n_hist[i]); fprintf(stderr, "\n");
}
#define GC_STAT(...) __VA_ARGS__
#else
#define GC_STAT(...)
#endif

#define GC_DSHIFT 0x40000000
#define GC_FAST_GAP 400 /* longest query gap whose GWFA call tries the LDS scratch first */
/* 32 bytes, 16-byte aligned: a cell is moved as two aligned 16-byte words.  (24 bytes at 8-byte alignment let the compiler fuse a copy into a 16-byte access at
 * an address that is only 8-byte aligned -- fine in HBM, an aperture fault when the flat address lies in LDS, which is where a call's scratch lives now.) */
typedef struct __attribute__((aligned(16))) { uint64_t vd; int32_t k, len; uint32_t xo; int32_t t; int32_t pad_[2]; } gc_diag_t;   /* vd = vertex<<32 | (DSHIFT + diagonal); xo = edits-ish<<1 | out-of-order */
typedef struct { uint64_t vd0, vd1; } gc_intv_t;
typedef struct { int32_t v, pre; } gc_trace_t;
typedef GC_VEC(gc_diag_t) gc_diag_v;
typedef GC_VEC(gc_intv_t) gc_intv_v;
typedef struct { GC_F uint64_t *k; GC_F int32_t *v; GC_F uint32_t *used; GC_F uint32_t cap; GC_F uint32_t cnt; } gc_u64map_t; /* used[0..cnt): the slots filled since the last clear */

GC_HD uint64_t gc_mk_vd(uint32_t v, int32_t d) { return (uint64_t)v << 32 | (uint32_t)(GC_DSHIFT + d); }
GC_HD uint32_t gc_u64slot(uint64_t key, uint32_t cap) { return (uint32_t)((key ^ key >> 29) * 0x9E3779B97F4A7C15ULL >> 40) & (cap - 1); }
GC_HDN int gc_u64map_put(gc_arena_t *A, gc_u64map_t *h, uint64_t key, int32_t **val, int *absent)
{
	if (h->cnt * 2 >= h->cap) {
		const uint32_t ocap = h->cap, ncap = ocap ? ocap * 2 : 64;
		const uint64_t *ok = h->k; const int32_t *ov = h->v;
		uint64_t *nk; int32_t *nv; uint32_t *nu;
		GC_ALLOC(A, uint64_t, nk, ncap); GC_ALLOC(A, int32_t, nv, ncap); GC_ALLOC(A, uint32_t, nu, ncap / 2 + 1);
		for (uint32_t j = 0; j < ncap; ++j) nk[j] = ~0ULL;
		uint32_t n_used = 0;
		for (uint32_t j = 0; j < ocap; ++j)
			if (ok[j] != ~0ULL) { uint32_t q = gc_u64slot(ok[j], ncap); while (nk[q] != ~0ULL) q = (q + 1) & (ncap - 1); nk[q] = ok[j], nv[q] = ov[j], nu[n_used++] = q; }
		h->k = nk, h->v = nv, h->used = nu, h->cap = ncap;
	}
	uint32_t i = gc_u64slot(key, h->cap);
	while (h->k[i] != ~0ULL && h->k[i] != key) i = (i + 1) & (h->cap - 1);
	*absent = h->k[i] == ~0ULL;
	if (*absent) h->k[i] = key, h->used[h->cnt++] = i;
	*val = &h->v[i];
	return GC_OK;
}
/* empty again: only the slots that were filled are touched ([measured, round 4, host] wiping the whole table at every GWFA step was 10 % of the host's chaining cycles) */
GC_HD void gc_u64map_clear(gc_u64map_t *h) { if (h->cnt == 0) return; GC_PAR_FOR(j, (int32_t)h->cnt) h->k[h->used[j]] = ~0ULL; gc_sync(); h->cnt = 0; }

typedef struct {
	GC_F const gc_graph_t *G;
	GC_F int32_t ql; GC_F const char *q;
	GC_F int32_t max_chk; GC_F int32_t bw_dyn; GC_F int32_t max_lag;
	GC_F int64_t i_term;
	gc_u64map_t seen, tnode;      /* (vertex, query position) entered in the current step; traceback node dedup */
	gc_intv_v done, fresh, swap;  /* finished diagonals: merged list, this step's additions, scratch */
	gc_diag_v ooo, wf[2], head;   /* sort scratch; the two wavefronts; the cells sitting on a vertex or query end */
	GC_F gc_kv_t *sort_kv; GC_F gc_diag_t *sort_tmp; GC_F int32_t m_sort;
	GC_VEC(gc_trace_t) tr;
	GC_F int32_t cur; GC_F int32_t s; GC_F int32_t end_tb; GC_F int32_t end_off;
	GC_F uint32_t end_v;
} gc_gw_t;

GC_HD int gc_trace_push(gc_arena_t *A, gc_gw_t *z, int32_t v, int32_t pre, int32_t *id) /* gfa-ed.c:213-227 */
{
	int absent;
	int32_t *val;
	GC_TRY(gc_u64map_put(A, &z->tnode, (uint64_t)(uint32_t)v << 32 | (uint32_t)pre, &val, &absent));
	if (absent) {
		gc_trace_t *t;
		GC_PUSH(A, z->tr, t);
		t->v = v, t->pre = pre;
		*val = z->tr.n - 1;
	}
	*id = *val;
	return GC_OK;
}
GC_HD int gc_diag_push(gc_arena_t *A, gc_diag_v *a, uint32_t v, int32_t d, int32_t k, uint32_t x, uint32_t ooo, int32_t t)
{
	gc_diag_t *p;
	GC_PUSH(A, *a, p);
	p->vd = gc_mk_vd(v, d), p->k = k, p->xo = x << 1 | ooo, p->t = t, p->len = 0;
	return GC_OK;
}
GC_HD int gc_diag_update(gc_diag_t *p, uint32_t v, int32_t d, int32_t k, uint32_t x, uint32_t ooo, int32_t t) /* gfa-ed.c:120-131 */
{
	if (p->vd == gc_mk_vd(v, d)) {
		if (!(p->k > k)) p->xo = x << 1 | ooo, p->t = t, p->k = k;
		return 0;
	}
	return 1;
}
/* furthest target offset reachable by exact matches from k on diagonal d of a vertex of length vl (gfa-ed.c:305-329), 8 bases per compare */
GC_HD int32_t gc_extend1(int32_t d, int32_t k, int32_t vl, const char *ts, int32_t ql, const char *qs)
{
	const int32_t max_k = (ql - d < vl ? ql - d : vl) - 1;
	const char *t = ts + 1, *q = qs + d + 1;
	while (k + 8 <= max_k) {
		uint64_t x, y;
		memcpy(&x, t + k, 8); memcpy(&y, q + k, 8);
		if (x != y) {
			const uint64_t z = x ^ y;
#if defined(__HIP_DEVICE_COMPILE__)
			return k + ((__ffsll((long long)z) - 1) >> 3);
#else
			return k + (__builtin_ctzll(z) >> 3);
#endif
		}
		k += 8;
	}
	while (k < max_k && t[k] == q[k]) ++k;
	return k;
}
GC_HD int32_t gc_intv_merge(int32_t n, gc_intv_t *a) /* gfa-ed.c:69-82 */
{
	if (n == 0) return 0;
	uint64_t st = a[0].vd0, en = a[0].vd1;
	int32_t k = 0;
	for (int32_t i = 1; i < n; ++i) {
		if (a[i].vd0 > en) { a[k].vd0 = st, a[k++].vd1 = en; st = a[i].vd0, en = a[i].vd1; }
		else en = en > a[i].vd1 ? en : a[i].vd1;
	}
	a[k].vd0 = st, a[k++].vd1 = en;
	return k;
}
/* sort a[] by vd: in-order cells keep their place, the flagged subset goes through the klib sort, stable merge (gfa-ed.c:143-171).
 * The split and the merge are done by rank -- an element's place is its own index plus the number of elements of the OTHER list that go
 * in front of it (ties: the in-order list first), found by binary search -- so every lane moves its own elements. */
GC_HDN int gc_diag_sort(gc_arena_t *A, gc_gw_t *z, int32_t n_a, gc_diag_t *a)
{
	int32_t n_c = 0;
	GC_TRY(gc_vec_reserve(A, z->ooo, n_a));
#if (GC_PARSORT & 1)
	for (int32_t base = 0; base < n_a; base += GC_NLANE) { const int32_t i = base + GC_LANE; n_c += gc_popc(gc_ballot(i < n_a && (a[i].xo & 1))); }
#else
	for (int32_t i = 0; i < n_a; ++i) n_c += a[i].xo & 1;
#endif
	const int32_t n_b = n_a - n_c;
	gc_diag_t *b = z->ooo.a, *c = b + n_b;
#if (GC_PARSORT & 1)
	for (int32_t base = 0, kb = 0, kc = 0; base < n_a; base += GC_NLANE) {
		const int32_t i = base + GC_LANE;
		const int in = i < n_a, flag = in && (a[i].xo & 1);
		const uint64_t mb = gc_ballot(in && !flag), mc = gc_ballot(flag);
		if (in) { if (flag) c[kc + gc_rank(mc)] = a[i]; else b[kb + gc_rank(mb)] = a[i]; }
		kb += gc_popc(mb), kc += gc_popc(mc);
	}
	gc_sync();
#else
	for (int32_t i = 0, j = 0, k = 0; i < n_a; ++i) { if (a[i].xo & 1) c[k++] = a[i]; else b[j++] = a[i]; }
#endif
	if (n_c > 1) {
		if (z->m_sort < n_c) {
			z->m_sort = n_c + (n_c >> 1) + 16;
			GC_ALLOC(A, gc_kv_t, z->sort_kv, z->m_sort);
			GC_ALLOC(A, gc_diag_t, z->sort_tmp, z->m_sort);
		}
		if (n_c <= 64) { /* klib sorts up to 64 records by insertion (ksort.h:118-128,160), a STABLE sort: every lane finds the place of its own record --
			              * records with a smaller key, plus equal ones in front of it -- and puts it there */
			GC_PAR_FOR(i, n_c) {
				const gc_diag_t me = c[i];
				int32_t r = 0;
				for (int32_t j = 0; j < n_c; ++j) { const uint64_t kj = c[j].vd; r += (kj < me.vd) | ((kj == me.vd) & (j < i)); }
				z->sort_tmp[r] = me;
			}
			gc_sync();
			GC_PAR_FOR(i, n_c) c[i] = z->sort_tmp[i];
			gc_sync();
		} else { /* the radix passes of the klib sort are not stable: replayed move by move */
			for (int32_t i = 0; i < n_c; ++i) z->sort_kv[i].key = c[i].vd, z->sort_kv[i].val = (uint64_t)i;
			GC_TRY(gc_ksort(A, z->sort_kv, n_c, 8));
			for (int32_t i = 0; i < n_c; ++i) z->sort_tmp[i] = c[z->sort_kv[i].val];
			memcpy(c, z->sort_tmp, (size_t)n_c * sizeof(gc_diag_t));
		}
	}
#if (GC_PARSORT & 4)
	GC_PAR_FOR(k, n_c) c[k].xo &= 0xfffffffeU;
	int b_unsorted = 0;
	GC_PAR_FOR(i, n_b) if (i > 0 && b[i - 1].vd > b[i].vd) b_unsorted = 1;
	gc_sync();
	if (!gc_ballot(b_unsorted)) {
		GC_PAR_FOR(i, n_b) { /* b[i] goes behind the c's that are strictly smaller */
			const uint64_t vd = b[i].vd;
			int32_t lo = 0, hi = n_c;
			while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (c[m].vd < vd) lo = m + 1; else hi = m; }
			a[i + lo] = b[i];
		}
		GC_PAR_FOR(j, n_c) { /* c[j] goes behind the b's that are smaller or equal */
			const uint64_t vd = c[j].vd;
			int32_t lo = 0, hi = n_b;
			while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (b[m].vd <= vd) lo = m + 1; else hi = m; }
			a[j + lo] = c[j];
		}
		gc_sync();
		return GC_OK;
	}
#else
	for (int32_t k = 0; k < n_c; ++k) c[k].xo &= 0xfffffffeU;
#endif
	{ /* sequential stable merge (with in-order cells that are not in order -- never seen -- it is what the reference would do) */
		int32_t i = 0, j = 0, k = 0;
		while (i < n_b && j < n_c) { if (b[i].vd <= c[j].vd) a[k++] = b[i++]; else a[k++] = c[j++]; }
		while (i < n_b) a[k++] = b[i++];
		while (j < n_c) a[k++] = c[j++];
	}
	return GC_OK;
}
/* add [x0, x1) to a canonical list of finished-diagonal ranges (ascending, disjoint, not touching): what sorting the new intervals in and
 * coalescing everything that overlaps or touches (gfa-ed.c:69-82,258-264) leaves, without walking the whole list -- a binary search, then
 * the ranges the new one reaches are replaced by their union.  The union of intervals has ONE canonical form, so the list is the reference's. */
GC_HDN int gc_intv_add(gc_arena_t *A, gc_intv_v *L, uint64_t x0, uint64_t x1)
{
	int32_t lo = 0, hi = L->n;
	while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (L->a[m].vd1 < x0) lo = m + 1; else hi = m; } /* first range that ends at or behind x0 */
	int32_t e = lo; /* ranges lo..e-1 overlap or touch [x0, x1) */
	while (e < L->n && L->a[e].vd0 <= x1) ++e;
	if (e == lo) { /* touches nothing: a new range at lo */
		GC_TRY(gc_vec_reserve(A, *L, L->n + 1));
		if (lo < L->n) gc_pmove_up(&L->a[lo + 1], &L->a[lo], (int64_t)(L->n - lo) * (int64_t)sizeof(gc_intv_t));
		L->a[lo].vd0 = x0, L->a[lo].vd1 = x1;
		++L->n;
		return GC_OK;
	}
	const uint64_t n0 = L->a[lo].vd0 < x0 ? L->a[lo].vd0 : x0, n1 = L->a[e - 1].vd1 > x1 ? L->a[e - 1].vd1 : x1;
	L->a[lo].vd0 = n0, L->a[lo].vd1 = n1;
	if (e - lo > 1) {
		if (e < L->n) gc_pmove_down(&L->a[lo + 1], &L->a[e], (int64_t)(L->n - e) * (int64_t)sizeof(gc_intv_t));
		L->n -= e - lo - 1;
	}
	return GC_OK;
}

/* in-place, order-preserving removal of the elements of a[0..n) that `drop` marks; the verdict of element i is computed by the lane that
 * owns it, a ballot gives every kept element its new place.  dst <= src, and a block is loaded completely before it is stored. */
#define GC_COMPACT_BEGIN(n_) { const int32_t gcn_ = (n_); int32_t gcm_ = 0; for (int32_t gcb_ = 0; gcb_ < gcn_; gcb_ += GC_NLANE) { const int32_t gci_ = gcb_ + GC_LANE; const int gcin_ = gci_ < gcn_;
#define GC_COMPACT_END(a_, val_, keep_, n_out_) const uint64_t gck_ = gc_ballot(gcin_ && (keep_)); gc_sync(); if (gcin_ && (keep_)) (a_)[gcm_ + gc_rank(gck_)] = (val_); gcm_ += gc_popc(gck_); gc_sync(); } (n_out_) = gcm_; }

#if defined(__HIP_DEVICE_COMPILE__)
GC_HD uint64_t gc_shfl64(uint64_t v, int src) { const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src); return (uint64_t)hi << 32 | lo; }
/* this step's finished diagonals join the list: sequentially that is a binary search and a shift per interval ([measured] most of what the dedup of a long bridge cost: ~30 intervals
 * per step).  The list is the canonical union of everything added so far, so a batch can be merged in one go: the list's ranges and the new ones side by side in the lanes (<= 64
 * together; more new ones come in several batches), sorted by start through their ranks, a running maximum of the ends tells where a range starts that touches nothing before it. */
GC_HDN int gc_intv_add_wave(gc_arena_t *A, gc_intv_v *L, int32_t n_f, const gc_intv_t *f)
{
	__shared__ __attribute__((aligned(16))) gc_intv_t iv_lds[64];
	const int32_t lane = GC_LANE;
	int32_t done_f = 0;
	while (done_f < n_f) {
		const int32_t n_d = L->n;
		if (n_d >= 48) { /* a long list: one by one (the generic way) */
			for (int32_t i = done_f; i < n_f; ++i) GC_TRY(gc_intv_add(A, L, f[i].vd0, f[i].vd1));
			return GC_OK;
		}
		const int32_t take = n_f - done_f < 64 - n_d ? n_f - done_f : 64 - n_d, T = n_d + take;
		GC_TRY(gc_vec_reserve(A, *L, T));
		uint64_t s0 = ~0ULL, e0 = 0;
		if (lane < n_d) s0 = L->a[lane].vd0, e0 = L->a[lane].vd1;
		else if (lane < T) s0 = f[done_f + lane - n_d].vd0, e0 = f[done_f + lane - n_d].vd1;
		int32_t r = 0;
		for (int32_t j = 0; j < T; ++j) { const uint64_t sj = gc_shfl64(s0, j); r += (sj < s0) | ((sj == s0) & (j < lane)); }
		if (lane < T) iv_lds[r].vd0 = s0, iv_lds[r].vd1 = e0;
		gc_sync();
		uint64_t s1 = ~0ULL, e1 = 0;
		if (lane < T) s1 = iv_lds[lane].vd0, e1 = iv_lds[lane].vd1;
		gc_sync();
		uint64_t pm = e1; /* inclusive running maximum of the ends */
		for (int32_t d = 1; d < 64; d <<= 1) { const uint64_t o = gc_shfl64(pm, lane >= d ? lane - d : lane); if (lane >= d && o > pm) pm = o; }
		uint64_t before = gc_shfl64(pm, lane > 0 ? lane - 1 : 0);
		const int start = lane < T && (lane == 0 || s1 > before); /* touches nothing before it (ranges that touch are one range: gfa-ed.c:69-82) */
		const uint64_t ms = __ballot(start);
		const uint64_t above = lane < 63 ? ms & ~((2ULL << lane) - 1ULL) : 0ULL; /* starts behind mine */
		const int32_t last = above ? (int32_t)__ffsll((long long)above) - 2 : T - 1; /* the last range of my group */
		const uint64_t en = gc_shfl64(pm, last < 0 ? 0 : last);
		if (start) { gc_intv_t *o = &L->a[gc_rank(ms)]; o->vd0 = s1, o->vd1 = en; }
		L->n = gc_popc(ms);
		gc_sync();
		done_f += take;
	}
	return GC_OK;
}
/* gwf_dedup on the device with few dependent trips to memory (round 4: [measured] the generic routine below was 56-66 % of a long bridge's time, ~120 k cycles per call on an otherwise
 * idle GPU -- a dozen passes, binary searches through HBM / L2 per cell).  Same result, three passes:
 *   A  every cell once: is the list out of order at all?  in-order cells close ranks in the scratch vector, the flagged ones (a handful: what the head cells pushed) go to LDS;
 *      the flagged ones are sorted by rank in registers (<= 64: klib's insertion sort is stable, so is a rank), each finds its place among the in-order ones by ONE binary
 *      search; every in-order cell finds its place by comparing with the sorted flagged keys broadcast lane by lane (no memory);
 *   B  the merged list in blocks that END AT A GROUP BOUNDARY (a lane also fetches the key behind the block), so that the furthest-cell-of-its-(vertex, diagonal) test sees whole
 *      groups through lane shuffles; the finished-diagonal test follows for the survivors; the output goes to the scratch vector, which then BECOMES the wavefront (the two vector
 *      headers are swapped) -- no in-place compaction, no barrier between blocks.
 * *handled = 0: left to the generic routine (more than 64 flagged cells, in-order cells that are not in order, a group that fills a block): B is valid input for it at that point. */
GC_HDN int gc_gw_dedup_wave(gc_arena_t *A, gc_gw_t *z, gc_diag_v *B, int *handled)
{
	__shared__ __attribute__((aligned(16))) gc_diag_t c_lds[2][64];
	const int32_t n = B->n, lane = GC_LANE;
	*handled = 0;
	GC_TRY(gc_vec_reserve(A, z->ooo, n > 0 ? n : 1));
	gc_diag_t *a = B->a, *tmp = z->ooo.a;
	int unsorted = 0, b_unsorted = 0;
	int32_t n_b = 0, n_c = 0;
	uint64_t carry_vd = 0, carry_b = 0;
	for (int32_t base = 0; base < n; base += 64) { /* ---- pass A ---- */
		const int32_t i = base + lane;
		const int in = i < n;
		gc_diag_t me;
		me.vd = ~0ULL, me.k = 0, me.len = 0, me.xo = 0, me.t = 0, me.pad_[0] = me.pad_[1] = 0;
		if (in) me = a[i];
		uint64_t pv = gc_shfl64(me.vd, lane > 0 ? lane - 1 : 0);
		if (lane == 0) pv = carry_vd;
		unsorted |= in && pv > me.vd;
		const int flag = in && (me.xo & 1);
		const uint64_t mc = __ballot(flag), mb = __ballot(in && !flag);
		const uint64_t lower = mb & ((1ULL << lane) - 1ULL);
		uint64_t pb = gc_shfl64(me.vd, lower ? 63 - __clzll((long long)lower) : lane); /* the in-order cell before mine */
		if (!lower) pb = carry_b;
		b_unsorted |= in && !flag && pb > me.vd;
		const int32_t last = n - base > 64 ? 63 : n - base - 1;
		carry_vd = gc_shfl64(me.vd, last);
		if (mb) carry_b = gc_shfl64(me.vd, 63 - __clzll((long long)mb));
		if (flag) { const int32_t ci = n_c + gc_rank(mc); if (ci < 64) c_lds[0][ci] = me; }
		else if (in) tmp[n_b + gc_rank(mb)] = me;
		n_b += gc_popc(mb), n_c += gc_popc(mc);
	}
	if (__ballot(unsorted)) {
		if (n_c > 64 || __ballot(b_unsorted)) return GC_OK; /* the generic routine */
		gc_sync();
		gc_diag_t c;
		uint64_t key = ~0ULL;
		c.vd = ~0ULL, c.k = 0, c.len = 0, c.xo = 0, c.t = 0, c.pad_[0] = c.pad_[1] = 0;
		if (lane < n_c) c = c_lds[0][lane], key = c.vd;
		int32_t r = 0;
		for (int32_t j = 0; j < n_c; ++j) { const uint64_t kj = gc_shfl64(key, j); r += (kj < key) | ((kj == key) & (j < lane)); } /* cells with a smaller key, and equal ones before mine */
		if (lane < n_c) { c.xo &= 0xfffffffeU; c_lds[1][r] = c; }
		gc_sync();
		uint64_t ck = ~0ULL;
		if (lane < n_c) { /* the sorted flagged cells: mine goes behind the in-order cells that are smaller or equal */
			c = c_lds[1][lane], ck = c.vd;
			int32_t lo = 0, hi = n_b;
			while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (tmp[m].vd <= ck) lo = m + 1; else hi = m; }
			a[lane + lo] = c;
		}
		for (int32_t base = 0; base < n_b; base += 64) { /* an in-order cell goes behind the flagged ones that are strictly smaller */
			const int32_t i = base + lane;
			gc_diag_t me;
			me.vd = 0, me.k = 0, me.len = 0, me.xo = 0, me.t = 0, me.pad_[0] = me.pad_[1] = 0;
			if (i < n_b) me = tmp[i];
			int32_t lo = 0;
			for (int32_t j = 0; j < n_c; ++j) lo += gc_shfl64(ck, j) < me.vd;
			if (i < n_b) a[i + lo] = me;
		}
		gc_sync();
	}
	{ /* ---- pass B ---- */
		const int32_t n_done = z->done.n;
		const gc_intv_t *dn = z->done.a;
		int32_t m = 0;
		for (int32_t base = 0; base < n;) {
			const int32_t i = base + lane;
			const int in = i < n;
			gc_diag_t me;
			me.vd = ~0ULL, me.k = 0, me.len = 0, me.xo = 0, me.t = 0, me.pad_[0] = me.pad_[1] = 0;
			if (in) me = a[i];
			uint64_t nx = ~0ULL;
			if (lane == 0 && base + 64 < n) nx = a[base + 64].vd;
			nx = gc_shfl64(nx, 0);
			uint64_t rv = gc_shfl64(me.vd, lane < 63 ? lane + 1 : 63);
			if (lane == 63) rv = nx;
			const uint64_t mbnd = __ballot(in && rv != me.vd); /* a group ends at my cell */
			if (mbnd == 0) return GC_OK; /* one group fills the block: the generic routine (the list in B is sorted by now, which is all it needs) */
			const int32_t e = 63 - __clzll((long long)mbnd);
			const int act = lane <= e;
			int keep = act;
			for (int32_t d = 1; d < 64; ++d) { /* the first of the furthest cells of a group stays */
				const uint64_t lv = gc_shfl64(me.vd, lane >= d ? lane - d : lane), rv2 = gc_shfl64(me.vd, lane + d <= 63 ? lane + d : lane);
				const int32_t lk = __shfl(me.k, lane >= d ? lane - d : lane), rk = __shfl(me.k, lane + d <= 63 ? lane + d : lane);
				const int same_l = act && lane >= d && lv == me.vd, same_r = act && lane + d <= e && rv2 == me.vd;
				if (same_l && !(lk < me.k)) keep = 0;
				if (same_r && me.k < rk) keep = 0;
				if (!__ballot(same_l | same_r)) break;
			}
			if (keep && n_done > 0) { /* not on a finished diagonal: the intervals are disjoint and ascending */
				int32_t lo = 0, hi = n_done;
				while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (dn[mid].vd1 <= me.vd) lo = mid + 1; else hi = mid; }
				keep = !(lo < n_done && me.vd >= dn[lo].vd0);
			}
			const uint64_t mk = __ballot(keep);
			if (keep) { me.len = 0; tmp[m + gc_rank(mk)] = me; }
			m += gc_popc(mk);
			base += e + 1;
		}
		gc_sync();
		gc_diag_t *ta = B->a; const int32_t tm = B->m;
		B->a = z->ooo.a, B->m = z->ooo.m, B->n = m;
		z->ooo.a = ta, z->ooo.m = tm;
	}
	*handled = 1;
	return GC_OK;
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
/* The same dedup with the cells' KEYS in LDS (round 4): {vertex|diagonal, offset, where the cell lies in the wavefront} of up to GC_DD_CAP cells -- 16 bytes each -- are fetched
 * in one sweep (independent loads, several blocks in flight), every pass of the routine above then runs on LDS (stable partition, rank sort of the flagged keys, merge by
 * shifting the in-order keys up block by block from the top, furthest-of-group by looking at the neighbouring keys, finished-diagonal test against a copy of the list in LDS,
 * compaction), and the surviving cells are gathered from the wavefront into the scratch vector, which becomes the wavefront.  [measured] a dependent trip to HBM / L2 costs a
 * lone wavefront 500-1500 cycles, one to LDS ~100: the routine above took ~90 k cycles per call of a long bridge.  *handled = 0: more cells than the LDS block holds, or one of
 * the cases the routine above leaves to the generic one. */
#define GC_DD_CAP 640
typedef struct { uint64_t vd; int32_t k; uint32_t src; } gc_ddkey_t; /* src: index into the wavefront | flagged << 31 */
GC_HDN int gc_gw_dedup_lds(gc_arena_t *A, gc_gw_t *z, gc_diag_v *B, int *handled)
{
	__shared__ __attribute__((aligned(16))) gc_ddkey_t K[GC_DD_CAP];
	__shared__ __attribute__((aligned(16))) gc_ddkey_t C[2][64];
	__shared__ __attribute__((aligned(16))) gc_intv_t D[64];
	const int32_t n = B->n, lane = GC_LANE;
	*handled = 0;
	if (n > GC_DD_CAP) return GC_OK;
	GC_DDP(long long t_0 = clock64());
	GC_TRY(gc_vec_reserve(A, z->ooo, n > 0 ? n : 1));
	const gc_diag_t *a = B->a;
	gc_diag_t *tmp = z->ooo.a;
	const int32_t n_done = z->done.n;
	const int done_lds = n_done > 0 && n_done <= 64;
	if (done_lds && lane < n_done) D[lane] = z->done.a[lane];
	int unsorted = 0, b_unsorted = 0;
	int32_t n_c = 0;
	uint64_t carry_vd = 0, carry_b = 0;
	for (int32_t base0 = 0; base0 < n; base0 += 256) { /* ---- keys in: four blocks of loads in flight ---- */
		uint64_t vd[4]; int32_t kk[4]; uint32_t xo[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { const int32_t i = base0 + u * 64 + lane; vd[u] = ~0ULL, kk[u] = 0, xo[u] = 0; if (i < n) vd[u] = a[i].vd, kk[u] = a[i].k, xo[u] = a[i].xo; }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int32_t base = base0 + u * 64, i = base + lane;
			if (base < n) {
				const int in = i < n, flag = in && (xo[u] & 1);
				uint64_t pv = gc_shfl64(vd[u], lane > 0 ? lane - 1 : 0);
				if (lane == 0) pv = carry_vd;
				unsorted |= in && pv > vd[u];
				const uint64_t mc = __ballot(flag), mb = __ballot(in && !flag);
				const uint64_t lower = mb & ((1ULL << lane) - 1ULL);
				uint64_t pb = gc_shfl64(vd[u], lower ? 63 - __clzll((long long)lower) : lane); /* the in-order cell before mine */
				if (!lower) pb = carry_b;
				b_unsorted |= in && !flag && pb > vd[u];
				carry_vd = gc_shfl64(vd[u], n - base > 64 ? 63 : n - base - 1);
				if (mb) carry_b = gc_shfl64(vd[u], 63 - __clzll((long long)mb));
				if (in) { gc_ddkey_t q; q.vd = vd[u], q.k = kk[u], q.src = (uint32_t)i | (uint32_t)flag << 31; K[i] = q; }
				n_c += gc_popc(mc);
			}
		}
	}
	gc_sync();
	GC_DDP(long long t_1 = clock64());
	const int sorted_now = __ballot(unsorted) != 0;
	if (sorted_now) {
		if (n_c > 64 || __ballot(b_unsorted)) return GC_OK; /* the generic routine */
		const int32_t n_b = n - n_c;
		{ /* stable partition in LDS: the in-order keys close ranks (downwards: a block is read completely before it is written, and it is written at or below where it was) */
			int32_t kb = 0, kc = 0;
			for (int32_t base = 0; base < n; base += 64) {
				const int32_t i = base + lane;
				gc_ddkey_t q;
				q.vd = 0, q.k = 0, q.src = 0;
				if (i < n) q = K[i];
				const int in = i < n, flag = in && (q.src >> 31);
				const uint64_t mc = __ballot(flag), mb = __ballot(in && !flag);
				gc_sync();
				if (flag) C[0][kc + gc_rank(mc)] = q; else if (in) K[kb + gc_rank(mb)] = q;
				kb += gc_popc(mb), kc += gc_popc(mc);
				gc_sync();
			}
		}
		gc_ddkey_t c;
		uint64_t key = ~0ULL;
		c.vd = ~0ULL, c.k = 0, c.src = 0;
		if (lane < n_c) c = C[0][lane], key = c.vd;
		int32_t r = 0;
		for (int32_t j = 0; j < n_c; ++j) { const uint64_t kj = gc_shfl64(key, j); r += (kj < key) | ((kj == key) & (j < lane)); } /* keys smaller than mine, and equal ones before mine */
		if (lane < n_c) C[1][r] = c;
		gc_sync();
		uint64_t ck = ~0ULL;
		int32_t c_pos = 0;
		if (lane < n_c) { /* the sorted flagged keys: mine goes behind the in-order keys that are smaller or equal */
			c = C[1][lane], ck = c.vd;
			int32_t lo = 0, hi = n_b;
			while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (K[m].vd <= ck) lo = m + 1; else hi = m; }
			c_pos = lane + lo;
		}
		for (int32_t base = (n_b - 1) & ~63; base >= 0; base -= 64) { /* an in-order key goes up by the number of flagged keys that are strictly smaller: blocks from the top down */
			const int32_t i = base + lane;
			gc_ddkey_t q;
			q.vd = 0, q.k = 0, q.src = 0;
			if (i < n_b) q = K[i];
			int32_t lo = 0;
			for (int32_t j = 0; j < n_c; ++j) lo += gc_shfl64(ck, j) < q.vd;
			gc_sync();
			if (i < n_b) K[i + lo] = q;
			gc_sync();
		}
		if (lane < n_c) K[c_pos] = c;
		gc_sync();
	}
	GC_DDP(long long t_2 = clock64());
	int32_t m = 0;
	for (int32_t base = 0; base < n; base += 64) { /* ---- the first of the furthest cells of every (vertex, diagonal) that is not finished; survivors close ranks ---- */
		const int32_t i = base + lane;
		gc_ddkey_t q;
		q.vd = 0, q.k = 0, q.src = 0;
		int keep = 0;
		if (i < n) {
			q = K[i];
			keep = 1;
			for (int32_t j = i - 1; j >= 0 && K[j].vd == q.vd; --j) if (!(K[j].k < q.k)) { keep = 0; break; } /* an earlier cell at least as far */
			if (keep) for (int32_t j = i + 1; j < n && K[j].vd == q.vd; ++j) if (q.k < K[j].k) { keep = 0; break; } /* a later cell strictly further */
			if (keep && n_done > 0) {
				int32_t lo = 0, hi = n_done;
				if (done_lds) { while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (D[mid].vd1 <= q.vd) lo = mid + 1; else hi = mid; } keep = !(lo < n_done && q.vd >= D[lo].vd0); }
				else { const gc_intv_t *dn = z->done.a; while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (dn[mid].vd1 <= q.vd) lo = mid + 1; else hi = mid; } keep = !(lo < n_done && q.vd >= dn[lo].vd0); }
			}
		}
		const uint64_t mk = __ballot(keep);
		gc_sync(); /* (every lane has read its neighbours' keys before a block is written: the writes land at or below the block) */
		if (keep) K[m + gc_rank(mk)] = q;
		m += gc_popc(mk);
		gc_sync();
	}
	GC_DDP(long long t_3 = clock64());
	for (int32_t base0 = 0; base0 < m; base0 += 256) { /* ---- the survivors' cells: wavefront -> scratch vector, four blocks in flight ---- */
		uint4 c0[4], c1[4]; /* a cell as two 16-byte words: {vd, k, len} {xo, t, -, -} */
#pragma unroll
		for (int u = 0; u < 4; ++u) { const int32_t p = base0 + u * 64 + lane; c0[u] = make_uint4(0, 0, 0, 0), c1[u] = c0[u]; if (p < m) { const uint4 *src = (const uint4*)&a[K[p].src & 0x7fffffffU]; c0[u] = src[0], c1[u] = src[1]; } }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const int32_t p = base0 + u * 64 + lane; if (p < m) { uint4 *dst = (uint4*)&tmp[p]; c0[u].w = 0; if (sorted_now) c1[u].x &= 0xfffffffeU; dst[0] = c0[u], dst[1] = c1[u]; } }
	}
	gc_sync();
	{
		gc_diag_t *ta = B->a; const int32_t tm = B->m;
		B->a = z->ooo.a, B->m = z->ooo.m, B->n = m;
		z->ooo.a = ta, z->ooo.m = tm;
	}
	GC_DDP(long long t_4 = clock64(); GC_COUNT(A, 1, t_1 - t_0); GC_COUNT(A, 2, t_2 - t_1); GC_COUNT(A, 3, t_3 - t_2); GC_COUNT(A, 4, t_4 - t_3));
	*handled = 1;
	return GC_OK;
}
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
/* gwf_dedup on ONE lane (the host instantiation; round 4): the generic routine below is written for 64 lanes -- partition, rank sort, merge by binary search, two verdict
 * passes and two compactions, a binary search per cell into the finished ranges -- which a single thread pays as half a dozen sweeps over the wavefront ([measured,
 * GC_HOST_PROF] 44 % of a bridge's cycles, next to 48 % for the runs).  Same result in two sweeps: the flagged cells (a handful: what the head cells pushed) are pulled out,
 * insertion-sorted (klib's own method up to 64 records: stable) and merged back from the top (ties: the in-order cell first); then one sweep keeps the first of the furthest
 * cells of every (vertex, diagonal) while a second cursor walks the finished ranges alongside (both lists ascend).  *handled = 0: more than 64 flagged cells or in-order cells
 * out of order -- the generic routine, on the untouched list. */
GC_HD int gc_gw_dedup_lane(gc_arena_t *A, gc_gw_t *z, int32_t *n_a_, gc_diag_t *a, int *handled)
{
	const int32_t n = *n_a_;
	int32_t n_c = 0, unsorted = 0, b_unsorted = 0, have_b = 0;
	uint64_t last_b = 0;
	*handled = 0;
	for (int32_t i = 0; i < n; ++i) {
		if (i && a[i - 1].vd > a[i].vd) unsorted = 1;
		if (a[i].xo & 1) ++n_c;
		else { if (have_b && last_b > a[i].vd) b_unsorted = 1; last_b = a[i].vd, have_b = 1; }
	}
	if (unsorted) {
		if (n_c > 64 || b_unsorted) return GC_OK;
		GC_TRY(gc_vec_reserve(A, z->ooo, n_c > 0 ? n_c : 1));
		gc_diag_t *c = z->ooo.a;
		int32_t nb = 0, nc = 0;
		for (int32_t i = 0; i < n; ++i) { if (a[i].xo & 1) c[nc++] = a[i]; else { if (nb != i) a[nb] = a[i]; ++nb; } }
		for (int32_t i = 1; i < nc; ++i) { /* insertion sort: stable */
			const gc_diag_t t = c[i];
			int32_t j = i;
			for (; j > 0 && t.vd < c[j - 1].vd; --j) c[j] = c[j - 1];
			c[j] = t;
		}
		for (int32_t j = 0; j < nc; ++j) c[j].xo &= 0xfffffffeU;
		for (int32_t i = nb - 1, j = nc - 1, k = n - 1; j >= 0; --k) { /* merge from the top; an in-order cell goes in FRONT of an equal flagged one */
			if (i >= 0 && a[i].vd > c[j].vd) a[k] = a[i--]; else a[k] = c[j--];
		}
	}
	const gc_intv_t *dn = z->done.a;
	const int32_t nd = z->done.n;
	int32_t di = 0, m = 0;
	for (int32_t i = 0; i < n;) {
		const uint64_t vd = a[i].vd;
		int32_t best = i, j = i + 1;
		for (; j < n && a[j].vd == vd; ++j) if (a[best].k < a[j].k) best = j; /* the FIRST of the furthest */
		while (di < nd && dn[di].vd1 <= vd) ++di;
		if (!(di < nd && vd >= dn[di].vd0)) { gc_diag_t t = a[best]; t.len = 0; a[m++] = t; }
		i = j;
	}
	*n_a_ = m, *handled = 1;
	return GC_OK;
}
#endif
GC_HD int gc_gw_dedup(gc_arena_t *A, gc_gw_t *z, int32_t *n_a_, gc_diag_t *a) /* gwf_dedup, gfa-ed.c:258-271 */
{
	int32_t n_a = *n_a_;
	for (int32_t i = 0; i < z->fresh.n; ++i) GC_TRY(gc_intv_add(A, &z->done, z->fresh.a[i].vd0, z->fresh.a[i].vd1)); /* this step's finished diagonals */
	GC_STAT(gc_stats.dedup_calls++; { int nc_ = 0, uns_ = 0; for (int32_t i = 0; i < n_a; ++i) { nc_ += a[i].xo & 1; if (i && a[i - 1].vd > a[i].vd) uns_ = 1; } gc_stats.dd_sum_n += n_a, gc_stats.dd_sum_fresh += z->fresh.n, gc_stats.dd_sum_nc += nc_, gc_stats.dd_unsorted += uns_;
		gc_stats.dd_n[gc_stats_bin(n_a)]++, gc_stats.dd_fresh[gc_stats_bin(z->fresh.n)]++, gc_stats.dd_nc[gc_stats_bin(nc_)]++; })
	GC_STAT(gc_stats.dd_sum_done += z->done.n; gc_stats.dd_done[gc_stats_bin(z->done.n)]++;)
	{
		int unsorted = 0;
		GC_PAR_FOR(i, n_a) if (i > 0 && a[i - 1].vd > a[i].vd) unsorted = 1;
		if (gc_ballot(unsorted)) { GC_COUNT(A, 10, 1); GC_TRY(gc_diag_sort(A, z, n_a, a)); }
	}
	/* keep the furthest cell of every (vertex, diagonal): the first of equals.  Groups are a handful of cells: every lane looks left and right
	 * of its own cell inside the group and leaves its verdict in the cell's (otherwise unused here) len field; then the survivors close ranks */
	GC_PAR_FOR(i, n_a) {
		const uint64_t vd = a[i].vd;
		const int32_t k = a[i].k;
		int keep = 1;
		for (int32_t j = i - 1; j >= 0 && a[j].vd == vd; --j) if (!(a[j].k < k)) { keep = 0; break; } /* an earlier cell at least as far */
		if (keep) for (int32_t j = i + 1; j < n_a && a[j].vd == vd; ++j) if (k < a[j].k) { keep = 0; break; } /* a later cell strictly further */
		a[i].len = keep;
	}
	gc_sync();
	GC_COMPACT_BEGIN(n_a)
		gc_diag_t me;
		int keep = 0;
		if (gcin_) { me = a[gci_]; keep = me.len; me.len = 0; }
	GC_COMPACT_END(a, me, keep, n_a)
	if (z->done.n > 0) { /* drop cells on finished diagonals (gfa-ed.c:192-202): the intervals are disjoint and ascending */
		const int32_t n_b = z->done.n;
		const gc_intv_t *b = z->done.a;
		GC_COMPACT_BEGIN(n_a)
			gc_diag_t me;
			int keep = 0;
			if (gcin_) {
				me = a[gci_];
				int32_t lo = 0, hi = n_b; /* first interval whose end lies beyond the cell */
				while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (b[m].vd1 <= me.vd) lo = m + 1; else hi = m; }
				keep = !(lo < n_b && me.vd >= b[lo].vd0);
			}
		GC_COMPACT_END(a, me, keep, n_a)
	}
	*n_a_ = n_a;
	return GC_OK;
}
GC_HDN int32_t gc_gw_prune(int32_t n_a, gc_diag_t *a, uint32_t max_lag, int32_t bw_dyn) /* gfa-ed.c:286-307 */
{
	int32_t max_i = -1, j = 0;
	uint32_t max_x = 0;
	for (int32_t i = 0; i < n_a; ++i) if (a[i].xo >> 1 > max_x) max_x = a[i].xo >> 1, max_i = i;
	const int32_t iq = (int32_t)a[max_i].vd - GC_DSHIFT + a[max_i].k, dq = (int32_t)(a[max_i].xo >> 1) - iq - iq;
	for (int32_t i = 0; i < n_a; ++i) {
		const gc_diag_t p = a[i];
		const int32_t ip = (int32_t)p.vd - GC_DSHIFT + p.k, dp = (int32_t)(p.xo >> 1) - ip - ip, w = dp > dq ? dp - dq : dq - dp;
		if (bw_dyn >= 0 && w > bw_dyn) continue;
		if ((p.xo >> 1) + max_lag < max_x) continue;
		a[j++] = p;
	}
	return j;
}
/* Landau-Vishkin over a run of n adjacent diagonals on one vertex (gfa-ed.c:331-403): every diagonal is extended along its matches, then
 * the next wavefront takes, per diagonal, the furthest of {insertion from the left neighbour, mismatch, deletion from the right neighbour};
 * cells that reached the end of the vertex or of the query go to H (handled one by one by the caller).  One lane per diagonal. */
GC_HDN int gc_gw_extend_run(gc_arena_t *A, gc_gw_t *z, int32_t n, gc_diag_t *a, gc_diag_v *B, gc_diag_v *H)
{
	const uint32_t v = (uint32_t)(a->vd >> 32);
	const int32_t vl = gc_vlen(z->G, v);
	const char *ts = gc_vseq(z->G, v);
	GC_TRY(gc_vec_reserve(A, *B, B->n + n + 2));
	GC_TRY(gc_vec_reserve(A, *H, H->n + n));
	GC_TRY(gc_vec_reserve(A, z->fresh, z->fresh.n + n + 2));
#if defined(__HIP_DEVICE_COMPILE__)
	if (n <= 64) { /* the usual case, a run fits the wavefront: a lane keeps its diagonal in registers from the extension to the compacted output, the
	                * neighbours' cells come by lane shuffles, places in H / B / the finished list by ballots -- one pass over memory, one fence */
		const int32_t j = GC_LANE;
		const int in = j < n;
		gc_diag_t me;
		me.vd = 0, me.k = 0, me.len = 0, me.xo = 0, me.t = 0, me.pad_[0] = me.pad_[1] = 0;
		if (in) {
			me = a[j];
			const int32_t k2 = gc_extend1((int32_t)me.vd - GC_DSHIFT, me.k, vl, ts, z->ql, z->q);
			me.len = k2 - me.k, me.xo += (uint32_t)me.len << 2, me.k = k2;
		}
		const int32_t kL = __shfl_up(me.k, 1), tL = __shfl_up(me.t, 1), kR = __shfl_down(me.k, 1), tR = __shfl_down(me.t, 1);
		const uint32_t xL = __shfl_up(me.xo, 1), xR = __shfl_down(me.xo, 1);
		/* the cell of my diagonal in the next wavefront */
		uint32_t mx = me.xo + 4;
		int32_t mk = me.k + 1, mt = me.t;
		if (in && j > 0 && kL > me.k + 1) mx = xL + 2, mk = kL, mt = tL;
		if (in && j + 1 < n && !(mk > kR + 1)) mx = xR + 2, mt = tR, mk = kR + 1;
		const int32_t d = (int32_t)me.vd - GC_DSHIFT;
		/* cells at a vertex / query end go to H, flagged, in diagonal order */
		{
			const int at_end = in && (me.k == vl - 1 || d + me.k == z->ql - 1);
			const uint64_t m = gc_ballot(at_end);
			if (at_end) { gc_diag_t h = me; h.xo |= 1; H->a[H->n + gc_rank(m)] = h; }
			H->n += gc_popc(m);
		}
		/* output order: the cell left of the run (lane 0), the run's cells, the cell right of it (lane n - 1) */
		const int is0 = in && j == 0, isN = in && j == n - 1;
		const int32_t k0 = me.k + 1, kN = me.k; /* left edge: vd - 1, xo + 2, k + 1; right edge: vd + 1, xo + 2, k */
		const int keep0 = is0 && (d - 1) + k0 < z->ql && k0 < vl, fin0 = is0 && !keep0 && k0 == vl;
		const int keepM = in && d + mk < z->ql && mk < vl, finM = in && !keepM && mk == vl;
		const int keepN = isN && (d + 1) + kN < z->ql && kN < vl, finN = isN && !keepN && kN == vl;
		const uint64_t b0 = gc_ballot(keep0), bM = gc_ballot(keepM), bN = gc_ballot(keepN), f0 = gc_ballot(fin0), fM = gc_ballot(finM), fN = gc_ballot(finN);
		gc_diag_t *b = &B->a[B->n];
		gc_intv_t *fr = &z->fresh.a[z->fresh.n];
		if (keep0) { gc_diag_t c; c.vd = me.vd - 1, c.k = k0, c.len = 0, c.xo = me.xo + 2, c.t = me.t; b[0] = c; }
		if (keepM) { gc_diag_t c; c.vd = me.vd, c.k = mk, c.len = 0, c.xo = mx, c.t = mt; b[gc_popc(b0) + gc_rank(bM)] = c; }
		if (keepN) { gc_diag_t c; c.vd = me.vd + 1, c.k = kN, c.len = 0, c.xo = me.xo + 2, c.t = me.t; b[gc_popc(b0) + gc_popc(bM)] = c; }
		if (fin0) { fr[0].vd0 = gc_mk_vd(v, d - 1), fr[0].vd1 = fr[0].vd0 + 1; }
		if (finM) { gc_intv_t *iv = &fr[gc_popc(f0) + gc_rank(fM)]; iv->vd0 = gc_mk_vd(v, d), iv->vd1 = iv->vd0 + 1; }
		if (finN) { gc_intv_t *iv = &fr[gc_popc(f0) + gc_popc(fM)]; iv->vd0 = gc_mk_vd(v, d + 1), iv->vd1 = iv->vd0 + 1; }
		B->n += gc_popc(b0) + gc_popc(bM) + gc_popc(bN);
		z->fresh.n += gc_popc(f0) + gc_popc(fM) + gc_popc(fN);
		gc_sync();
		return GC_OK;
	}
#endif
	GC_PAR_FOR(j, n) {
		const int32_t k = gc_extend1((int32_t)a[j].vd - GC_DSHIFT, a[j].k, vl, ts, z->ql, z->q);
		a[j].len = k - a[j].k, a[j].xo += (uint32_t)(k - a[j].k) << 2, a[j].k = k;
	}
	gc_sync();
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(GC_AB_NO_LANE_RUNS)
	{ /* ONE lane (round 4): what the three lane-parallel passes below leave -- the run's cells at a vertex / query end to H (flagged), the next wavefront's cells of the run and of
	   * the two diagonals beside it to B, the diagonals that ran off the vertex to the finished list, each in diagonal order -- in one sweep over the extended cells */
		gc_diag_t *bo = &B->a[B->n], *ho = &H->a[H->n];
		gc_intv_t *fo = &z->fresh.a[z->fresh.n];
		int32_t nb = 0, nh = 0, nf = 0;
		const int32_t ql = z->ql;
#define GC_RUN_EMIT(vd_, k_, x_, t_) do { const uint64_t evd_ = (vd_); const int32_t ek_ = (k_), ed_ = (int32_t)evd_ - GC_DSHIFT; \
			if (ed_ + ek_ < ql && ek_ < vl) { gc_diag_t *o_ = &bo[nb++]; o_->vd = evd_, o_->k = ek_, o_->len = 0, o_->xo = (x_), o_->t = (t_), o_->pad_[0] = o_->pad_[1] = 0; } \
			else if (ek_ == vl) { fo[nf].vd0 = evd_, fo[nf].vd1 = evd_ + 1, ++nf; } } while (0)
		GC_RUN_EMIT(a[0].vd - 1, a[0].k + 1, a[0].xo + 2, a[0].t);
		for (int32_t j = 0; j < n; ++j) {
			uint32_t x;
			int32_t k, t;
			if (j > 0 && a[j - 1].k > a[j].k + 1) x = a[j - 1].xo + 2, k = a[j - 1].k, t = a[j - 1].t; /* insertion from the left neighbour ... */
			else x = a[j].xo + 4, t = a[j].t, k = a[j].k + 1;                                      /* ... unless the mismatch gets at least as far */
			if (j + 1 < n && !(k > a[j + 1].k + 1)) x = a[j + 1].xo + 2, t = a[j + 1].t, k = a[j + 1].k + 1; /* deletion from the right neighbour, when it gets at least as far */
			GC_RUN_EMIT(a[j].vd, k, x, t);
		}
		GC_RUN_EMIT(a[n - 1].vd + 1, a[n - 1].k, a[n - 1].xo + 2, a[n - 1].t);
#undef GC_RUN_EMIT
		for (int32_t j = 0; j < n; ++j) /* (behind the sweep above, which reads the cells' flag bits as they came) */
			if (a[j].k == vl - 1 || (int32_t)a[j].vd - GC_DSHIFT + a[j].k == ql - 1) { a[j].xo |= 1; ho[nh++] = a[j]; }
		B->n += nb, H->n += nh, z->fresh.n += nf;
		return GC_OK;
	}
#endif
	gc_diag_t *b = &B->a[B->n];
	GC_PAR_FOR(j, n) { /* b[j + 1]: the cell of diagonal a[j].vd in the next wavefront */
		uint32_t x;
		int32_t k, t;
		if (j > 0) {
			x = a[j - 1].xo + 2, k = a[j - 1].k, t = a[j - 1].t;
			if (!(k > a[j].k + 1)) x = a[j].xo + 4, t = a[j].t, k = a[j].k + 1;
		} else x = a[0].xo + 4, t = a[0].t, k = a[0].k + 1;
		if (j + 1 < n && !(k > a[j + 1].k + 1)) x = a[j + 1].xo + 2, t = a[j + 1].t, k = a[j + 1].k + 1;
		b[j + 1].vd = a[j].vd, b[j + 1].k = k, b[j + 1].xo = x, b[j + 1].t = t, b[j + 1].len = 0;
	}
	b[0].vd = a[0].vd - 1, b[0].xo = a[0].xo + 2, b[0].k = a[0].k + 1, b[0].t = a[0].t, b[0].len = 0;
	b[n + 1].vd = a[n - 1].vd + 1, b[n + 1].xo = a[n - 1].xo + 2, b[n + 1].t = a[n - 1].t, b[n + 1].k = a[n - 1].k, b[n + 1].len = 0;
	gc_sync();
	for (int32_t base = 0; base < n; base += GC_NLANE) { /* cells at a vertex / query end, in diagonal order */
		const int32_t j = base + GC_LANE;
		const int at_end = j < n && (a[j].k == vl - 1 || (int32_t)a[j].vd - GC_DSHIFT + a[j].k == z->ql - 1);
		const uint64_t m = gc_ballot(at_end);
		if (at_end) { a[j].xo |= 1; H->a[H->n + gc_rank(m)] = a[j]; }
		H->n += gc_popc(m);
	}
	gc_sync();
	int32_t n_keep = 0;
	for (int32_t base = 0; base < n + 2; base += GC_NLANE) { /* cells that left the vertex or the query are dropped; a diagonal that ran off the vertex end is finished */
		const int32_t j = base + GC_LANE;
		gc_diag_t p;
		int keep = 0, fin = 0;
		if (j < n + 2) {
			p = b[j];
			const int32_t d = (int32_t)p.vd - GC_DSHIFT;
			keep = d + p.k < z->ql && p.k < vl;
			fin = !keep && p.k == vl;
		}
		const uint64_t mk = gc_ballot(keep), mf = gc_ballot(fin);
		gc_sync();
		if (keep) b[n_keep + gc_rank(mk)] = p;
		if (fin) { gc_intv_t *iv = &z->fresh.a[z->fresh.n + gc_rank(mf)]; iv->vd0 = gc_mk_vd(v, (int32_t)p.vd - GC_DSHIFT), iv->vd1 = iv->vd0 + 1; }
		n_keep += gc_popc(mk), z->fresh.n += gc_popc(mf);
		gc_sync();
	}
	B->n += n_keep;
	return GC_OK;
}
/* one edit-distance step: consumes wf[cur], builds wf[cur^1]; *reached = 1 when (v1, off1) was hit (gfa-ed.c:405-507) */
GC_HD int gc_gw_step(gc_arena_t *A, gc_gw_t *z, uint32_t v1, int32_t off1, int *reached)
{
	const gc_graph_t *G = z->G;
	gc_diag_v *Cur = &z->wf[z->cur], *B = &z->wf[z->cur ^ 1], *H = &z->head;
	gc_diag_t *a = Cur->a;
	int32_t n = Cur->n, head = 0, do_dedup = 1;
	*reached = 0;
	GC_STAT(int st_runs = 0; int st_long = 0;)
	H->n = B->n = 0;
	z->end_v = (uint32_t)-1, z->end_off = z->end_tb = -1;
	z->fresh.n = 0;
	GC_TICK(A, 8);
	gc_u64map_clear(&z->seen);
	GC_TRY(gc_vec_reserve(A, *B, n * 2 + 4));
	GC_TICK(A, 11);
	{ /* runs of adjacent diagonals on one vertex: the lanes look for the run ends, the runs are then taken in order */
		int32_t x = 0;
		for (int32_t base = 0; base < n; base += GC_NLANE) {
			const int32_t i = base + GC_LANE + 1; /* a run ends in front of position i */
			uint64_t m = gc_ballot(i <= n && (i == n || a[i].vd != a[i - 1].vd + 1));
			while (m) {
#if defined(__HIP_DEVICE_COMPILE__)
				const int32_t bit = (int32_t)__ffsll((long long)m) - 1;
#else
				const int32_t bit = 0;
#endif
				const int32_t e = base + bit + 1;
				m &= m - 1;
				GC_TRY(gc_gw_extend_run(A, z, e - x, &Cur->a[x], B, H));
				GC_DDP(if (0)) GC_COUNT(A, 3, 1);
				GC_STAT(++st_runs; if (e - x > 64) st_long = 1;)
				x = e;
			}
		}
	}
	if (H->n == 0) do_dedup = 0;
	GC_DDP(if (0)) { GC_COUNT(A, 1, 1); GC_COUNT(A, 2, n); GC_COUNT(A, 4, H->n); }
	GC_STAT(gc_stats.steps++; gc_stats.cells += n; gc_stats.runs += st_runs; gc_stats.heads0 += H->n; gc_stats.single += st_runs == 1; gc_stats.nohead += H->n == 0; gc_stats.single_nohead += st_runs == 1 && H->n == 0;
			if (st_runs == 1 && H->n == 0) gc_stats.simple_cells += n; gc_stats.le64 += !st_long; gc_stats.dedup_steps += do_dedup; gc_stats.n_hist[gc_stats_bin(n)]++;)
	GC_TICK(A, 12);
	while (head < H->n) {
		const gc_diag_t t = H->a[head++];
		const uint32_t v = (uint32_t)(t.vd >> 32), ooo = t.xo & 1;
		const int32_t d = (int32_t)t.vd - GC_DSHIFT, vl = gc_vlen(G, v);
		const int32_t k = gc_extend1(d, t.k, vl, gc_vseq(G, v), z->ql, z->q), qi = k + d;
		const uint32_t x0 = (t.xo >> 1) + ((uint32_t)(k - t.k) << 1);
		if (k + 1 < vl && qi + 1 < z->ql) { /* inside a vertex */
			GC_STAT(gc_stats.hd_in++;)
			int push1 = 1, push2 = 1;
			if (B->n >= 2) push1 = gc_diag_update(&B->a[B->n - 2], v, d - 1, k + 1, x0 + 1, ooo, t.t);
			if (B->n >= 1) push2 = gc_diag_update(&B->a[B->n - 1], v, d, k + 1, x0 + 2, ooo, t.t);
			if (push1) GC_TRY(gc_diag_push(A, B, v, d - 1, k + 1, x0 + 1, 1, t.t));
			if (push2 || push1) GC_TRY(gc_diag_push(A, B, v, d, k + 1, x0 + 2, 1, t.t));
			GC_TRY(gc_diag_push(A, B, v, d + 1, k, x0 + 1, ooo, t.t));
		} else if (qi + 1 < z->ql) { /* end of the vertex, query not finished: fan out over the arcs */
			GC_STAT(gc_stats.hd_arc++;)
			const int32_t nv = gc_n_arc(G, v);
			const gc_arc_t *av = gc_arcs(G, v);
			int32_t n_ext = 0, tw;
			gc_intv_t *iv;
			GC_PUSH(A, z->fresh, iv);
			iv->vd0 = gc_mk_vd(v, d), iv->vd1 = iv->vd0 + 1;
			GC_TRY(gc_trace_push(A, z, (int32_t)v, t.t, &tw));
			for (int32_t j = 0; j < nv; ++j) {
				const uint32_t w = av[j].w;
				const int32_t ol = av[j].ow;
				int absent;
				int32_t *dummy;
				GC_TRY(gc_u64map_put(A, &z->seen, (uint64_t)w << 32 | (uint32_t)(qi + 1), &dummy, &absent));
				if (z->q[qi + 1] == gc_vseq(G, w)[ol]) {
					++n_ext;
					if (absent) {
						gc_diag_t *p;
						GC_PUSH(A, *H, p);
						p->vd = gc_mk_vd(w, qi + 1 - ol), p->k = ol, p->xo = (x0 + 2) << 1 | 1, p->t = tw, p->len = 0;
					}
				} else if (absent) {
					GC_TRY(gc_diag_push(A, B, w, qi - ol, ol, x0 + 1, 1, tw));
					GC_TRY(gc_diag_push(A, B, w, qi + 1 - ol, ol, x0 + 2, 1, tw));
				}
			}
			if (nv == 0 || n_ext != nv) GC_TRY(gc_diag_push(A, B, v, d + 1, k, x0 + 1, 1, t.t));
		} else if (v1 == (uint32_t)-1 || (v == v1 && k == off1)) { /* query finished at the requested end */
			z->end_v = v, z->end_off = k, z->end_tb = t.t;
			B->n = 0;
			*reached = 1;
			return GC_OK;
		} else if (k + 1 < vl) { /* query finished inside a vertex: delete the next target base */
			GC_TRY(gc_diag_push(A, B, v, d - 1, k + 1, x0 + 1, ooo, t.t));
		} else if (v != v1) { /* query and vertex both finished, not the last vertex */
			const int32_t nv = gc_n_arc(G, v);
			const gc_arc_t *av = gc_arcs(G, v);
			int32_t tw;
			GC_TRY(gc_trace_push(A, z, (int32_t)v, t.t, &tw));
			for (int32_t j = 0; j < nv; ++j) GC_TRY(gc_diag_push(A, B, av[j].w, qi - av[j].ow, av[j].ow, x0 + 1, 1, tw));
		}
	}
	GC_TICK(A, 13);
	GC_COUNT(A, 6, B->n); GC_COUNT(A, 9, H->n);
	GC_STAT(gc_stats.heads += H->n; gc_stats.out += B->n;)
	if (do_dedup) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GC_AB_NO_WAVE_DEDUP)
		int handled = 0;
		GC_COUNT(A, 7, z->done.n); GC_COUNT(A, 15, z->fresh.n);
		GC_TRY(gc_intv_add_wave(A, &z->done, z->fresh.n, z->fresh.a)); /* this step's finished diagonals */
		z->fresh.n = 0;
		GC_TICK(A, 0);
		GC_TRY(gc_gw_dedup_lds(A, z, B, &handled));
		GC_COUNT(A, 10, handled ? 1 << 20 : 0); /* (profiling: high part of slot 10 = dedups done on LDS) */
		if (!handled) GC_TRY(gc_gw_dedup_wave(A, z, B, &handled));
		if (!handled)
#elif !defined(__HIP_DEVICE_COMPILE__) && !defined(GC_AB_NO_LANE_DEDUP)
		int handled = 0;
		for (int32_t i = 0; i < z->fresh.n; ++i) GC_TRY(gc_intv_add(A, &z->done, z->fresh.a[i].vd0, z->fresh.a[i].vd1)); /* this step's finished diagonals */
		z->fresh.n = 0;
		GC_TRY(gc_gw_dedup_lane(A, z, &B->n, B->a, &handled));
		if (!handled)
#endif
		GC_TRY(gc_gw_dedup(A, z, &B->n, B->a));
	}
	GC_TICK(A, 14);
	if (z->max_lag > 0 && B->n > z->max_chk && ((z->s + 1) & 0xf) == 0) B->n = gc_gw_prune(B->n, B->a, (uint32_t)z->max_lag, z->bw_dyn);
	z->cur ^= 1;
	return GC_OK;
}
/* unit-cost edit distance of q[0..ql) against the walks from (v0, off0) that end at (v1, off1); *ed = -1 when not reached within
 * s_term edits.  The vertex walk goes to a vector in the arena (gfa_ed_init/step, gfa-ed.c:524-617, as gchain1.c:349-381 calls them). */
GC_HDN int gc_gwfa(gc_arena_t *A, const gc_graph_t *G, int32_t ql, const char *q, uint32_t v0, int32_t off0, uint32_t v1, int32_t off1,
				  int32_t max_lag, int32_t s_term, int32_t *ed, int32_t **path, int32_t *n_path)
{
	GC_STATE(gc_gw_t, z);
	memset(&z, 0, sizeof z);
	*ed = -1, *path = 0, *n_path = 0;
	z.G = G, z.ql = ql, z.q = q;
	z.max_chk = 1000, z.bw_dyn = 1000, z.max_lag = max_lag, z.i_term = 500000000LL; /* gchain1.c:361-363 */
	GC_TRY(gc_vec_reserve(A, z.wf[0], 16));
	memset(&z.wf[0].a[0], 0, sizeof(gc_diag_t));
	z.wf[0].a[0].vd = gc_mk_vd(v0, -off0), z.wf[0].a[0].k = off0 - 1, z.wf[0].a[0].xo = 0, z.wf[0].a[0].t = 0;
	z.wf[0].n = 1;
	{ gc_trace_t *t; GC_PUSH(A, z.tr, t); t->v = -1, t->pre = -1; } /* root of the traceback forest (gfa-ed.c:568) */
	z.end_v = (uint32_t)-1, z.end_off = -1;
	int64_t n_iter = 0;
	while (z.wf[z.cur].n > 0) {
		int reached;
		GC_TRY(gc_gw_step(A, &z, v1, off1, &reached));
		n_iter += z.wf[z.cur].n;
		if (reached || z.end_off >= 0 || z.wf[z.cur].n == 0) break;
		if (s_term >= 0 && z.s >= s_term) break;
		if (z.i_term > 0 && n_iter > z.i_term) break;
		++z.s;
	}
	if (z.end_off >= 0) { /* gwf_traceback, gfa-ed.c:509-522 */
		int32_t i = z.end_tb, n = 1, *p;
		while (i >= 0 && z.tr.a[i].v >= 0) ++n, i = z.tr.a[i].pre;
		GC_ALLOC(A, int32_t, p, n);
		i = z.end_tb, n = 0;
		p[n++] = (int32_t)z.end_v;
		while (i >= 0 && z.tr.a[i].v >= 0) p[n++] = z.tr.a[i].v, i = z.tr.a[i].pre;
		for (i = 0; i < n >> 1; ++i) { const int32_t k = p[i]; p[i] = p[n - 1 - i], p[n - 1 - i] = k; }
		*path = p, *n_path = n;
	}
	*ed = z.end_v != (uint32_t)-1 ? z.s : -1;
	GC_STAT(gc_stats.calls++; gc_stats.sum_ql += ql; gc_stats.reached += *ed >= 0; gc_stats.steps_hist[gc_stats_bin(z.s + 1)]++;)
	return GC_OK;
}

/* ------------------------------------------------------------------------------------------------ walk assembly */

typedef struct {
	GC_VEC(mg_llchain_t) lc;   /* vertices of all graph chains, in walk order; anchor-free vertices have cnt == 0 */
	GC_F int32_t n_a;               /* anchors copied to the output so far */
	GC_F mg128_t *a_out;
	GC_F int32_t n_gwfa; GC_F int32_t n_shortk; GC_F int32_t n_fast;
} gc_asm_t;

GC_HD int gc_asm_vertex(gc_arena_t *A, gc_asm_t *S, uint32_t v)
{
	mg_llchain_t *q;
	GC_PUSH(A, S->lc, q);
	q->off = q->cnt = q->score = 0, q->v = v, q->ed = -1;
	return GC_OK;
}
GC_HD int gc_asm_chain(gc_arena_t *A, gc_asm_t *S, const gc_chain_t *c, const mg128_t *a, int32_t ed)
{
	mg_llchain_t *q;
	GC_PUSH(A, S->lc, q);
	q->cnt = c->cnt, q->v = c->v, q->score = c->score, q->ed = ed, q->off = S->n_a;
	gc_pcopy(&S->a_out[S->n_a], &a[c->off], (int64_t)c->cnt * (int64_t)sizeof(mg128_t));
	S->n_a += c->cnt;
	return GC_OK;
}

/* two consecutive chains of a graph chain may share anchors at the junction: cut the tail of the first / the head of the second
 * back to where they are monotone on the query (and on the segment when they share it) (gchain1.c:409-441) */
GC_HD void gc_untangle(gc_chain_t *c0, gc_chain_t *c1, const mg128_t *a)
{
	int32_t j, x = GC_AX(a[c1->off]), y = GC_AY(a[c1->off]);
	const int same = c0->v == c1->v;
	for (j = c0->cnt - 1; j >= 0; --j)
		if (GC_AY(a[c0->off + j]) <= y && (!same || GC_AX(a[c0->off + j]) <= x)) break;
	const int32_t drop0 = c0->cnt - 1 - j;
	x = GC_AX(a[c0->off + c0->cnt - 1]), y = GC_AY(a[c0->off + c0->cnt - 1]);
	for (j = 0; j < c1->cnt; ++j)
		if (GC_AY(a[c1->off + j]) >= y && (!same || GC_AX(a[c1->off + j]) >= x)) break;
	const int32_t drop1 = j;
	if (drop0 > 0) {
		c0->cnt -= drop0;
		if (c0->cnt) c0->qe = GC_AY(a[c0->off + c0->cnt - 1]) + 1, c0->re = GC_AX(a[c0->off + c0->cnt - 1]) + 1;
	}
	if (drop1 > 0) {
		c1->off += drop1, c1->cnt -= drop1;
		c1->qs = GC_AY(a[c1->off]) + 1 - GC_ASPAN(a[c1->off]), c1->rs = GC_AX(a[c1->off]) + 1 - GC_ASPAN(a[c1->off]);
	}
	if (c0->cnt == 0) c0->qs = c0->qe = c1->qs, c0->rs = c0->re = c1->rs;
}

/* What a bridge between two chains on different segments comes to: the walk between them, or "no walk of the chosen length" (the reference's "chain skiped").  It depends on the two
 * chain records, the graph and the query only -- not on the assembly so far -- which is what lets the device compute a read's bridges on several wavefronts (gc_read_p1/p2/p3 below). */
typedef struct { int32_t failed, ed, n_mid; int32_t *mid; int32_t n_gwfa, n_shortk, n_fast; } gc_bres_t;

/* the vertices between chain c0 and chain c1 (different segments): by aligning the query between them to the graph, or, when that
 * gives up, by the shortest walk of the length the DP chose (gchain1.c:319-407).  b->mid lies in A (scratch: the caller resets A when it has used it) */
GC_HD int gc_bridge_walk(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, int32_t span, const gc_chain_t *c0, const gc_chain_t *c1, const char *qseq, gc_bres_t *b)
{
	int32_t ed = -1, *path = 0, n_path = 0, n_mid = 0;
	const int64_t mark = A->top;
	const char *base = A->base;
	b->failed = 0, b->ed = -1, b->n_mid = 0, b->mid = 0, b->n_gwfa = b->n_shortk = b->n_fast = 0;
	{
		const int32_t qs = c0->qe - span, qe = c1->qs + span;
		int rc = GC_E_ARENA;
		GC_TICK(A, 5);
		/* (the LDS scratch holds a call over a query gap of a few hundred bases -- [measured] 30 % of the calls of the bench workload fit 16 KB -- and a failed
		 * attempt is paid twice, so only short gaps try it) */
		if (A->fast_base && qe - qs <= GC_FAST_GAP) { gc_arena_t F; gc_arena_init(&F, A->fast_base, A->fast_cap, 0); F.ticks = A->ticks, F.tick_last = A->tick_last; rc = gc_gwfa(&F, G, qe - qs, qseq + qs, c0->v, c0->re - span, c1->v, c1->rs + span - 1, P->gdp_max_ed / 2, P->gdp_max_ed, &ed, &path, &n_path); A->tick_last = F.tick_last; if (rc == GC_OK) ++b->n_fast; }
		GC_STAT(const int64_t st_top0 = A->top; const int64_t st_peak0 = A->peak; A->peak = A->top;)
		if (rc == GC_E_ARENA) rc = gc_gwfa(A, G, qe - qs, qseq + qs, c0->v, c0->re - span, c1->v, c1->rs + span - 1, P->gdp_max_ed / 2, P->gdp_max_ed, &ed, &path, &n_path);
		GC_STAT(if (A->base == base) { const int64_t used = A->peak - st_top0; gc_stats.arena_hist[gc_stats_bin(used >> 9)]++; gc_stats.arena_sum += used; if (used <= 16384) gc_stats.fit16++; if (used <= 32768) gc_stats.fit32++; if (used <= 65536) gc_stats.fit64++; } if (A->peak < st_peak0) A->peak = st_peak0;)
		if (rc != GC_OK) return rc;
		GC_TICK(A, 8);
		++b->n_gwfa;
	}
	if (ed >= 0) {
		n_mid = n_path - 2 > 0 ? n_path - 2 : 0;
		if (n_mid) gc_pmove_down(path, path + 1, (int64_t)n_mid * 4); /* the inner vertices */
	} else {
		gc_dst_t dst;
		gc_walkv_t *w = 0;
		int32_t n_w = 0;
		if (A->base == base) A->top = mark;
		memset(&dst, 0, sizeof dst);
		dst.v = c0->v ^ 1, dst.target_dist = c1->dist_pre, dst.target_hash = c1->hash_pre, dst.check_hash = 1;
		int rc = GC_E_ARENA;
		if (A->fast_base) { gc_arena_t F; gc_arena_init(&F, A->fast_base, A->fast_cap, 0); rc = gc_shortest_k(&F, G, c1->v ^ 1, 1, &dst, dst.target_dist, MG_MAX_SHORT_K, &w, &n_w); }
		if (rc == GC_E_ARENA) { memset(&dst, 0, sizeof dst); dst.v = c0->v ^ 1, dst.target_dist = c1->dist_pre, dst.target_hash = c1->hash_pre, dst.check_hash = 1; rc = gc_shortest_k(A, G, c1->v ^ 1, 1, &dst, dst.target_dist, MG_MAX_SHORT_K, &w, &n_w); }
		if (rc == GC_E_ARENA) return rc;
		++b->n_shortk;
		if (rc != GC_OK || n_w == 0 || dst.target_hash != dst.hash) { if (A->base == base) A->top = mark; b->failed = 1; return GC_OK; } /* "chain skiped" (gchain1.c:333-338) */
		n_mid = n_w - 2 > 0 ? n_w - 2 : 0;
		GC_ALLOC(A, int32_t, path, n_mid > 0 ? n_mid : 1);
		for (int32_t s = n_w - 2, k = 0; s >= 1; --s) path[k++] = (int32_t)(w[s].v ^ 1); /* found backwards: reverse and flip */
	}
	b->ed = ed, b->n_mid = n_mid, b->mid = path;
	return GC_OK;
}
/* the walk's inner vertices, then chain c1, go behind what is assembled; [mark, base]: where the walk's scratch began (released when nothing was allocated on top of it) */
GC_HD int gc_bridge_append(gc_arena_t *A, gc_asm_t *S, const gc_chain_t *c1, const mg128_t *a, int32_t ed, int32_t n_mid, const int32_t *mid, int64_t mark, const char *base)
{
	if (S->lc.n + n_mid + 1 <= S->lc.m) { /* fits the reserved room: nothing is allocated while the vertices are appended */
		for (int32_t j = 0; j < n_mid; ++j) GC_TRY(gc_asm_vertex(A, S, (uint32_t)mid[j]));
		if (A->base == base) A->top = mark;
	} else for (int32_t j = 0; j < n_mid; ++j) GC_TRY(gc_asm_vertex(A, S, (uint32_t)mid[j])); /* long walk: the scratch stays until the read is done */
	return gc_asm_chain(A, S, c1, a, ed);
}
/* c1 lies on the segment of c0: its anchors beyond the end of c0 extend the last vertex */
GC_HD void gc_bridge_same(gc_asm_t *S, const gc_chain_t *c0, const gc_chain_t *c1, const mg128_t *a)
{
	mg_llchain_t *t = &S->lc.a[S->lc.n - 1];
	int32_t k = 0;
	while (k < c1->cnt && !(GC_AX(a[c1->off + k]) > c0->re && GC_AY(a[c1->off + k]) > c0->qe)) ++k;
	if (k < c1->cnt) {
		t->cnt += c1->cnt - k, t->score += c1->score;
		gc_pcopy(&S->a_out[S->n_a], &a[c1->off + k], (int64_t)(c1->cnt - k) * (int64_t)sizeof(mg128_t));
		S->n_a += c1->cnt - k;
	}
}
/* chain c1 behind chain c0 in the assembly; *failed = 1 when c1 could not be attached */
GC_HD int gc_bridge(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, gc_asm_t *S, int32_t span, const gc_chain_t *c0, const gc_chain_t *c1,
					const mg128_t *a, const char *qseq, int *failed)
{
	*failed = 0;
	if (c1->v != c0->v) {
		gc_bres_t b;
		GC_TRY(gc_vec_reserve(A, S->lc, S->lc.n + 66)); /* room for the usual walk, so that the search's scratch can be released afterwards */
		const int64_t mark = A->top;
		const char *base = A->base;
		GC_TRY(gc_bridge_walk(A, G, P, span, c0, c1, qseq, &b));
		S->n_gwfa += b.n_gwfa, S->n_shortk += b.n_shortk, S->n_fast += b.n_fast;
		if (b.failed) { *failed = 1; return GC_OK; }
		GC_TRY(gc_bridge_append(A, S, c1, a, b.ed, b.n_mid, b.mid, mark, base));
	} else gc_bridge_same(S, c0, c1, a);
	return GC_OK;
}

GC_HD void gc_measure(const gc_graph_t *G, gc_result_t *R)
{
	for (int32_t i = 0; i < R->n_gc; ++i) {
		gc_rec_t *p = &R->gc[i];
		p->qs = p->qe = p->ps = p->pe = -1, p->plen = p->blen = p->mlen = 0, p->n_mini = 0, p->q_span = 0;
		if (p->cnt == 0) continue;
		const mg_llchain_t *first = &R->lc[p->off], *last = &R->lc[p->off + p->cnt - 1];
		const mg128_t *a0 = &R->a[first->off], *a1 = &R->a[last->off + last->cnt - 1];
		p->q_span = GC_ASPAN(*a0);
		p->qs = GC_AY(*a0) + 1 - p->q_span, p->ps = GC_AX(*a0) + 1 - p->q_span;
		p->qe = GC_AY(*a1) + 1;
		const int32_t tail = gc_vlen(G, last->v) - GC_AX(*a1) - 1;
		int32_t n_mini = (int32_t)(a1->x >> 32) - (int32_t)(a0->x >> 32) + 1, rest = 0;
		int32_t blen = 0, mlen = 0, plen = 0, d_mini = 0; /* blen, mlen, d_mini: this lane's share of the sums */
		const mg128_t *before = a0; /* the anchor before the first one of the current vertex */
		for (int32_t j = 0; j < p->cnt; ++j) {
			const mg_llchain_t *q = &R->lc[p->off + j];
			const int32_t vlen = gc_vlen(G, q->v);
			plen += vlen;
			GC_PAR_FOR(k, q->cnt) { /* an anchor is compared with the one before it, wherever that lies */
				const mg128_t *r = &R->a[q->off + k], *prev = k == 0 ? before : r - 1;
				const int32_t span = GC_ASPAN(*r);
				int32_t pl, ql = GC_AY(*r) - GC_AY(*prev);
				if (j == 0 && k == 0) pl = ql = span;
				else if (k == 0) pl = GC_AX(*r) + 1 + rest;
				else pl = GC_AX(*r) - GC_AX(*prev);
				if (ql < 0) ql = -ql, d_mini += (int32_t)(prev->x >> 32) - (int32_t)(r->x >> 32); /* query overlap at a junction */
				blen += pl > ql ? pl : ql;
				mlen += pl > span && ql > span ? span : pl < ql ? pl : ql;
			}
			if (q->cnt == 0) rest += vlen;
			else rest = vlen - GC_AX(R->a[q->off + q->cnt - 1]) - 1, before = &R->a[q->off + q->cnt - 1];
		}
		p->plen = plen, p->blen = gc_sum(blen), p->mlen = gc_sum(mlen);
		p->pe = p->plen - tail;
		p->n_mini = n_mini + gc_sum(d_mini);
	}
}

/* graph chains in descending (score, hash) order through the klib sort; lc[] and a[] follow (gcmisc.c:6-71) */
GC_HD int gc_order_by_score(gc_arena_t *A, gc_result_t *R)
{
	const int64_t mark = A->top;
	const int32_t n = R->n_gc;
	if (n == 0) return GC_OK;
	if (n == 1) { /* one chain: records, vertices and anchors are in their final order; only the vertices' anchor offsets are counted up as below */
		for (int32_t i = 0, k = 0; i < R->n_lc; ++i) R->lc[i].off = k, k += R->lc[i].cnt;
		return GC_OK;
	}
	gc_kv_t *z;
	gc_rec_t *g2;
	mg_llchain_t *l2;
	mg128_t *a2;
	GC_ALLOC(A, gc_kv_t, z, n); GC_ALLOC(A, gc_rec_t, g2, n);
	GC_ALLOC(A, mg_llchain_t, l2, R->n_lc > 0 ? R->n_lc : 1); GC_ALLOC(A, mg128_t, a2, R->n_a > 0 ? R->n_a : 1);
	for (int32_t i = 0; i < n; ++i) z[i].key = (uint64_t)(uint32_t)R->gc[i].score << 32 | R->gc[i].hash, z[i].val = (uint64_t)i;
	GC_TRY(gc_ksort(A, z, n, 8));
	int32_t n_lc = 0, n_a = 0;
	for (int32_t i = n - 1; i >= 0; --i) {
		gc_rec_t g = R->gc[z[i].val];
		gc_pcopy(&l2[n_lc], &R->lc[g.off], (int64_t)g.cnt * (int64_t)sizeof(mg_llchain_t));
		gc_pcopy(&a2[n_a], &R->a[R->lc[g.off].off], (int64_t)g.n_anchor * (int64_t)sizeof(mg128_t));
		g.off = n_lc;
		g2[n - 1 - i] = g;
		n_lc += g.cnt, n_a += g.n_anchor;
	}
	gc_pcopy(R->gc, g2, (int64_t)n * (int64_t)sizeof(gc_rec_t));
	gc_pcopy(R->lc, l2, (int64_t)R->n_lc * (int64_t)sizeof(mg_llchain_t));
	gc_pcopy(R->a, a2, (int64_t)R->n_a * (int64_t)sizeof(mg128_t));
	for (int32_t i = 0, k = 0; i < R->n_lc; ++i) R->lc[i].off = k, k += R->lc[i].cnt;
	A->top = mark;
	return GC_OK;
}

/* from the DP's grouping of the chains (u[], c[]) to graph chains with their vertex walks (gchain1.c:443-520) */
/* A bridge of a read as a unit of work of its own (device: gc_read_p1 lists them, a wavefront per bridge computes them, gc_read_p3 consumes them in order) */
#define GC_JOB_OK     0
#define GC_JOB_FAILED 1   /* no walk of the chosen length ("chain skiped") */
#define GC_JOB_ARENA  2   /* the scratch arena was too small: the read is run again the monolithic way in a large one */
typedef struct {
	const gc_chain_t *c0, *c1;    /* in the read's arena */
	const char *qseq;
	int32_t read, span, status;
	int32_t ed, n_mid, n_gwfa, n_shortk, n_fast;
	int64_t mid_off;              /* the walk's inner vertices: mid_pool[mid_off .. mid_off + n_mid) */
} gc_job_t;

GC_HD int gc_asm_kept(const gc_par_t *P, const uint64_t *u, const gc_chain_t *c, int32_t i, int32_t st)
{
	const int32_t ni = (int32_t)(uint32_t)u[i];
	int32_t m = 0;
	for (int32_t j = 0; j < ni; ++j) m += c[st + j].cnt;
	return m >= P->min_gc_cnt && (int64_t)(u[i] >> 32) >= P->min_gc_score;
}
/* first half of the assembly: which groups of the DP become graph chains, their records with score and hash -- both decided on the chains AS THE DP LEFT THEM, before any junction
 * is touched (gchain1.c:452-456,472-484: the anchor count and the hash are taken first, resolve_overlap runs afterwards) --, then the junctions of consecutive chains made
 * monotone (they only touch the chains of their own graph chain), and the number of bridges between chains on different segments, in the order gc_assemble_run() meets them.
 * kept[i] != 0: group i is graph chain number kept[i] - 1. */
GC_HD int gc_assemble_begin(gc_arena_t *A, const gc_par_t *P, int32_t n_u, const uint64_t *u, gc_chain_t *c, const mg128_t *a, uint32_t hash, gc_result_t *R, int32_t **kept_, int32_t *n_jobs)
{
	int32_t n_gc = 0, nj = 0, *kept;
	R->n_gc = R->n_lc = R->n_a = 0, R->gc = 0, R->lc = 0;
	*n_jobs = 0, *kept_ = 0;
	GC_ALLOC(A, int32_t, kept, n_u > 0 ? n_u : 1);
	for (int32_t i = 0, st = 0; i < n_u; st += (int32_t)(uint32_t)u[i], ++i) kept[i] = gc_asm_kept(P, u, c, i, st) ? ++n_gc : 0;
	R->n_gc = n_gc, *kept_ = kept;
	if (n_gc == 0) return GC_OK;
	GC_ALLOC(A, gc_rec_t, R->gc, n_gc);
	memset(R->gc, 0, (size_t)n_gc * sizeof(gc_rec_t));
	for (int32_t i = 0, st = 0; i < n_u; st += (int32_t)(uint32_t)u[i], ++i) {
		const int32_t ni = (int32_t)(uint32_t)u[i];
		if (!kept[i]) continue;
		gc_rec_t *g = &R->gc[kept[i] - 1];
		uint32_t h = hash;
		for (int32_t j = 0; j < ni; ++j) h += gc_hash32((uint32_t)c[st + j].qs) + gc_hash32((uint32_t)c[st + j].re) + gc_hash32(c[st + j].v);
		g->hash = gc_hash32(h), g->score = (int32_t)(u[i] >> 32);
		for (int32_t j = 1; j < ni; ++j) gc_untangle(&c[st + j - 1], &c[st + j], a);
		for (int32_t j0 = 0, j = 1; j < ni; ++j) {
			if (c[st + j].cnt <= 0) continue;
			nj += c[st + j].v != c[st + j0].v;
			j0 = j;
		}
	}
	*n_jobs = nj;
	return GC_OK;
}
/* the bridges gc_assemble_begin() counted, in the same order */
GC_HD void gc_assemble_jobs(int32_t n_u, const uint64_t *u, const int32_t *kept, const gc_chain_t *c, int32_t read, int32_t span, const char *qseq, gc_job_t *jobs)
{
	int32_t nj = 0;
	for (int32_t i = 0, st = 0; i < n_u; st += (int32_t)(uint32_t)u[i], ++i) {
		const int32_t ni = (int32_t)(uint32_t)u[i];
		if (!kept[i]) continue;
		for (int32_t j0 = 0, j = 1; j < ni; ++j) {
			if (c[st + j].cnt <= 0) continue;
			if (c[st + j].v != c[st + j0].v) {
				gc_job_t *q = &jobs[nj++];
				q->c0 = &c[st + j0], q->c1 = &c[st + j], q->qseq = qseq, q->read = read, q->span = span, q->status = GC_JOB_OK;
				q->ed = -1, q->n_mid = 0, q->n_gwfa = q->n_shortk = q->n_fast = 0, q->mid_off = 0;
			}
			j0 = j;
		}
	}
}
/* one bridge, on whatever wavefront (or host thread) gets it: scratch in A, the inner vertices of the walk copied to mid_out[0 .. n_mid) by the caller's rule */
GC_HD int gc_job_run(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, gc_job_t *q, gc_bres_t *b)
{
	const int rc = gc_bridge_walk(A, G, P, q->span, q->c0, q->c1, q->qseq, b);
	if (rc == GC_E_ARENA) { q->status = GC_JOB_ARENA; return GC_OK; }
	if (rc != GC_OK) return rc;
	q->status = b->failed ? GC_JOB_FAILED : GC_JOB_OK;
	q->ed = b->ed, q->n_mid = b->n_mid, q->n_gwfa = b->n_gwfa, q->n_shortk = b->n_shortk, q->n_fast = b->n_fast;
	return GC_OK;
}
/* second half: chains and walks in order (gchain1.c:409-465), measuring, ordering.  jobs == 0: every bridge is computed where it is met; otherwise the bridges between chains on
 * different segments were computed beforehand (jobs[], inner vertices in mid_pool) and are consumed in order. */
GC_HD int gc_assemble_run(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, int32_t n_u, const uint64_t *u, const int32_t *kept, gc_chain_t *c, const mg128_t *a,
						  const char *qseq, gc_result_t *R, const gc_job_t *jobs, const int32_t *mid_pool)
{
	if (R->n_gc == 0) return GC_OK;
	GC_STATE(gc_asm_t, S);
	memset(&S, 0, sizeof S);
	S.a_out = R->a;
	const int32_t span = GC_ASPAN(a[0]);
	int32_t jk = 0;
	for (int32_t i = 0, st = 0; i < n_u; st += (int32_t)(uint32_t)u[i], ++i) {
		const int32_t ni = (int32_t)(uint32_t)u[i], n_a0 = S.n_a, n_lc0 = S.lc.n;
		if (!kept[i]) continue;
		gc_rec_t *g = &R->gc[kept[i] - 1]; /* (score and hash: gc_assemble_begin) */
		g->off = n_lc0;
		GC_TRY(gc_asm_chain(A, &S, &c[st], a, -1));
		for (int32_t j0 = 0, j = 1; j < ni; ++j) {
			if (c[st + j].cnt <= 0) continue; /* emptied by the untangling: skipped, its neighbours are bridged directly */
			int failed = 0;
			if (jobs && c[st + j].v != c[st + j0].v) {
				const gc_job_t *q = &jobs[jk++];
				if (q->status == GC_JOB_ARENA) return GC_E_ARENA;
				S.n_gwfa += q->n_gwfa, S.n_shortk += q->n_shortk, S.n_fast += q->n_fast;
				if (q->status == GC_JOB_FAILED) failed = 1;
				else {
					GC_TRY(gc_vec_reserve(A, S.lc, S.lc.n + 66));
					GC_TRY(gc_bridge_append(A, &S, &c[st + j], a, q->ed, q->n_mid, mid_pool + q->mid_off, A->top, A->base));
				}
			} else GC_TRY(gc_bridge(A, G, P, &S, span, &c[st + j0], &c[st + j], a, qseq, &failed));
			if (failed) /* no walk of the chosen length between the two: go through the emptied chains in between, pair by pair */
				for (int32_t t = j0; t < j; ++t) { GC_TRY(gc_bridge(A, G, P, &S, span, &c[st + t], &c[st + t + 1], a, qseq, &failed)); if (failed) return GC_E_BUG; }
			j0 = j;
		}
		g->cnt = S.lc.n - n_lc0, g->n_anchor = S.n_a - n_a0;
	}
	R->n_lc = S.lc.n, R->n_a = S.n_a, R->lc = S.lc.a;
	R->n_gwfa = S.n_gwfa, R->n_shortk += S.n_shortk, R->n_fast = S.n_fast;
	GC_TICK(A, 5);
	gc_measure(G, R);
	GC_TICK(A, 9);
	{ const int rc_ = gc_order_by_score(A, R); GC_TICK(A, 10); return rc_; }
}
GC_HD int gc_assemble(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, int32_t n_u, const uint64_t *u, gc_chain_t *c, const mg128_t *a, uint32_t hash,
					  const char *qseq, gc_result_t *R)
{
	int32_t n_jobs, *kept;
	GC_TRY(gc_assemble_begin(A, P, n_u, u, c, a, hash, R, &kept, &n_jobs));
	return gc_assemble_run(A, G, P, n_u, u, kept, c, a, qseq, R, 0, 0);
}


/* ------------------------------------------------------------------------------------------------ primary / secondary, filters */

/* a chain whose query interval is mostly covered by a better one becomes its secondary (gcmisc.c:73-128) */
GC_HD int gc_assign_parents(gc_arena_t *A, const gc_par_t *P, int32_t n, gc_rec_t *r)
{
	if (n <= 0) return GC_OK;
	const int64_t mark = A->top;
	uint64_t *cov;
	int32_t *prim, n_prim = 1;
	GC_ALLOC(A, uint64_t, cov, n); GC_ALLOC(A, int32_t, prim, n);
	for (int32_t i = 0; i < n; ++i) r[i].id = i;
	prim[0] = 0, r[0].parent = 0;
	for (int32_t i = 1; i < n; ++i) {
		gc_rec_t *ri = &r[i];
		const int32_t si = ri->qs, ei = ri->qe;
		int32_t n_cov = 0, uncov = 0, j;
		for (j = 0; j < n_prim; ++j) { /* the parts of [si, ei) covered by primaries */
			int32_t sj = r[prim[j]].qs, ej = r[prim[j]].qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
		}
		j = n_prim;
		if (n_cov > 0) {
			int32_t x = si;
			gc_sort_u64(cov, n_cov);
			for (int32_t c = 0; c < n_cov; ++c) {
				if ((int32_t)(cov[c] >> 32) > x) uncov += (int32_t)(cov[c] >> 32) - x;
				x = (int32_t)(uint32_t)cov[c] > x ? (int32_t)(uint32_t)cov[c] : x;
			}
			if (ei > x) uncov += ei - x;
			for (j = 0; j < n_prim; ++j) {
				gc_rec_t *rp = &r[prim[j]];
				const int32_t sj = rp->qs, ej = rp->qe;
				if (ej <= si || sj >= ei) continue;
				const int32_t lo = ej - sj < ei - si ? ej - sj : ei - si, hi = ej - sj > ei - si ? ej - sj : ei - si;
				const int32_t ol = (ei < ej ? ei : ej) - (si > sj ? si : sj); /* > 0 here */
				if ((float)ol / lo - (float)uncov / hi > P->mask_level) {
					ri->parent = rp->parent;
					if (rp->subsc < ri->score) rp->subsc = ri->score;
					if (ri->cnt >= rp->cnt) ++rp->n_sub;
					break;
				}
			}
		}
		if (j == n_prim) prim[n_prim++] = i, ri->parent = i, ri->n_sub = 0;
	}
	A->top = mark;
	return GC_OK;
}

/* secondaries far below their primary, or beyond best_n, or identical to it in coordinates, are filtered (gcmisc.c:130-148) */
GC_HD void gc_filter_secondaries(const gc_par_t *P, int32_t n, gc_rec_t *r)
{
	if (!(P->pri_ratio > 0.0f) || n <= 0) return;
	int32_t n_2nd = 0;
	for (int32_t i = 0; i < n; ++i) {
		const gc_rec_t *rp = &r[r[i].parent];
		if (r[i].parent == i) { r[i].flt = 0; continue; }
		const int close = (float)r[i].score >= (float)rp->score * P->pri_ratio || r[i].score + P->k * 2 >= rp->score;
		const int same = r[i].qs == rp->qs && r[i].qe == rp->qe && r[i].ps == rp->ps && r[i].pe == rp->pe;
		if (close && n_2nd < P->best_n && !same) r[i].flt = 0, ++n_2nd;
		else r[i].flt = 1;
	}
}

/* remove the filtered chains, close the gaps in lc[] and a[] (gcmisc.c:150-188) */
GC_HD int gc_drop_filtered(gc_arena_t *A, gc_result_t *R)
{
	if (R->n_gc == 0) return GC_OK;
	const int64_t mark = A->top;
	int32_t *o2n, n_gc = 0, n_lc = 0, n_a = 0, lc0 = 0, a0 = 0;
	GC_ALLOC(A, int32_t, o2n, R->n_gc);
	for (int32_t i = 0; i < R->n_gc; ++i) o2n[i] = (R->gc[i].flt || R->gc[i].cnt == 0) ? -1 : n_gc++;
	n_gc = 0;
	for (int32_t i = 0; i < R->n_gc; ++i) {
		const gc_rec_t r = R->gc[i];
		if (o2n[i] >= 0) {
			gc_pmove_down(&R->a[n_a], &R->a[a0], (int64_t)r.n_anchor * (int64_t)sizeof(mg128_t));
			gc_pmove_down(&R->lc[n_lc], &R->lc[lc0], (int64_t)r.cnt * (int64_t)sizeof(mg_llchain_t));
			R->gc[n_gc] = r;
			R->gc[n_gc].id = n_gc, R->gc[n_gc].parent = o2n[r.parent];
			++n_gc, n_lc += r.cnt, n_a += r.n_anchor;
		}
		lc0 += r.cnt, a0 += r.n_anchor;
	}
	R->n_gc = n_gc, R->n_lc = n_lc, R->n_a = n_a;
	for (int32_t i = 0, l = 0, k = 0; i < n_gc; ++i) { /* offsets again */
		gc_rec_t *g = &R->gc[i];
		g->off = l, g->n_anchor = 0;
		for (int32_t j = 0; j < g->cnt; ++j) { mg_llchain_t *q = &R->lc[l + j]; q->off = k, k += q->cnt, g->n_anchor += q->cnt; }
		l += g->cnt;
	}
	A->top = mark;
	return GC_OK;
}

/* ------------------------------------------------------------------------------------------------ one read */

typedef struct {
	GC_F int32_t qlen; GC_F uint32_t hash;
	GC_F int32_t n_u; GC_F const uint64_t *u;      /* linear chains: score<<32 | count */
	GC_F mg128_t *a;                          /* their anchors, chain after chain; MODIFIED in place (flags, minimizer ranks) */
	GC_F int32_t n_mini; GC_F const int32_t *mini_pos;
	GC_F const char *qseq;
} gc_read_t;

/* What gc_read_p1() leaves for gc_read_p3() (next to A, rd and R, which the caller keeps) */
typedef struct { GC_F gc_chain_t *c; GC_F uint64_t *u2; GC_F int32_t *kept; GC_F int32_t n_c; GC_F int32_t n_u2; GC_F int32_t n_jobs; GC_F int32_t done; } gc_split_t;

/* One read in three parts, so that the device can give the bridges of a read -- independent GWFA calls / graph searches, most of the cycles, and what makes one read take
 * fifty times another -- a wavefront each: (1) chain records, clean-up, anchor ranks, DP + reachability, the assembly's first half; sp->n_jobs bridges are then listed with
 * gc_assemble_jobs() and run with gc_job_run(), in any order, anywhere; (3) assembly, measuring, ordering, parents, filters.  jobs == 0 in part 3: bridges are computed where
 * they are met (gc_map_read(): the host, the device's large-arena retry).
 * R->a must point to a buffer for as many anchors as the chains hold.  Returns GC_OK, GC_E_ARENA (nothing usable in R), or GC_E_BUG
 * (R holds zero chains: the reference's own bail-out paths). */
GC_HD int gc_read_p1(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, const gc_read_t *rd, gc_result_t *R, gc_split_t *sp)
{
	gc_chain_t *c = 0;
	uint64_t *u2 = 0;
	int32_t n_c = rd->n_u, n_u2 = 0, n_jobs = 0, *kept = 0;
	R->n_gc = R->n_lc = R->n_a = 0, R->gc = 0, R->lc = 0, R->n_gwfa = R->n_shortk = R->n_fast = 0;
	sp->c = 0, sp->u2 = 0, sp->kept = 0, sp->n_c = sp->n_u2 = sp->n_jobs = 0, sp->done = 1;
	if (rd->n_u <= 0) return GC_OK;
	GC_TICK(A, 0);
	GC_TRY(gc_make_chains(A, rd->n_u, rd->u, rd->a, &c));
	GC_TICK(A, 1);
	if (n_c > 1) GC_TRY(gc_clean_chains(A, P, rd->a, c, &n_c));
	GC_TICK(A, 2);
	for (int32_t i = 0; i < n_c; ++i) GC_TRY(gc_index_anchors(&rd->a[c[i].off], c[i].cnt, rd->mini_pos, rd->n_mini));
	GC_TICK(A, 3);
	GC_TRY(gc_chain_dp(A, G, P, rd->qlen, rd->a, c, &n_c, &u2, &n_u2, &R->n_shortk));
	GC_TICK(A, 4);
	if (n_u2 == 0) return GC_OK;
	GC_TRY(gc_assemble_begin(A, P, n_u2, u2, c, rd->a, rd->hash, R, &kept, &n_jobs));
	sp->c = c, sp->u2 = u2, sp->kept = kept, sp->n_c = n_c, sp->n_u2 = n_u2, sp->n_jobs = n_jobs, sp->done = 0;
	return GC_OK;
}
GC_HD int gc_read_p3(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, const gc_read_t *rd, gc_result_t *R, const gc_split_t *sp, const gc_job_t *jobs, const int32_t *mid_pool)
{
	if (sp->done) return GC_OK;
	GC_TRY(gc_assemble_run(A, G, P, sp->n_u2, sp->u2, sp->kept, sp->c, rd->a, rd->qseq, R, jobs, mid_pool));
	GC_TICK(A, 5);
	for (int32_t i = 0; i < R->n_gc; ++i) R->gc[i].parent = R->gc[i].id = i, R->gc[i].subsc = R->gc[i].n_sub = R->gc[i].flt = 0;
	GC_TRY(gc_assign_parents(A, P, R->n_gc, R->gc));
	gc_filter_secondaries(P, R->n_gc, R->gc);
	{ const int rc_ = gc_drop_filtered(A, R); GC_TICK(A, 6); return rc_; }
}
GC_HD int gc_map_read(gc_arena_t *A, const gc_graph_t *G, const gc_par_t *P, const gc_read_t *rd, gc_result_t *R)
{
	gc_split_t sp;
	GC_TRY(gc_read_p1(A, G, P, rd, R, &sp));
	return gc_read_p3(A, G, P, rd, R, &sp, 0, 0);
}

#endif
