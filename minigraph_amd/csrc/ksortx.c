/*
 * ksortx.c -- host emulation of the exact permutation applied by klib's in-place MSD byte radix
 * sort (reference ksort.h:112-162).  See mga_host.h for why the permutation itself matters.
 * Iterative formulation with an explicit range stack (the device twin is dev_klibsort.h).
 */
#include "mga_host.h"

typedef struct { uint64_t key; int64_t src; } kx_t;

static void kx_insertion(kx_t *a, int64_t n) /* strict '<' : stable (ksort.h:118-128) */
{
	int64_t i, j;
	for (i = 1; i < n; ++i) {
		kx_t t = a[i];
		for (j = i; j > 0 && t.key < a[j-1].key; --j) a[j] = a[j-1];
		a[j] = t;
	}
}

typedef struct { int64_t b, e; int sh; } rng_t;

/* Ranges are independent once their parent has been partitioned -- the order in which they are taken cannot change the result -- so a sort of 10^7 records runs its first
 * partitioning passes on the calling thread and hands the sub-ranges to the pool (round 5: one 50 Mbp contig's anchor sort and the (score, index) sort of its backtrack were
 * 0.4 + 0.4 s of ONE thread on the critical path of every contig of a -x asm batch).  stop_at > 0: return to the caller, with the open ranges left in *stk_ / *top_, as soon as
 * that many ranges are open or the largest open range is below min_par. */
static void kx_sort_ranges(kx_t *a, rng_t **stk_, int64_t *top_, int64_t *cap_, int64_t stop_at, int64_t min_par);

static void kx_sort(kx_t *a, int64_t n, int key_bytes)
{
	rng_t *stk;
	int64_t top = 0, cap = n / 64 + 8;
	if (n <= 64) { kx_insertion(a, n); return; }
	stk = MGA_MALLOC(rng_t, cap);
	stk[top].b = 0, stk[top].e = n, stk[top].sh = (key_bytes - 1) * 8, ++top;
	{ /* A pass whose byte is the same in every key of its range moves nothing (every element is "already in its bucket", ksort.h:146) and hands the whole range to the next byte:
	   * the leading passes over scores (32-bit values in a 64-bit key) or over one strand's anchors are skipped without looking -- the bytes above the highest differing bit */
		uint64_t d = 0;
		int64_t i;
		for (i = 1; i < n; ++i) d |= a[i].key ^ a[0].key;
		while (stk[0].sh > 0 && (d >> stk[0].sh) == 0) stk[0].sh -= 8;
	}
	kx_sort_ranges(a, &stk, &top, &cap, 0, 0);
	free(stk);
}

static void kx_sort_ranges(kx_t *a, rng_t **stk_, int64_t *top_, int64_t *cap_, int64_t stop_at, int64_t min_par)
{
	rng_t *stk = *stk_;
	int64_t top = *top_, cap = *cap_;
	while (top > 0) {
		int64_t head[256], tail[256], cnt[256], i, pos;
		rng_t r;
		int k;
		if (stop_at > 0) { /* the parallel driver: take the LARGEST open range next; stop when there are enough of them or none is worth splitting further */
			int64_t big = 0, j;
			for (j = 1; j < top; ++j) if (stk[j].e - stk[j].b > stk[big].e - stk[big].b) big = j;
			if (top >= stop_at || stk[big].e - stk[big].b < min_par) break;
			r = stk[big]; stk[big] = stk[--top];
		} else r = stk[--top];
		memset(cnt, 0, sizeof cnt);
		for (i = r.b; i < r.e; ++i) ++cnt[a[i].key >> r.sh & 0xff];
		if (r.sh > 0 && cnt[a[r.b].key >> r.sh & 0xff] == r.e - r.b) { stk[top].b = r.b, stk[top].e = r.e, stk[top].sh = r.sh > 8 ? r.sh - 8 : 0, ++top; continue; } /* (one bucket: the same, found by counting) */
		for (k = 0, pos = r.b; k < 256; ++k) head[k] = pos, pos += cnt[k], tail[k] = pos;
		for (k = 0; k < 256; ++k) { /* displacement cycles, lowest bucket first (ksort.h:141-153) */
			while (head[k] != tail[k]) {
				kx_t carry = a[head[k]];
				int l = (int)(carry.key >> r.sh & 0xff);
				if (l == k) { ++head[k]; continue; }
				do {
					kx_t t = a[head[l]];
					a[head[l]++] = carry;
					if ((head[l] & 3) == 0) __builtin_prefetch(&a[head[l] + 12], 1, 0); /* every bucket's cursor walks forward: the next lines of the one just written are on their way before the chain of displacements comes back to it ([measured] the pass over 9 M records is a chain of dependent misses otherwise) */
					carry = t;
					l = (int)(carry.key >> r.sh & 0xff);
				} while (l != k);
				a[head[k]++] = carry;
			}
		}
		if (r.sh > 0) { /* ksort.h:155-160 */
			int nsh = r.sh > 8 ? r.sh - 8 : 0;
			for (k = 0; k < 256; ++k) {
				int64_t st = tail[k] - cnt[k];
				if (cnt[k] > 64) {
					if (top == cap) { cap += cap >> 1; stk = MGA_REALLOC(rng_t, stk, cap); }
					stk[top].b = st, stk[top].e = tail[k], stk[top].sh = nsh, ++top;
				} else if (cnt[k] > 1) kx_insertion(a + st, cnt[k]);
			}
		}
	}
	*stk_ = stk, *top_ = top, *cap_ = cap;
}

typedef struct { kx_t *a; rng_t *r; } kx_par_t;
static void kx_range_worker(void *data, int64_t i, int tid)
{
	kx_par_t *P = (kx_par_t*)data;
	rng_t *stk;
	int64_t top = 1, cap = (P->r[i].e - P->r[i].b) / 64 + 8;
	(void)tid;
	stk = MGA_MALLOC(rng_t, cap);
	stk[0] = P->r[i];
	kx_sort_ranges(P->a, &stk, &top, &cap, 0, 0);
	free(stk);
}

__thread int mga_ksort_threads = 1; /* threads a large sort started on THIS thread may use: set by the worker that sorts for the duration of its task (mapper.c: rq_*_worker), so that one batch's thread count never leaks into another call */

static void kx_sort_mt(kx_t *a, int64_t n, int key_bytes, int n_threads)
{
	rng_t *stk;
	int64_t top = 0, cap = n / 64 + 8, i;
	kx_par_t P;
	if (n_threads <= 1 || n < (1 << 18)) { kx_sort(a, n, key_bytes); return; }
	stk = MGA_MALLOC(rng_t, cap);
	stk[top].b = 0, stk[top].e = n, stk[top].sh = (key_bytes - 1) * 8, ++top;
	{ /* (the leading bytes every key shares: as in kx_sort) */
		uint64_t d = 0;
		for (i = 1; i < n; ++i) d |= a[i].key ^ a[0].key;
		while (stk[0].sh > 0 && (d >> stk[0].sh) == 0) stk[0].sh -= 8;
	}
	kx_sort_ranges(a, &stk, &top, &cap, (int64_t)n_threads * 8, n / ((int64_t)n_threads * 4) > 4096 ? n / ((int64_t)n_threads * 4) : 4096);
	if (top > 0) { P.a = a, P.r = stk; mga_parallel_for(n_threads, top, kx_range_worker, &P); }
	free(stk);
}

void mga_ksort_perm(int64_t n, const uint64_t *key, int key_bytes, int64_t *perm)
{
	int64_t i;
	kx_t *a;
	if (n <= 0) return;
	a = MGA_MALLOC(kx_t, n);
	for (i = 0; i < n; ++i) a[i].key = key[i], a[i].src = i;
	kx_sort(a, n, key_bytes);
	for (i = 0; i < n; ++i) perm[i] = a[i].src;
	free(a);
}

void mga_ksort_128x(int64_t n, mg128_t *a) /* radix_sort_128x (ksort.h via map-algo.c:12): in place, the key is x, y travels with it -- the record the sort above moves IS an mg128_t */
{
	_Static_assert(sizeof(kx_t) == sizeof(mg128_t), "kx_t / mg128_t");
	if (n <= 1) return;
	kx_sort_mt((kx_t*)a, n, 8, mga_ksort_threads);
}

static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

void mga_ksort_u64(int64_t n, uint64_t *a) /* whole value is the key: any correct sort gives the same array */
{
	if (n > 1) qsort(a, (size_t)n, 8, cmp_u64);
}
