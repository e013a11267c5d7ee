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

static void kx_sort(kx_t *a, int64_t n, int key_bytes)
{
	typedef struct { int64_t b, e; int sh; } rng_t;
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
	while (top > 0) {
		int64_t head[256], tail[256], cnt[256], i, pos;
		rng_t r = stk[--top];
		int k;
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
	kx_sort((kx_t*)a, n, 8);
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
