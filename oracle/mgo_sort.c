/*
 * ORACLE restatement of klib's in-place MSD byte radix sort (ksort.h:112-162, RS_MIN_SIZE=64,
 * RS_MAX_BITS=8).  The sort is NOT stable for n>64 and the permutation it applies to records with
 * equal keys is part of the reference's observable behaviour (tie order of anchors, of chain ends
 * by score, ...), so it is restated move for move:
 *   - n <= 64: insertion sort with a strict '<' (stable)                      ksort.h:118-128
 *   - else bucket by the current key byte, permute in place by following displacement cycles,
 *     bucket 0 first (ksort.h:141-153), then recurse on the next byte into buckets > 64 and
 *     insertion-sort the others (ksort.h:155-160); nothing below byte 0.
 * Record movement depends on keys only, so we sort (key, original index) pairs and then gather.
 */
#include <stdlib.h>
#include <string.h>
#include "mgo.h"

typedef struct { uint64_t key; int64_t idx; } kp_t;

static void ins_sort(kp_t *a, int64_t n)
{
	int64_t i, j;
	for (i = 1; i < n; ++i) {
		kp_t t = a[i];
		if (!(t.key < a[i-1].key)) continue;
		for (j = i; j > 0 && t.key < a[j-1].key; --j) a[j] = a[j-1];
		a[j] = t;
	}
}

static void flag_sort(kp_t *a, int64_t n, int shift)
{
	int64_t head[256], tail[256], cnt[256], i;
	int k;
	memset(cnt, 0, sizeof cnt);
	for (i = 0; i < n; ++i) ++cnt[a[i].key >> shift & 0xff];
	for (k = 0, i = 0; k < 256; ++k) head[k] = i, i += cnt[k], tail[k] = i;
	for (k = 0; k < 256; ++k) {
		while (head[k] != tail[k]) {
			int l = (int)(a[head[k]].key >> shift & 0xff);
			if (l == k) { ++head[k]; continue; }
			{
				kp_t carry = a[head[k]];
				do { /* drop carry at the head of its bucket, pick up what was there */
					kp_t t = a[head[l]];
					a[head[l]++] = carry;
					carry = t;
					l = (int)(carry.key >> shift & 0xff);
				} while (l != k);
				a[head[k]++] = carry;
			}
		}
	}
	if (shift > 0) {
		int next = shift > 8 ? shift - 8 : 0;
		for (k = 0, i = 0; k < 256; ++k) {
			if (cnt[k] > 64) flag_sort(a + i, cnt[k], next);
			else if (cnt[k] > 1) ins_sort(a + i, cnt[k]);
			i += cnt[k];
		}
	}
}

static void kp_sort(kp_t *a, int64_t n, int key_bytes)
{
	if (n <= 64) ins_sort(a, n);
	else flag_sort(a, n, (key_bytes - 1) * 8);
}

void mgo_sort128x(mgo128_t *a, int64_t n)
{
	int64_t i;
	kp_t *p;
	mgo128_t *b;
	if (n <= 1) return;
	p = (kp_t*)malloc(n * sizeof(kp_t));
	b = (mgo128_t*)malloc(n * sizeof(mgo128_t));
	for (i = 0; i < n; ++i) p[i].key = a[i].x, p[i].idx = i;
	kp_sort(p, n, 8);
	for (i = 0; i < n; ++i) b[i] = a[p[i].idx];
	memcpy(a, b, n * sizeof(mgo128_t));
	free(p); free(b);
}

void mgo_sort64(uint64_t *a, int64_t n)
{
	int64_t i;
	kp_t *p;
	if (n <= 1) return;
	p = (kp_t*)malloc(n * sizeof(kp_t));
	for (i = 0; i < n; ++i) p[i].key = a[i], p[i].idx = i;
	kp_sort(p, n, 8);
	for (i = 0; i < n; ++i) a[i] = p[i].key;
	free(p);
}
