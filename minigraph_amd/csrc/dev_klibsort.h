// dev_klibsort.h -- device emulation of klib's in-place MSD byte radix sort (reference ksort.h:112-162,
// instantiated as radix_sort_128x in misc.c:9-10), wave-cooperative.
//
// The reference sort is unstable for n > 64 and the order it leaves equal keys in is observable
// (anchor ties, chain ends of equal score, ...).  Results must be bit-identical, so the permutation
// is reproduced move for move:
//   n <= 64            insertion sort with strict '<'  == any stable sort  -> wave rank sort (parallel)
//   n  > 64            histogram of the current key byte (parallel, LDS atomics), bucket bounds
//                      (wave prefix sum), then the displacement-cycle permutation, which is inherently
//                      sequential, by lane 0; buckets > 64 recurse on the next byte via an explicit
//                      range stack, buckets 2..64 get the stable rank sort.
// A level whose elements all share one byte value moves nothing and goes straight to the next byte.
#ifndef MGA_DEV_KLIBSORT_H
#define MGA_DEV_KLIBSORT_H

#include "dev_common.h"
#include "../../include/minigraph_amd.h"

struct klib_lds_t { int32_t cnt[256], head[256], tail[256]; };

// stable sort of a[0..m), m <= 64, by x: lane i owns element i, rank by 64 shuffles
__device__ __forceinline__ void klib_rank_sort64(mg128_t *a, int m)
{
	const int lane = threadIdx.x & 63;
	mg128_t e; e.x = ~0ULL, e.y = 0;
	if (lane < m) e = a[lane];
	int rank = 0;
	for (int j = 0; j < m; ++j) {
		const uint64_t kj = __shfl(e.x, j);
		rank += (kj < e.x || (kj == e.x && j < lane)) ? 1 : 0;
	}
	__syncthreads();
	if (lane < m) a[rank] = e;
	__syncthreads();
}

// a[0..n): in-place, all 64 lanes of a single-wave workgroup call with uniform arguments.
// stk: global scratch, >= 3*(n/64+2) int32.  L: LDS scratch.
__device__ void klib_sort128x(mg128_t *a, int64_t n, int32_t *stk, klib_lds_t *L)
{
	const int lane = threadIdx.x & 63;
	if (n <= 1) return;
	if (n <= 64) { klib_rank_sort64(a, (int)n); return; }
	int top = 0;
	if (lane == 0) { stk[0] = 0; stk[1] = (int32_t)n; stk[2] = 56; }
	top = 1;
	__syncthreads();
	while (top > 0) {
		--top;
		const int32_t b = stk[3 * top], e = stk[3 * top + 1], sh = stk[3 * top + 2];
		const int32_t m = e - b;
		__syncthreads();
		for (int q = lane; q < 256; q += 64) L->cnt[q] = 0;
		__syncthreads();
		for (int32_t i = lane; i < m; i += 64) atomicAdd(&L->cnt[(a[b + i].x >> sh) & 0xff], 1);
		__syncthreads();
		{ // bucket bounds: lane owns 4 consecutive buckets
			const int c0 = L->cnt[4 * lane], c1 = L->cnt[4 * lane + 1], c2 = L->cnt[4 * lane + 2], c3 = L->cnt[4 * lane + 3];
			const int s4 = c0 + c1 + c2 + c3;
			const int excl = mga_wave_incl_scan_i32(s4) - s4 + b;
			L->head[4 * lane] = excl; L->tail[4 * lane] = excl + c0;
			L->head[4 * lane + 1] = excl + c0; L->tail[4 * lane + 1] = excl + c0 + c1;
			L->head[4 * lane + 2] = excl + c0 + c1; L->tail[4 * lane + 2] = excl + c0 + c1 + c2;
			L->head[4 * lane + 3] = excl + c0 + c1 + c2; L->tail[4 * lane + 3] = excl + s4;
		}
		__syncthreads();
		// does one bucket hold everything?  then the permutation is the identity
		bool single = false;
		for (int q = lane; q < 256; q += 64) single = single || (L->cnt[q] == m);
		single = __ballot(single) != 0;
		if (!single) {
			if (lane == 0) { // displacement cycles, bucket 0 first (ksort.h:141-153)
				for (int k = 0; k < 256; ++k) {
					int32_t hk = L->head[k];
					const int32_t tk = L->tail[k];
					while (hk != tk) {
						mg128_t carry = a[hk];
						int l = (int)(carry.x >> sh & 0xff);
						if (l == k) { ++hk; continue; }
						do {
							const int32_t hl = L->head[l];
							const mg128_t t = a[hl];
							a[hl] = carry;
							L->head[l] = hl + 1;
							carry = t;
							l = (int)(carry.x >> sh & 0xff);
						} while (l != k);
						a[hk++] = carry;
					}
					L->head[k] = hk;
				}
			}
			__syncthreads();
		}
		if (sh > 0) {
			const int32_t nsh = sh > 8 ? sh - 8 : 0;
			for (int k = 0; k < 256; ++k) { // uniform loop
				const int32_t c = L->cnt[k];
				if (c <= 1) continue;
				const int32_t st = L->tail[k] - c;
				if (c > 64) {
					if (lane == 0) { stk[3 * top] = st; stk[3 * top + 1] = st + c; stk[3 * top + 2] = nsh; }
					++top;
				} else klib_rank_sort64(a + st, c);
			}
			__syncthreads();
		}
	}
}

#endif
