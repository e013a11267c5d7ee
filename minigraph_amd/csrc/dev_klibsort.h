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

// ---- the same permutation for 64 < n <= KLIB_SMALL_CAP elements, with the sequential part in LDS (round 6) ----
// klib_sort128x() above runs the displacement cycles on the 16-byte elements where they lie: two dependent trips to global memory per element moved, by one lane, and one
// rank sort -- load, 64 shuffles, store, two barriers -- per bucket of 2..64 elements.  [measured, round 6, profiles/r06f_lchain_phases.txt] 780 k cycles per backtrack of
// k_lchain (250 chain ends sorted by score), 16 % of that kernel, and as much again for the 300 anchors of a long-join rescue.
// The moves of the reference's sort depend on the BUCKET of each element only, so here one lane replays them on (byte, index) pairs in LDS -- two dependent LDS reads per
// element moved -- and the elements themselves move once, at the end, through a scratch array:
//   * one pass over the keys finds the highest byte in which any two differ: every level above moves nothing (all elements in one bucket) and is skipped;
//   * a level: the lanes fetch the byte of each element (through the permutation so far), histogram by LDS atomics, bucket bounds by a wave scan, the cycles by lane 0 over
//     the non-empty buckets only (ballot masks instead of a loop over 256 counters);
//   * the insertion sorts of the buckets of 2..64 elements (= stable sorts by the whole key): consecutive buckets are packed into passes of up to 64 positions, every lane
//     ranks its key among the lanes of its own bucket -- one gather and 64 lane reads per pass instead of per bucket; buckets of more than 64 go on the range stack (LDS).
// Single-wave workgroup, uniform arguments; tmp: n elements of scratch, not aliasing a.
#define KLIB_SMALL_CAP 1024
typedef unsigned long long klib_v2_t __attribute__((ext_vector_type(2), aligned(8)));
struct klib_small_lds_t { int32_t cnt[256]; uint16_t head[256], tail[256], perm[KLIB_SMALL_CAP]; uint8_t byte[KLIB_SMALL_CAP]; uint16_t stk[3 * 16]; }; // 5216 bytes
__device__ __forceinline__ void klib_sort128x_small(mg128_t *a, int32_t n, mg128_t *tmp, klib_small_lds_t *S)
{
	const int lane = threadIdx.x & 63;
	if (n <= 1) return;
	if (n <= 64) { klib_rank_sort64(a, n); return; }
	uint64_t o = 0, an = ~0ULL;
	for (int32_t q0 = lane; q0 < n; q0 += 256) { // four loads in flight
		const int32_t q1 = q0 + 64, q2 = q0 + 128, q3 = q0 + 192;
		const uint64_t x0 = a[q0].x, x1 = a[q1 < n ? q1 : q0].x, x2 = a[q2 < n ? q2 : q0].x, x3 = a[q3 < n ? q3 : q0].x;
		o |= x0 | x1 | x2 | x3, an &= x0 & x1 & x2 & x3;
		S->perm[q0] = (uint16_t)q0;
		if (q1 < n) S->perm[q1] = (uint16_t)q1;
		if (q2 < n) S->perm[q2] = (uint16_t)q2;
		if (q3 < n) S->perm[q3] = (uint16_t)q3;
	}
	for (int d = 32; d > 0; d >>= 1) { o |= __shfl_xor(o, d); an &= __shfl_xor(an, d); }
	if (o == an) return; // all keys equal: no level moves anything
	int top = 1;
	if (lane == 0) { S->stk[0] = 0, S->stk[1] = (uint16_t)n, S->stk[2] = (uint16_t)((63 - __clzll(o ^ an)) & ~7); }
	mga_wave_sync();
	while (top > 0) {
		--top;
		const int32_t b = S->stk[3 * top], e = S->stk[3 * top + 1], sh = S->stk[3 * top + 2];
		const int32_t m = e - b;
		for (int q = lane; q < 256; q += 64) S->cnt[q] = 0;
		mga_wave_sync();
		for (int32_t q0 = b + lane; q0 < e; q0 += 256) {
			const int32_t q1 = q0 + 64, q2 = q0 + 128, q3 = q0 + 192;
			const uint64_t x0 = a[S->perm[q0]].x, x1 = a[S->perm[q1 < e ? q1 : q0]].x, x2 = a[S->perm[q2 < e ? q2 : q0]].x, x3 = a[S->perm[q3 < e ? q3 : q0]].x;
			const int b0 = (int)(x0 >> sh & 0xff), b1 = (int)(x1 >> sh & 0xff), b2 = (int)(x2 >> sh & 0xff), b3 = (int)(x3 >> sh & 0xff);
			S->byte[q0] = (uint8_t)b0, atomicAdd(&S->cnt[b0], 1);
			if (q1 < e) S->byte[q1] = (uint8_t)b1, atomicAdd(&S->cnt[b1], 1);
			if (q2 < e) S->byte[q2] = (uint8_t)b2, atomicAdd(&S->cnt[b2], 1);
			if (q3 < e) S->byte[q3] = (uint8_t)b3, atomicAdd(&S->cnt[b3], 1);
		}
		mga_wave_sync();
		uint64_t ne0, ne1, ne2, ne3;
		bool single;
		{ // bucket bounds: a lane owns 4 consecutive buckets
			const int c0 = S->cnt[4 * lane], c1 = S->cnt[4 * lane + 1], c2 = S->cnt[4 * lane + 2], c3 = S->cnt[4 * lane + 3];
			const int s4 = c0 + c1 + c2 + c3;
			const int at = mga_wave_incl_scan_i32(s4) - s4 + b;
			S->head[4 * lane] = (uint16_t)at, S->tail[4 * lane] = (uint16_t)(at + c0);
			S->head[4 * lane + 1] = (uint16_t)(at + c0), S->tail[4 * lane + 1] = (uint16_t)(at + c0 + c1);
			S->head[4 * lane + 2] = (uint16_t)(at + c0 + c1), S->tail[4 * lane + 2] = (uint16_t)(at + c0 + c1 + c2);
			S->head[4 * lane + 3] = (uint16_t)(at + c0 + c1 + c2), S->tail[4 * lane + 3] = (uint16_t)(at + s4);
			ne0 = __ballot(c0 > 0), ne1 = __ballot(c1 > 0), ne2 = __ballot(c2 > 0), ne3 = __ballot(c3 > 0);
			single = __ballot(c0 == m || c1 == m || c2 == m || c3 == m) != 0;
		}
		mga_wave_sync();
		if (!single && lane == 0) { // the displacement cycles (ksort.h:141-153), bucket 0 first
			uint64_t mm = ne0 | ne1 | ne2 | ne3;
			while (mm) {
				const int l = (int)__builtin_ctzll(mm);
				mm &= mm - 1;
				for (int r = 0; r < 4; ++r) {
					const uint64_t nr = r == 0 ? ne0 : r == 1 ? ne1 : r == 2 ? ne2 : ne3;
					if (!(nr >> l & 1)) continue;
					const int k = 4 * l + r;
					int hk = S->head[k];
					const int tk = S->tail[k];
					while (hk != tk) {
						int cb = S->byte[hk];
						if (cb == k) { ++hk; continue; }
						uint16_t cp = S->perm[hk];
						do {
							const int hl = S->head[cb];
							const int nb = S->byte[hl];
							const uint16_t np = S->perm[hl];
							S->byte[hl] = (uint8_t)cb, S->perm[hl] = cp, S->head[cb] = (uint16_t)(hl + 1);
							cb = nb, cp = np;
						} while (cb != k);
						S->byte[hk] = (uint8_t)cb, S->perm[hk] = cp;
						++hk;
					}
					S->head[k] = (uint16_t)hk;
				}
			}
		}
		mga_wave_sync();
		if (sh > 0) {
			const int32_t nsh = sh - 8;
			int32_t q0 = b;
			while (q0 < e) {
				const int32_t c0 = S->cnt[S->byte[q0]];
				if (c0 > 64) { // next byte
					if (lane == 0) { S->stk[3 * top] = (uint16_t)q0, S->stk[3 * top + 1] = (uint16_t)(q0 + c0), S->stk[3 * top + 2] = (uint16_t)nsh; }
					++top, q0 += c0;
					continue;
				}
				const int32_t q = q0 + lane;
				int32_t c = 65, be = 0;
				if (q < e) { const int k = S->byte[q]; c = S->cnt[k], be = S->tail[k]; }
				const uint64_t fits = __ballot(c <= 64 && be <= q0 + 64);
				const int np = ~fits ? (int)__builtin_ctzll(~fits) : 64; // whole buckets from q0 on (the first one fits: np >= c0)
				const bool mine = lane < np;
				if (__ballot(mine && c >= 2)) {
					const uint16_t pi = mine ? S->perm[q] : 0;
					const uint64_t key = mine ? a[pi].x : 0;
					const int32_t st = be - c;
					const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
					int32_t rank = 0;
					for (int j = 0; j < np; ++j) {
						const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(khi, j) << 32 | (uint32_t)__builtin_amdgcn_readlane(klo, j);
						const int32_t sj = __builtin_amdgcn_readlane(st, j);
						rank += (sj == st && (kj < key || (kj == key && j < lane))) ? 1 : 0;
					}
					mga_wave_sync();
					if (mine) S->perm[st + rank] = pi;
					mga_wave_sync();
				}
				q0 += np;
			}
		}
		mga_wave_sync();
	}
	klib_v2_t *av = (klib_v2_t*)a, *tv = (klib_v2_t*)tmp; // (an element as one 16-byte value)
	for (int32_t q0 = lane; q0 < n; q0 += 256) { // four loads in flight
		const int32_t q1 = q0 + 64, q2 = q0 + 128, q3 = q0 + 192;
		const klib_v2_t x0 = av[S->perm[q0]], x1 = av[S->perm[q1 < n ? q1 : q0]], x2 = av[S->perm[q2 < n ? q2 : q0]], x3 = av[S->perm[q3 < n ? q3 : q0]];
		tv[q0] = x0;
		if (q1 < n) tv[q1] = x1;
		if (q2 < n) tv[q2] = x2;
		if (q3 < n) tv[q3] = x3;
	}
	__syncthreads();
	for (int32_t q0 = lane; q0 < n; q0 += 256) {
		const int32_t q1 = q0 + 64, q2 = q0 + 128, q3 = q0 + 192;
		const klib_v2_t x0 = tv[q0], x1 = tv[q1 < n ? q1 : q0], x2 = tv[q2 < n ? q2 : q0], x3 = tv[q3 < n ? q3 : q0];
		av[q0] = x0;
		if (q1 < n) av[q1] = x1;
		if (q2 < n) av[q2] = x2;
		if (q3 < n) av[q3] = x3;
	}
	__syncthreads();
}

#endif
