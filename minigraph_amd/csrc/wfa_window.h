// wfa_window.h -- the window of diagonals in which a gap's alignment is decided exactly, and its score bound (k_wfa_w.hip has the argument).
#ifndef MGA_WFA_WINDOW_H
#define MGA_WFA_WINDOW_H
#include <stdint.h>

#define WFW_SMAX 256 // the windowed tiers of k_wfa_w.hip decide scores < 256 only: below the reference's first band trimming (miniwfa.c:139-169, :420)

__host__ __device__ __forceinline__ int32_t wfw_gap(int32_t n) // cheapest way to move n diagonals (penalties 4 / 4,2 / 15,1: miniwfa.c:11-18)
{
	if (n < 0) n = -n;
	if (n == 0) return 0;
	const int32_t a = 4 + 2 * n, b = 15 + n;
	return a < b ? a : b;
}

// window of W diagonals for a tl x ql problem, centred between diagonal 0 and the end diagonal ql - tl, clipped to the matrix's [-tl, ql];
// returns the bound B (at most cap): an alignment inside [*lo, *lo + W - 1] that scores < B is THE alignment
__host__ __device__ __forceinline__ int32_t wfw_window(int32_t W, int32_t tl, int32_t ql, int32_t *lo_, int32_t cap)
{
	const int32_t e = ql - tl, c = e / 2;
	int32_t lo = c - W / 2, hi;
	if (lo < -tl) lo = -tl;
	hi = lo + W - 1;
	if (hi > ql) { hi = ql; lo = hi - W + 1; if (lo < -tl) lo = -tl; }
	*lo_ = lo;
	if (lo > 0 || hi < 0 || e < lo || e > hi) return 0; /* the start diagonal 0 or the end diagonal lies outside: nothing can be decided in this window */
	const int32_t blo = lo - 1 >= -tl ? wfw_gap(lo - 1) + wfw_gap(e - (lo - 1)) : cap;
	const int32_t bhi = hi + 1 <= ql ? wfw_gap(hi + 1) + wfw_gap(hi + 1 - e) : cap;
	const int32_t b = blo < bhi ? blo : bhi;
	return b < cap ? b : cap;
}
#endif
