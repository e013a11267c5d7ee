/* CPU check of the windowed-WFA exactness claim (minigraph_amd/csrc/k_wfa_w.hip, wfa_window.h): for random gap-like sequence pairs, the oracle's exact WFA
 * restricted to a window of W diagonals (cells outside read NEG_INF, the run stops when the score reaches the window's bound) must return the SAME score and
 * CIGAR as the unrestricted oracle whenever it returns at all.  mgo_wfa_win() is oracle/mgo_wfa.c with three lines patched in by tests/test_wfa_window_model.py.
 * Test infrastructure only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mgo.h"
int32_t mgo_wfa_win(const mgo_wfa_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, uint32_t *cigar, int32_t cap, int32_t *n_cigar, int64_t *n_iter_, int32_t WL, int32_t WR, int32_t WB);

static int gapc(int n) { if (n < 0) n = -n; if (n == 0) return 0; { int a = 4 + 2 * n, b = 15 + n; return a < b ? a : b; } }
static int window(int W, int tl, int ql, int *L, int *R, int cap) /* == wfw_window() of wfa_window.h */
{
	int e = ql - tl, c = e / 2, lo = c - W / 2, hi, blo, bhi, b;
	if (lo < -tl) lo = -tl;
	hi = lo + W - 1;
	if (hi > ql) { hi = ql; lo = hi - W + 1; if (lo < -tl) lo = -tl; }
	*L = lo, *R = hi;
	if (lo > 0 || hi < 0 || e < lo || e > hi) return 0;
	blo = lo - 1 >= -tl ? gapc(lo - 1) + gapc(e - (lo - 1)) : cap;
	bhi = hi + 1 <= ql ? gapc(hi + 1) + gapc(hi + 1 - e) : cap;
	b = blo < bhi ? blo : bhi;
	return b < cap ? b : cap;
}
static uint64_t rng_s = 88172645463325252ULL;
static uint32_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 11); }

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 20000;
	static const int Ws[] = { 16, 32, 64, 128, 192, 256, 512 };
	long solved[8] = { 0 }, mism = 0, tried = 0;
	mgo_wfa_opt_t opt = { 4, 4, 2, 15, 1, 100000000 };
	int it;
	for (it = 0; it < n; ++it) {
		char t[1200], q[2600];
		uint32_t c0[4096], c1[4096];
		int32_t n0, n1, tl = 1 + (int)(rnd() % (it % 5 == 0 ? 700 : 160)), ql = 0, i, k, S, first = 7;
		const int err = it % 7 == 0 ? 30 : 10; /* percent */
		int64_t iter;
		for (i = 0; i < tl; ++i) t[i] = "ACGT"[rnd() & 3];
		for (i = 0; i < tl; ++i) { /* substitutions, insertions, deletions; now and then a long indel (a bubble allele) */
			const uint32_t r = rnd() % 1000;
			if (r < (uint32_t)err * 4) q[ql++] = "ACGT"[rnd() & 3];
			else if (r < (uint32_t)err * 7) { q[ql++] = t[i]; q[ql++] = "ACGT"[rnd() & 3]; }
			else if (r < (uint32_t)err * 10) continue;
			else q[ql++] = t[i];
			if (it % 11 == 0 && i == tl / 2) { int g = 1 + (int)(rnd() % 140); if (rnd() & 1) { while (g-- > 0 && ql < 2500) q[ql++] = "ACGT"[rnd() & 3]; } else i += g; }
		}
		if (ql == 0) q[ql++] = 'A';
		S = mgo_wfa_exact(&opt, tl, t, ql, q, c0, 4096, &n0, &iter);
		for (k = 0; k < 7; ++k) {
			int L, R, B = window(Ws[k], tl, ql, &L, &R, k < 6 ? 256 : 0x3fffffff), s1;
			if (S >= B) { /* the window must NOT decide: the model gives up */
				s1 = mgo_wfa_win(&opt, tl, t, ql, q, c1, 4096, &n1, &iter, L, R, B);
				if (s1 >= 0) { ++mism; fprintf(stderr, "window %d decided score %d >= bound %d (tl %d ql %d)\n", Ws[k], s1, B, tl, ql); }
				continue;
			}
			++tried;
			s1 = mgo_wfa_win(&opt, tl, t, ql, q, c1, 4096, &n1, &iter, L, R, B);
			if (s1 != S || n1 != n0 || memcmp(c0, c1, 4 * (size_t)n0)) { ++mism; fprintf(stderr, "MISMATCH window %d tl %d ql %d: score %d vs %d\n", Ws[k], tl, ql, S, s1); }
			if (first == 7) first = k;
		}
		++solved[first];
	}
	printf("pairs %d, windowed runs %ld, mismatches %ld; smallest deciding window:", n, tried, mism);
	for (it = 0; it < 8; ++it) printf(" %d:%ld", it < 7 ? Ws[it] : -1, solved[it]);
	printf("\n");
	return mism != 0;
}
