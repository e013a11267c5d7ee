/* CPU model of the PACKED windowed WFA forward pass (minigraph_amd/csrc/k_wfa_w.hip: k_wfa_fwp) -- two neighbouring diagonals per lane in the 16-bit halves of one
 * register, the recurrence on packed 16-bit arithmetic, the neighbour exchange as a lane shift + a funnel shift, match masks, traceback rows with reachable diagonals only --
 * followed by the walk of k_wfa_tb (wfw_trace), against the oracle's exact WFA (oracle/mgo_wfa.c): whenever the window decides (score below its bound), score and CIGAR must be
 * the oracle's; cells are stored biased (unreachable = 0), no half may carry into its neighbour; the model must terminate within the bound's steps; the walk must never read a traceback dword that was not written.  Lanes are loops here, registers arrays
 * indexed by lane; every arithmetic step is the kernel's.  Test infrastructure only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mgo.h"

#define BIAS 0x2000   /* cells are stored + 0x2000 per half: "unreachable" is 0 */
#define NEGPK 0u
#define ONEPK 0x00010001u
#define SMAX 256
#define POISON 0xA5A5A5A5u

static int gapc(int n) { if (n < 0) n = -n; if (n == 0) return 0; { int a = 4 + 2 * n, b = 15 + n; return a < b ? a : b; } }
static int window(int W, int tl, int ql, int *L, int cap) /* == wfw_window() of wfa_window.h */
{
	int e = ql - tl, c = e / 2, lo = c - W / 2, hi, blo, bhi, b;
	if (lo < -tl) lo = -tl;
	hi = lo + W - 1;
	if (hi > ql) { hi = ql; lo = hi - W + 1; if (lo < -tl) lo = -tl; }
	*L = lo;
	if (lo > 0 || hi < 0 || e < lo || e > hi) return 0;
	blo = lo - 1 >= -tl ? gapc(lo - 1) + gapc(e - (lo - 1)) : cap;
	bhi = hi + 1 <= ql ? gapc(hi + 1) + gapc(hi + 1 - e) : cap;
	b = blo < bhi ? blo : bhi;
	return b < cap ? b : cap;
}
static int reach(int s) { int a, b; if (s < 6) return 0; a = (s - 4) >> 1, b = s - 15; return a > b ? a : b; }

static uint32_t pk_max(uint32_t a, uint32_t b) { int16_t al = (int16_t)a, ah = (int16_t)(a >> 16), bl = (int16_t)b, bh = (int16_t)(b >> 16); return (uint16_t)(al > bl ? al : bl) | (uint32_t)(uint16_t)(ah > bh ? ah : bh) << 16; }
static uint32_t pk_add(uint32_t a, uint32_t b) { return (uint16_t)((uint16_t)a + (uint16_t)b) | (uint32_t)(uint16_t)((uint16_t)(a >> 16) + (uint16_t)(b >> 16)) << 16; }
static uint32_t pk_lt(uint32_t a, uint32_t b) /* sign masks of the packed (wrapping) differences */
{
	const int16_t dl = (int16_t)((uint16_t)a - (uint16_t)b), dh = (int16_t)((uint16_t)(a >> 16) - (uint16_t)(b >> 16));
	return (dl < 0 ? 0xffffu : 0u) | (dh < 0 ? 0xffff0000u : 0u);
}
static uint32_t sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
static uint32_t alignbit(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }

static int NL = 64; /* lanes a problem has: 64 (k_wfa_fwp), or a group of 16 / 32 (k_wfa_fwq: W = 2 NL) */
#define MAXJP 2
#define MAXSEQ 512
#define MROWS (MAXSEQ / 32 + 3)

typedef struct { int done, score, lst, steps; } fw_res_t;

/* forward pass: region = rows of W dwords (four scores per dword, the earliest in the top byte); returns like the kernel's result record */
static fw_res_t forward(int W, int tl, const char *T, int ql, const char *Q, uint32_t *region, int n_rows)
{
	const int JP = (W + 127) / 128, e = ql - tl;
	NL = W >= 128 ? 64 : W / 2;
	static uint32_t Mk[MROWS + 1][128 * MAXJP];
	uint32_t H[MAXJP][18][64], E1[MAXJP][3][64], F1[MAXJP][3][64], E2[MAXJP][2][64], F2[MAXJP][2][64], okv[MAXJP][64], accA[MAXJP][64], accB[MAXJP][64], fc[MAXJP][64];
	int lo = 0, bnd, s = 0, j, l, a, h, q, row = 0;
	fw_res_t R = { 0, -1, 0, 0 };
	bnd = window(W, tl, ql, &lo, SMAX);
	if (bnd > W + 30) bnd = W + 30;
	if (bnd <= 0) return R;
	memset(Mk, 0, sizeof Mk);
	for (q = 0; q < 2 * JP; ++q) /* match masks: bit b of word w = T[32 w + b] == Q[d + 32 w + b], valid positions only */
		for (l = 0; l < NL; ++l) {
			const int d = lo + 128 * (q >> 1) + 2 * l + (q & 1), kmin = d < 0 ? -d : 0, kmax = tl < ql - d ? tl : ql - d;
			int w, b;
			for (w = 0; w <= (tl >> 5); ++w) {
				uint32_t bits = 0;
				for (b = 0; b < 32; ++b) { const int k = 32 * w + b; if (k >= kmin && k < kmax && T[k] == Q[d + k]) bits |= 1u << b; }
				Mk[w][64 * q + l] = bits;
			}
		}
	for (j = 0; j < JP; ++j)
		for (l = 0; l < NL; ++l) {
			const int dA = lo + 128 * j + 2 * l, dB = dA + 1;
			for (a = 0; a < 18; ++a) H[j][a][l] = NEGPK;
			for (a = 0; a < 3; ++a) E1[j][a][l] = F1[j][a][l] = NEGPK;
			for (a = 0; a < 2; ++a) E2[j][a][l] = F2[j][a][l] = NEGPK;
			okv[j][l] = ((dA >= -tl && dA <= ql && 128 * j + 2 * l < W) ? 0x0000ffffu : 0u) | ((dB >= -tl && dB <= ql && 128 * j + 2 * l + 1 < W) ? 0xffff0000u : 0u);
			accA[j][l] = accB[j][l] = 0;
			if (dA == 0) H[j][2][l] = (uint32_t)(BIAS - 1);
			if (dB == 0) H[j][2][l] = (uint32_t)(BIAS - 1) << 16;
			fc[j][l] = (dA == e ? (uint32_t)(tl - 1 + BIAS) : 0xffffu) | (dB == e ? (uint32_t)(tl - 1 + BIAS) : 0xffffu) << 16;
		}
	for (;;) {
		const int P = s & 1, rs = reach(s), rn = reach(s + 1);
#define HP(j_, a_) H[j_][(a_) + 2 - P]
		int fin = 0, flst = 0;
		uint32_t nH[MAXJP][64], nE1[MAXJP][64], nF1[MAXJP][64], nE2[MAXJP][64], nF2[MAXJP][64];
		if (++R.steps > bnd + 2) { fprintf(stderr, "model does not terminate (W %d tl %d ql %d)\n", W, tl, ql); exit(2); }
		for (j = 0; j < JP; ++j) { /* extension of slice s: both cells of a lane, one loop for runs that fill their window */
			const int b0 = lo + 128 * j;
			uint32_t inv[2][64], n[2][64];
			int tp[2][64], wi[2][64], it, any;
			if (b0 > rs || b0 + 127 < -rs) continue;
			for (l = 0; l < NL; ++l) {
				const uint32_t x = HP(j, 0)[l];
				for (h = 0; h < 2; ++h) {
					const int val = (tp[h][l] = (int)(h ? x >> 16 : x & 0xffffu) + (1 - BIAS), (uint32_t)tp[h][l] <= (uint32_t)tl);
					wi[h][l] = val ? tp[h][l] >> 5 : 0;
					inv[h][l] = ~alignbit(Mk[wi[h][l] + 1][64 * (2 * j + h) + l], Mk[wi[h][l]][64 * (2 * j + h) + l], tp[h][l] & 31);
					if (!val) inv[h][l] = 1u;
					n[h][l] = inv[h][l] ? (uint32_t)__builtin_ctz(inv[h][l]) : 32u;
				}
			}
			for (it = 1;; ++it) {
				for (any = 0, l = 0; l < NL; ++l) any |= inv[0][l] == 0u || inv[1][l] == 0u;
				if (!any) break;
				if (it > MROWS + 4) { fprintf(stderr, "match run does not end (W %d tl %d ql %d)\n", W, tl, ql); exit(2); }
				for (l = 0; l < NL; ++l)
					for (h = 0; h < 2; ++h) {
						const int w = wi[h][l] + it < MROWS - 2 ? wi[h][l] + it : MROWS - 2;
						const uint32_t v = ~alignbit(Mk[w + 1][64 * (2 * j + h) + l], Mk[w][64 * (2 * j + h) + l], tp[h][l] & 31);
						if (inv[h][l] == 0u) n[h][l] += v ? (uint32_t)__builtin_ctz(v) : 32u, inv[h][l] = v;
					}
			}
			for (l = 0; l < NL; ++l) {
				const uint32_t nf = n[0][l] | n[1][l] << 16, xn = HP(j, 0)[l] + nf, df = xn ^ fc[j][l];
				if ((n[0][l] | n[1][l]) >> 15) { fprintf(stderr, "a run overflows its half\n"); exit(2); }
				HP(j, 0)[l] = xn;
				if ((df & 0xffffu) == 0u) fin = 1, flst = (nf & 0xffffu) == 0u ? (int)(accA[j][l] & 7u) : 0;
				else if (df < 0x10000u) fin = 1, flst = (nf >> 16) == 0u ? (int)(accB[j][l] & 7u) : 0;
			}
		}
		if (fin) { R.done = 1, R.score = s, R.lst = flst; break; }
		if (s + 1 >= bnd) break;
		for (j = 0; j < JP; ++j) { /* slice s + 1 */
			const int b0 = lo + 128 * j;
			if (b0 > rn || b0 + 127 < -rn) { for (l = 0; l < NL; ++l) nH[j][l] = nE1[j][l] = nF1[j][l] = nE2[j][l] = nF2[j][l] = NEGPK; continue; }
			for (l = 0; l < NL; ++l) {
#define FROM_L(R_, a_) alignbit(R_[j][a_][l], l > 0 ? R_[j][a_][l - 1] : (j > 0 ? R_[j - 1][a_][NL - 1] : NEGPK), 16)
#define FROM_R(R_, a_) alignbit(l < NL - 1 ? R_[j][a_][l + 1] : (j < JP - 1 ? R_[j + 1][a_][0] : NEGPK), R_[j][a_][l], 16)
				const uint32_t ho1l = FROM_L(H, 5 + 2 - P), e1l = FROM_L(E1, 1), ho2l = FROM_L(H, 15 + 2 - P), e2l = FROM_L(E2, 0);
				const uint32_t ho1r = FROM_R(H, 5 + 2 - P), f1r = FROM_R(F1, 1), ho2r = FROM_R(H, 15 + 2 - P), f2r = FROM_R(F2, 0);
				const uint32_t hx1 = pk_add(HP(j, 3)[l], ONEPK);
				const uint32_t vE1 = pk_max(ho1l, e1l), vE2 = pk_max(ho2l, e2l);
				const uint32_t vF1 = pk_add(pk_max(ho1r, f1r), ONEPK), vF2 = pk_add(pk_max(ho2r, f2r), ONEPK);
				const uint32_t bits = (pk_lt(ho1l, e1l) & 0x00080008u) | (pk_lt(ho2l, e2l) & 0x00200020u) | (pk_lt(ho1r, f1r) & 0x00100010u) | (pk_lt(ho2r, f2r) & 0x00400040u);
				const uint32_t ee = pk_max(vE1, vE2), ff = pk_max(vF1, vF2), hh = pk_max(ee, ff);
				const uint32_t ze = (pk_lt(vE1, vE2) & 0x00020002u) | ONEPK, zf = (pk_lt(vF1, vF2) & 0x00060006u) ^ 0x00020002u;
				uint32_t z = sel(pk_lt(ee, ff), zf, ze), vH, bz;
				z &= pk_lt(hx1, hh);
				vH = pk_max(hx1, hh), bz = bits | z;
				accA[j][l] = accA[j][l] << 8 | (bz & 0xffu), accB[j][l] = accB[j][l] << 8 | (bz >> 16);
				nH[j][l] = vH & okv[j][l], nE1[j][l] = vE1 & okv[j][l], nF1[j][l] = vF1 & okv[j][l], nE2[j][l] = vE2 & okv[j][l], nF2[j][l] = vF2 & okv[j][l];
			}
		}
		for (j = 0; j < JP; ++j) /* every cell stays inside its half with room to spare */
			for (l = 0; l < NL; ++l)
				if ((nH[j][l] & 0xffffu) > BIAS + 600u || (nH[j][l] >> 16) > BIAS + 600u || (nF1[j][l] & 0xffffu) > BIAS + 600u || (nF2[j][l] >> 16) > BIAS + 600u) { fprintf(stderr, "a cell leaves its range (W %d tl %d ql %d)\n", W, tl, ql); exit(2); }
		for (j = 0; j < JP; ++j) /* age shift */
			for (l = 0; l < NL; ++l) {
				HP(j, -1)[l] = nH[j][l];
				if (P == 1) for (a = 17; a > 1; --a) H[j][a][l] = H[j][a - 2][l];
				E1[j][2][l] = E1[j][1][l]; E1[j][1][l] = E1[j][0][l]; E1[j][0][l] = nE1[j][l];
				F1[j][2][l] = F1[j][1][l]; F1[j][1][l] = F1[j][0][l]; F1[j][0][l] = nF1[j][l];
				E2[j][1][l] = E2[j][0][l]; E2[j][0][l] = nE2[j][l];
				F2[j][1][l] = F2[j][0][l]; F2[j][0][l] = nF2[j][l];
			}
		if ((s & 3) == 3) { /* a row is full: reachable diagonals only */
			const int rr = reach(s + 1);
			if (row >= n_rows) { fprintf(stderr, "traceback rows exhausted (W %d)\n", W); exit(2); }
			for (j = 0; j < JP; ++j)
				for (l = 0; l < NL; ++l) {
					const int dA = lo + 128 * j + 2 * l, dB = dA + 1, ad = abs(dA) < abs(dB) ? abs(dA) : abs(dB);
					if (ad <= rr && 128 * j + 2 * l < W) { region[(size_t)row * W + 128 * j + 2 * l] = accA[j][l]; if (128 * j + 2 * l + 1 < W) region[(size_t)row * W + 128 * j + 2 * l + 1] = accB[j][l]; }
				}
			++row;
		}
		++s;
	}
	if (R.done && (s & 3)) {
		const int rr = reach(s + 1);
		for (j = 0; j < JP; ++j)
			for (l = 0; l < NL; ++l) {
				const int dA = lo + 128 * j + 2 * l, dB = dA + 1, ad = abs(dA) < abs(dB) ? abs(dA) : abs(dB);
				if (ad <= rr && 128 * j + 2 * l < W) { region[(size_t)row * W + 128 * j + 2 * l] = accA[j][l] << (8 * (4 - (s & 3))); if (128 * j + 2 * l + 1 < W) region[(size_t)row * W + 128 * j + 2 * l + 1] = accB[j][l] << (8 * (4 - (s & 3))); }
			}
	}
	return R;
}

/* the walk of k_wfa_tb (wfw_trace, phase 0): operators last to first into out[]; returns their number or -1 */
static int trace(int tl, int ql, const char *ts, const char *qs, int S, int last, const uint32_t *reg, int W, int lo, uint32_t *out, int cap)
{
	int i = ql - 1, k = tl - 1, sc = S, n = 0, cur_op = -1, cur_len = 0;
#define PUSH(op_, len_) do { if (cur_op == (op_)) cur_len += (len_); else { if (cur_op >= 0) { if (n < cap) out[n] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n; } cur_op = (op_), cur_len = (len_); } } while (0)
	while (i >= 0 && k >= 0) {
		uint32_t dw, x;
		int state, ext, p;
		if (last == 0) {
			int run = 0;
			while (i >= 0 && k >= 0 && qs[i] == ts[k]) --i, --k, ++run;
			if (run > 0) PUSH(7, run);
			if (i < 0 || k < 0) break;
		}
		if ((uint32_t)((i - k) - lo) >= (uint32_t)W || sc <= 0) return -1;
		p = sc - 1;
		dw = reg[(size_t)(p >> 2) * W + ((i - k) - lo)];
		if (dw == POISON) { fprintf(stderr, "the walk reads a traceback dword that was never written (score %d diagonal %d)\n", sc, i - k); return -2; }
		x = dw >> (8 * (3 - (p & 3))) & 0xffu;
		state = last == 0 ? (int)(x & 7) : last;
		ext = state > 0 ? (int)(x >> (state + 2) & 1) : 0;
		if (state == 0) { PUSH(8, 1); --i, --k, sc -= 4; }
		else if (state == 1) { PUSH(1, 1); --i, sc -= ext ? 2 : 6; }
		else if (state == 3) { PUSH(1, 1); --i, sc -= ext ? 1 : 16; }
		else if (state == 2) { PUSH(2, 1); --k, sc -= ext ? 2 : 6; }
		else { PUSH(2, 1); --k, sc -= ext ? 1 : 16; }
		last = state > 0 && ext ? state : 0;
	}
	if (i >= 0) PUSH(1, i + 1);
	else if (k >= 0) PUSH(2, k + 1);
	PUSH(15, 0);
#undef PUSH
	return n;
}

static uint64_t rng_s = 88172645463325252ULL;
static uint32_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 11); }

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 4000;
	static const int Ws[5] = { 128, 192, 256, 64, 32 }, caps[5] = { 384, 384, 512, 256, 192 };
	long solved[5] = { 0 }, gave_up[5] = { 0 }, mism = 0;
	mgo_wfa_opt_t opt = { 4, 4, 2, 15, 1, 100000000 };
	int it;
	for (it = 0; it < n; ++it) {
		static char t[1400], q[2800];
		static uint32_t c0[8192], c1[8192], region[(SMAX / 4 + 8) * 256];
		int32_t n0, tl = 1 + (int)(rnd() % (it % 3 == 0 ? 500 : 260)), ql = 0, i, k, S;
		const int err = it % 7 == 0 ? 30 : it % 5 == 0 ? 3 : 12; /* percent */
		int64_t iter;
		for (i = 0; i < tl; ++i) t[i] = "ACGT"[rnd() & 3];
		for (i = 0; i < tl; ++i) {
			const uint32_t r = rnd() % 1000;
			if (r < (uint32_t)err * 4) q[ql++] = "ACGT"[rnd() & 3];
			else if (r < (uint32_t)err * 7) { q[ql++] = t[i]; q[ql++] = "ACGT"[rnd() & 3]; }
			else if (r < (uint32_t)err * 10) continue;
			else q[ql++] = t[i];
			if (it % 4 == 0 && i == tl / 2) { int g = 1 + (int)(rnd() % 110); if (rnd() & 1) { while (g-- > 0 && ql < 2500) q[ql++] = "ACGT"[rnd() & 3]; } else i += g; }
		}
		if (it % 97 == 0) ql = 0; /* an empty side */
		if (it % 101 == 0) tl = 1;
		if (ql > 2500) ql = 2500;
		memset(t + tl, 0, 64); memset(q + ql, 0, 64);
		S = mgo_wfa_exact(&opt, tl, t, ql, q, c0, 8192, &n0, &iter);
		for (k = 0; k < 5; ++k) {
			const int W = Ws[k], n_rows = SMAX / 4 + 8;
			int lo, B, m;
			fw_res_t R;
			if (tl > caps[k] || ql > caps[k]) continue;
			B = window(W, tl, ql, &lo, SMAX);
			if (B > W + 30) B = W + 30;
			for (i = 0; i < n_rows * W; ++i) region[i] = POISON;
			R = forward(W, tl, t, ql, q, region, n_rows);
			if (S >= B || B <= 0) { /* the window must not decide */
				if (R.done) { ++mism; fprintf(stderr, "window %d decided score %d >= bound %d (tl %d ql %d)\n", W, R.score, B, tl, ql); }
				++gave_up[k];
				continue;
			}
			if (!R.done || R.score != S) { ++mism; fprintf(stderr, "MISMATCH window %d tl %d ql %d: score %d vs %d (done %d)\n", W, tl, ql, S, R.score, R.done); continue; }
			m = trace(tl, ql, t, q, R.score, R.lst, region, W, lo, c1, 8192);
			if (m != n0) { ++mism; fprintf(stderr, "MISMATCH window %d tl %d ql %d: %d operators vs %d\n", W, tl, ql, m, n0); continue; }
			for (i = 0; i < m; ++i) if (c1[m - 1 - i] != c0[i]) { ++mism; fprintf(stderr, "MISMATCH window %d tl %d ql %d: operator %d\n", W, tl, ql, i); break; }
			++solved[k];
		}
	}
	printf("pairs %d, mismatches %ld; decided / gave up per window:", n, mism);
	for (it = 0; it < 5; ++it) printf(" %d:%ld/%ld", Ws[it], solved[it], gave_up[it]);
	printf("\n");
	return mism != 0;
}
