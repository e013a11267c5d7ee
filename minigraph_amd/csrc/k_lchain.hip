// k_lchain.hip -- linear chaining of a batch of reads, one wavefront per read.
//
// Replaces mg_lchain_dp() + comput_sc() + mg_chain_backtrack() + compact_a() (reference
// lchain.c:9-219) for single-segment long reads (is_cdna = 0, n_seg = 1).
//
// DP (lchain.c:168-207): f[i] = max_j f[j] + sc(i,j) over predecessors j = i-1 .. st, scanned in
// DESCENDING j with order-dependent heuristics (max_skip / t[] marking / early break) that a plain
// max-reduction would not reproduce.  The wave scores 64 predecessors at a time (lane l <-> j0-l,
// i.e. ascending lane = the reference's visiting order) and replays the sequential semantics:
//   - "sc > max_f"  is an exclusive prefix-max over lanes (__shfl_up scan) seeded with the carry
//   - "t[j] == i"   lanes first publish t[p[j]] = i, then read t[j]: a write to t[j] can only come
//                   from a lane with larger j', i.e. one visited earlier, exactly as in the reference
//   - n_skip        the saturating counter is replayed over the ballot masks of improving / skip-hit
//                   lanes by scalar code; the first lane where it exceeds max_skip cuts the scan
// Float semantics: mg_log2 bit trick, float mul/add without FMA contraction (-ffp-contract=off),
// truncation by (int) -- identical to the reference built with -msse4.
// Backtrack + compaction follow the reference order of operations; the two klib sorts use the exact
// permutation emulation of dev_klibsort.h.
#include <stdio.h>
#include <stdlib.h>
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_klibsort.h"
#include "dev_lcscan.h"
#include <string.h>


// ---- long-join rescue (map-algo.c:407-417 -> mg_lchain_rmq, lchain.c:252-372) -------------------------------
// When the first pass leaves a read in several chains that cover too little of it, the reference re-chains ALL
// chained anchors (x-sorted) with the RMQ chainer and a wide band.  Per anchor i that chainer takes (1) the
// minimum-priority node of an AVL tree over the active anchors with y in (y_i - max_dist, y_i - 1), then (2)
// walks a second tree in DESCENDING (y, index) order over the anchors within max_dist_inner, with the same
// order-dependent skip heuristic as the first pass.  Here:
//   (1) is a wave-wide arg-min over the active window.  The tree's answer equals the arg-min whenever the minimum
//       is unique; when two candidates tie on the (double) priority the result would depend on the AVL shape, so
//       the kernel gives up on that read (flag 2) and the host runs the sequential tree (rmq.c) for it;
//   (2) gathers the <= 64 candidates, rank-sorts them by (y, index) and replays the heuristic from ballot masks
//       exactly like the first pass; more than 64 candidates also defers the read to the host.
// [measured] the rescue fires on ~48 % of 10 kb reads and cost 46 us/read of host CPU, a third of the host budget.
// profiling aid (MGA_LC_PROF=1): cycles per phase summed over reads: [0] first-pass DP, [1] its backtrack + compaction, [2] rescue sort, [3] rescue DP, [4] rescue backtrack
__device__ unsigned long long g_lc_prof[32]; // [8..12] counts: anchors, chain ends, walks, walk steps, backtracks
__device__ int g_lc_prof_on;
// (a read's cycles are summed in LDS and added to the global counters once, when the read is done: [measured, round 6] one atomicAdd per tick from every wavefront on ONE
// address made the ticks inside the walk -- 15 per read -- cost more than the kernel: every barrier waits for the wave's outstanding atomics too)
#define LC_TICK(id) do { if (g_lc_prof_on && lane == 0) { const long long now_ = (long long)clock64(); LP->prof[id] += (unsigned long long)(now_ - tick_); tick_ = now_; } } while (0)
#define LC_COUNT(id, val) do { if (g_lc_prof_on && lane == 0) LP->prof[id] += (unsigned long long)(val); } while (0)
#define LC_RESCUE_DEV_MAX 16384 // chained anchors of a read beyond which the long-join rescue is left to the host tree (~0.5 Mbp of read)
struct lc_rescue_t {
	int32_t enabled;          // bw_long > bw, long-read mode
	int32_t max_dist, max_dist_inner, bw, max_skip, cap, min_cnt, min_sc;
	float pen_gap, pen_skip;
	int32_t rescue_size;
	float rescue_ratio;
	int32_t frag_len, frag_min_gap; // -F: per-read reference gap max(max_frag_len - qlen, max_gap) (map-algo.c:383-386); 0 = off
};

__device__ __forceinline__ int32_t lc_score(uint64_t xi, uint64_t yi, uint64_t xj, uint64_t yj, const mga_lchain_par_t &P) // lchain.c:114-139
{
	const int32_t dq = (int32_t)yi - (int32_t)yj;
	if (dq <= 0 || dq > P.max_dist_x) return LC_NONE;
	const int32_t dr = (int32_t)(xi - xj);
	if (dr == 0 || dq > P.max_dist_y) return LC_NONE;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (dd > P.bw) return LC_NONE;
	const int32_t dg = dr < dq ? dr : dq;
	const int32_t span = (int32_t)(yj >> 32 & 0xff);
	int32_t sc = span < dg ? span : dg;
	if (dd || dg > span) {
		const float lin = P.chn_pen_gap * (float)dd + P.chn_pen_skip * (float)dg;
		const float lg = dd >= 1 ? lc_log2((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	return sc;
}

struct lc_ws_t { int32_t *f, *p, *v, *t; mg128_t *z; };
#define LC_RQ_LDS 336 // chained anchors up to which the rescue's DP keeps y, priority and marks of every anchor in LDS (5 216 bytes with its candidate lists)
// the LDS of a read's wavefront, one phase after the other: the marks of the DP's register window, the two forms of the klib sort, the candidate lists of the rescue's DP.
// 5 344 bytes: 28 single-wave workgroups (7 per SIMD, what the registers allow) fit a CU's 160 KB.
struct lc_lds_t {
	union { int32_t tm[128]; klib_lds_t big; klib_small_lds_t small; struct { int32_t cand_j[64], cand_y[64], sorted_j[64]; } rq;
	        struct { double pri[LC_RQ_LDS]; int32_t y[LC_RQ_LDS]; uint16_t t[LC_RQ_LDS]; int32_t cand_y[64]; uint16_t cand_j[64], sorted_j[64]; } rql; /* lc_dp_rmq_lds */ };
	unsigned long long prof[32]; // MGA_LC_PROF: this read's cycles per phase
};
__device__ __forceinline__ void lc_sort(mg128_t *a, int32_t n, mg128_t *tmp, int32_t *stk, lc_lds_t *L) // klib's radix_sort_128x, permutation and all; tmp: n elements of scratch
{
	if (n <= KLIB_SMALL_CAP) klib_sort128x_small(a, n, tmp, &L->small);
	else klib_sort128x(a, n, stk, &L->big);
}

// ---------------- first-pass DP (lchain.c:168-207) ----------------
__device__ __forceinline__ void lc_dp(const mg128_t *__restrict__ a, int32_t n, mga_lchain_par_t P, lc_ws_t W, int lane)
{
	int32_t *f = W.f, *p = W.p, *v = W.v, *t = W.t;
	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	__syncthreads();
	int32_t st = 0, max_ii = -1;
	for (int32_t i = 0; i < n; ++i) {
		{ // Anchors without any predecessor in reach -- the previous anchor (by x) lies on another (segment, strand) or more than max_dist_x back:
		  // on a multi-gigabase graph more than half of a read's seed hits are such strays -- leave no trace in the sequential state of the loop:
		  // f = v = span, p = -1, and the "best anchor in reach" becomes the anchor itself (the recomputation at lchain.c:191-196 finds nothing, the
		  // update at :203 then takes i).  A run of them is written by the lanes at once instead of costing one trip of the loop each.
			const int32_t k = i + lane;
			bool iso = false;
			uint64_t yk = 0;
			if (k < n) {
				const uint64_t xk = a[k].x;
				yk = a[k].y;
				if (k == 0) iso = true;
				else { const uint64_t xp = a[k - 1].x; iso = xk >> 32 != xp >> 32 || xk > xp + (uint64_t)(int64_t)P.max_dist_x; }
			}
			const uint64_t m = __ballot(iso);
			const int r = (~m) ? (int)__builtin_ctzll(~m) : 64; // leading run of strays
			if (r > 0) {
				if (lane < r) { const int32_t sp = (int32_t)(yk >> 32 & 0xff); f[k] = sp, p[k] = -1, v[k] = sp; }
				max_ii = i + r - 1;
				if (st < i + r - 1) st = i + r - 1; // nothing left of the last stray is in reach of what follows (x ascending)
				i += r - 1;
				__syncthreads();
				continue;
			}
		}
		// Round 4: the loads of an anchor's step that do not depend on each other leave together -- its own record, the window's first anchor and the best-scoring anchor in
		// reach (known from the previous step) at the top; a predecessor block's anchors with their f, p AND v in one trip (they used to follow the score test) --
		// [measured] ~10 k cycles per anchor were 6-8 DEPENDENT trips to memory at 4 % VALU utilisation.
		const uint64_t xi = a[i].x, yi = a[i].y;
		uint64_t xs0 = st < i ? a[st].x : 0;
		mg128_t am; am.x = am.y = 0;
		int32_t fm = 0, vm = 0;
		if (max_ii >= 0) am = a[max_ii], fm = f[max_ii], vm = v[max_ii];
		while (st < i) { // lchain.c:171
			if (xi >> 32 != xs0 >> 32 || xi > xs0 + (uint64_t)(int64_t)P.max_dist_x) { ++st; if (st < i) xs0 = a[st].x; } else break;
		}
		if (i - st > P.max_iter) st = i - P.max_iter;
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1, max_v = 0, n_skip = 0, end_j = st - 1;
		bool cut = false;
		for (int32_t j0 = i - 1; j0 >= st && !cut; j0 -= 64) {
			const int32_t j = j0 - lane;
			const bool act = j >= st;
			int32_t sc = LC_NONE, pj = -1, vj = 0;
			if (act) {
				const mg128_t aj = a[j];
				const int32_t fj = f[j];
				pj = p[j], vj = v[j];
				sc = lc_score(xi, yi, aj.x, aj.y, P);
				if (sc != LC_NONE) sc += fj; else pj = -1;
			}
			const bool valid = sc != LC_NONE;
			if (valid && pj >= 0) t[pj] = i; // lchain.c:188 (harmless beyond the cut: only compared against this i)
			__syncthreads();
			const bool hit_t = valid && t[j] == i;
			// exclusive prefix max of valid scores in visiting order, seeded with the running max_f
			const int32_t pm = lc_scan_max(valid ? sc : INT32_MIN, INT32_MIN);
			int32_t ex = lc_prev_lane(pm, INT32_MIN);
			if (ex < max_f) ex = max_f;
			const bool improve = valid && sc > ex;
			const uint64_t m_imp = __ballot(improve);
			const int cut_lane = lc_skip_replay(improve, hit_t && !improve, P.max_skip, &n_skip);
			const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
			const uint64_t imp_b = m_imp & before;
			if (imp_b) {
				const int bl = 63 - __clzll(imp_b);
				max_f = __shfl(sc, bl), max_v = __shfl(vj, bl), max_j = j0 - bl;
			}
			if (cut_lane < 64) { cut = true; end_j = j0 - cut_lane; }
			__syncthreads();
		}
		// lchain.c:191-196: best-scoring anchor within reach, recomputed when it fell out of range
		if (max_ii < 0 || xi - am.x > (uint64_t)(int64_t)P.max_dist_x) {
			int32_t bf = INT32_MIN, bj = -1;
			for (int32_t j = i - 1 - lane; j >= st; j -= 64) { const int32_t fj = f[j]; if (bf < fj) bf = fj, bj = j; } // descending j per lane: first max kept
			for (int d = 32; d > 0; d >>= 1) {
				const int32_t of = __shfl_xor(bf, d), oj = __shfl_xor(bj, d);
				if (of > bf || (of == bf && oj > bj)) bf = of, bj = oj; // ties: the larger j was met first
			}
			max_ii = bj;
			if (max_ii >= 0) am = a[max_ii], fm = f[max_ii], vm = v[max_ii];
		}
		if (max_ii >= 0 && max_ii < end_j) { // lchain.c:197-201
			const int32_t tmp = lc_score(xi, yi, am.x, am.y, P);
			if (tmp != LC_NONE && max_f < tmp + fm) max_f = tmp + fm, max_j = max_ii, max_v = vm;
		}
		int32_t vi = max_f;
		if (max_j >= 0 && max_v > max_f) vi = max_v;
		if (lane == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; }
		if (max_ii < 0 || (xi - am.x <= (uint64_t)(int64_t)P.max_dist_x && fm < max_f)) max_ii = i;
		__syncthreads();
	}
}

// ---------------- first-pass DP, round 6: the last 64 anchors in REGISTERS ----------------
// [measured, round 6, profiles/r06d_lchain_occupancy.txt] the kernel is bound by the LATENCY of its dependent global round trips, not by instruction issue: forced to 4 / 2
// resident waves per SIMD instead of 7 it takes 82.7 / 128 ms instead of 69.3 while the cycles a wave spends in each phase fall by only 10 / 22 % -- a wave waits, whoever
// shares its SIMD.  lc_dp() pays ~6 such trips per anchor: the anchor, the window's first anchor (one trip per anchor it moves by), the best anchor in reach, a block of 64
// predecessors' a / f / p / v, the t[] mark store -> load, and the f / p / v store the next anchor reads back.  Here the window of the last 64 anchors IS a set of registers --
// lane l holds anchor i - 1 - l: x, y, span, f, p, v -- shifted by one lane per anchor (DPP wave_shr:1, the new anchor enters at lane 0):
//   * the first block of predecessors (26 are visited on average, [measured] round 5) needs no load at all; marks t[p[j]] = i whose target is inside the window go through an
//     LDS ring, the others to global memory as before;
//   * the window's start moves by a ballot over the lanes (anchors are x-sorted: the in-reach lanes are a prefix), the best anchor in reach is read from a lane;
//   * the anchors themselves arrive 64 at a time (one coalesced load per 64 steps), with their "no predecessor in reach" flags as one ballot mask per block;
//   * f / p / v are still stored per anchor (the backtrack and blocks beyond the window read them) but nothing waits for the store.
// Anything further back than 64 anchors -- windows in repeats, max_iter-long scans -- takes lc_dp()'s global-memory block loop unchanged, behind a wait for the stores.
// Same replay of the reference's heuristics, same values (tests/test_gpu_stages.py: test_lchain_*, run in both forms).
struct lcw_win_t { int32_t xl, xh, yl, sp, f, p, v; };
#define LCW_NOSEG ((int32_t)0xffffffff) /* x >> 32 of an empty window slot: no anchor has it (it would be segment 2^31 - 1, reverse strand) */

__device__ __forceinline__ int32_t lcw_score(int32_t xil, int32_t yil, int32_t xjl, int32_t yjl, int32_t span, const mga_lchain_par_t &P) // lc_score() on the low words (same segment and strand)
{
	const int32_t dq = yil - yjl;
	if (dq <= 0 || dq > P.max_dist_x) return LC_NONE;
	const int32_t dr = xil - xjl;
	if (dr == 0 || dq > P.max_dist_y) return LC_NONE;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (dd > P.bw) return LC_NONE;
	const int32_t dg = dr < dq ? dr : dq;
	int32_t sc = span < dg ? span : dg;
	if (dd || dg > span) {
		const float lin = P.chn_pen_gap * (float)dd + P.chn_pen_skip * (float)dg;
		const float lg = dd >= 1 ? lc_log2((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	return sc;
}

__device__ __forceinline__ void lc_dp_w(const mg128_t *__restrict__ a, int32_t n, mga_lchain_par_t P, lc_ws_t W, lc_lds_t *L, int lane)
{
	int32_t *tm = L->tm; // marks of the window's anchors: tm[j & 127] == i <=> t[j] == i
	int32_t *f = W.f, *p = W.p, *v = W.v, *t = W.t;
	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	tm[lane] = -1, tm[64 + lane] = -1;
	lcw_win_t w; w.xl = 0, w.xh = LCW_NOSEG, w.yl = 0, w.sp = 0, w.f = 0, w.p = -1, w.v = 0;
	int32_t nfill = 0;                 // anchors in the window: [i - nfill, i)
	int32_t ib = -64;                  // the block of anchors in (bxl, bxh, byl, bsp): lane l holds anchor ib + l
	int32_t bxl = 0, bxh = 0, byl = 0, bsp = 0, carry_xl = 0, carry_xh = LCW_NOSEG;
	uint64_t m_iso = 0;
	int32_t st = 0, max_ii = -1;
	uint64_t am_x = 0; int32_t am_yl = 0, am_sp = 0, fm = 0, vm = 0; // the best-scoring anchor in reach (lchain.c: max_ii), kept with the values memory would give
	__syncthreads();
	for (int32_t i = 0; i < n; ++i) {
		if (i >= ib + 64) { // next block of anchors + its stray flags
			ib = i;
			const int32_t k = ib + lane;
			mg128_t ak; ak.x = ak.y = 0;
			if (k < n) ak = a[k];
			bxl = (int32_t)ak.x, bxh = (int32_t)(ak.x >> 32), byl = (int32_t)ak.y, bsp = (int32_t)(ak.y >> 32 & 0xff);
			const int32_t pxl = lc_prev_lane(bxl, carry_xl), pxh = lc_prev_lane(bxh, carry_xh); // anchor k - 1
			const bool iso = k < n && (k == 0 || bxh != pxh || (uint32_t)(bxl - pxl) > (uint32_t)P.max_dist_x);
			m_iso = __ballot(iso);
			carry_xl = __builtin_amdgcn_readlane(bxl, 63), carry_xh = __builtin_amdgcn_readlane(bxh, 63);
		}
		const int o = i - ib;
		{ // a run of anchors without any predecessor in reach (see lc_dp): f = v = span, p = -1, written by the lanes at once; only the last of them can precede what follows
			const uint64_t rest = ~(m_iso >> o);
			const int r = rest ? (int)__builtin_ctzll(rest) : 64;
			if (r > 0) {
				if (lane >= o && lane < o + r) { const int32_t k = ib + lane; f[k] = bsp, p[k] = -1, v[k] = bsp; }
				const int ls = o + r - 1;
				const int32_t sxl = __builtin_amdgcn_readlane(bxl, ls), sxh = __builtin_amdgcn_readlane(bxh, ls), syl = __builtin_amdgcn_readlane(byl, ls), ssp = __builtin_amdgcn_readlane(bsp, ls);
				w.xl = sxl, w.yl = syl, w.sp = ssp, w.f = ssp, w.p = -1, w.v = ssp;
				w.xh = lane == 0 ? sxh : LCW_NOSEG;
				nfill = 1;
				max_ii = i + r - 1;
				am_x = (uint64_t)(uint32_t)sxh << 32 | (uint32_t)sxl, am_yl = syl, am_sp = ssp, fm = ssp, vm = ssp;
				if (st < i + r - 1) st = i + r - 1;
				i += r - 1;
				continue;
			}
		}
		const int32_t xil = __builtin_amdgcn_readlane(bxl, o), xih = __builtin_amdgcn_readlane(bxh, o), yil = __builtin_amdgcn_readlane(byl, o), spi = __builtin_amdgcn_readlane(bsp, o);
		const uint64_t xi = (uint64_t)(uint32_t)xih << 32 | (uint32_t)xil;
		// lchain.c:171: the window's start.  In reach: same segment and strand, not more than max_dist_x back; anchors are x-sorted, so the lanes in reach are a prefix
		bool deep = false; // predecessors beyond the register window
		{
			const bool in = w.xh == xih && (uint32_t)(xil - w.xl) <= (uint32_t)P.max_dist_x;
			const uint64_t m_out = ~__ballot(in);
			const int r = m_out ? (int)__builtin_ctzll(m_out) : 64;
			if (r < 64 || i - 64 <= st) { if (st < i - r) st = i - r; }
			else { // the whole window is in reach and the start lies before it: the reference's loop over memory, up to the window's first anchor
				__syncthreads();
				while (st < i - 64) { const uint64_t xs = a[st].x; if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)P.max_dist_x) ++st; else break; }
			}
			if (i - st > P.max_iter) st = i - P.max_iter;
			deep = i - st > 64;
		}
		int32_t max_f = spi, max_j = -1, max_v = 0, n_skip = 0, end_j = st - 1;
		bool cut = false;
		{ // the first block of predecessors: lane l <-> anchor i - 1 - l, from the registers
			const int32_t j = i - 1 - lane;
			const bool act = j >= st;
			int32_t sc = LC_NONE, pj = -1;
			if (act) {
				sc = lcw_score(xil, yil, w.xl, w.yl, w.sp, P);
				if (sc != LC_NONE) sc += w.f, pj = w.p;
			}
			const bool valid = sc != LC_NONE;
			if (valid && pj >= 0) { if (i - 1 - pj < 64) tm[pj & 127] = i; else t[pj] = i; } // lchain.c:188
			mga_wave_sync();
			const bool hit_t = valid && tm[j & 127] == i;
			const int32_t pm = lc_scan_max(valid ? sc : INT32_MIN, INT32_MIN);
			int32_t ex = lc_prev_lane(pm, INT32_MIN);
			if (ex < max_f) ex = max_f;
			const bool improve = valid && sc > ex;
			const uint64_t m_imp = __ballot(improve);
			const int cut_lane = lc_skip_replay(improve, hit_t && !improve, P.max_skip, &n_skip);
			const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
			const uint64_t imp_b = m_imp & before;
			if (imp_b) {
				const int bl = 63 - __clzll(imp_b);
				max_f = __builtin_amdgcn_readlane(sc, bl), max_v = __builtin_amdgcn_readlane(w.v, bl), max_j = i - 1 - bl;
			}
			if (cut_lane < 64) { cut = true; end_j = i - 1 - cut_lane; }
		}
		if (deep && !cut) { // further back than the window: lc_dp()'s block loop over memory (the stores of f / p / v and of the marks are waited for first)
			__syncthreads();
			for (int32_t j0 = i - 65; j0 >= st && !cut; j0 -= 64) {
				const int32_t j = j0 - lane;
				const bool act = j >= st;
				int32_t sc = LC_NONE, pj = -1, vj = 0;
				if (act) {
					const mg128_t aj = a[j];
					const int32_t fj = f[j];
					pj = p[j], vj = v[j];
					sc = lc_score(xi, (uint64_t)(uint32_t)yil | (uint64_t)spi << 32, aj.x, aj.y, P);
					if (sc != LC_NONE) sc += fj; else pj = -1;
				}
				const bool valid = sc != LC_NONE;
				if (valid && pj >= 0) t[pj] = i;
				__syncthreads();
				const bool hit_t = valid && t[j] == i;
				const int32_t pm = lc_scan_max(valid ? sc : INT32_MIN, INT32_MIN);
				int32_t ex = lc_prev_lane(pm, INT32_MIN);
				if (ex < max_f) ex = max_f;
				const bool improve = valid && sc > ex;
				const uint64_t m_imp = __ballot(improve);
				const int cut_lane = lc_skip_replay(improve, hit_t && !improve, P.max_skip, &n_skip);
				const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
				const uint64_t imp_b = m_imp & before;
				if (imp_b) {
					const int bl = 63 - __clzll(imp_b);
					max_f = __shfl(sc, bl), max_v = __shfl(vj, bl), max_j = j0 - bl;
				}
				if (cut_lane < 64) { cut = true; end_j = j0 - cut_lane; }
				__syncthreads();
			}
		}
		// lchain.c:191-196: best-scoring anchor within reach, recomputed when it fell out of range
		if (max_ii < 0 || xi - am_x > (uint64_t)(int64_t)P.max_dist_x) {
			if (!deep) { // the candidates are the window's lanes [0, i - st): largest f, ties to the larger j = the smaller lane
				const bool act = i - 1 - lane >= st;
				const int32_t top = __builtin_amdgcn_readlane(lc_scan_max(act ? w.f : INT32_MIN, INT32_MIN), 63);
				const uint64_t m_top = __ballot(act && w.f == top);
				if (m_top) {
					const int lm = (int)__builtin_ctzll(m_top);
					max_ii = i - 1 - lm;
					am_x = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(w.xh, lm) << 32 | (uint32_t)__builtin_amdgcn_readlane(w.xl, lm);
					am_yl = __builtin_amdgcn_readlane(w.yl, lm), am_sp = __builtin_amdgcn_readlane(w.sp, lm), fm = top, vm = __builtin_amdgcn_readlane(w.v, lm);
				} else max_ii = -1;
			} else {
				__syncthreads();
				int32_t bf = INT32_MIN, bj = -1;
				for (int32_t j = i - 1 - lane; j >= st; j -= 64) { const int32_t fj = f[j]; if (bf < fj) bf = fj, bj = j; } // descending j per lane: first max kept
				for (int d = 32; d > 0; d >>= 1) {
					const int32_t of = __shfl_xor(bf, d), oj = __shfl_xor(bj, d);
					if (of > bf || (of == bf && oj > bj)) bf = of, bj = oj; // ties: the larger j was met first
				}
				max_ii = bj;
				if (max_ii >= 0) { const mg128_t am = a[max_ii]; am_x = am.x, am_yl = (int32_t)am.y, am_sp = (int32_t)(am.y >> 32 & 0xff), fm = f[max_ii], vm = v[max_ii]; }
			}
		}
		if (max_ii >= 0 && max_ii < end_j) { // lchain.c:197-201
			const int32_t tmp = lc_score(xi, (uint64_t)(uint32_t)yil | (uint64_t)spi << 32, am_x, (uint64_t)(uint32_t)am_yl | (uint64_t)am_sp << 32, P);
			if (tmp != LC_NONE && max_f < tmp + fm) max_f = tmp + fm, max_j = max_ii, max_v = vm;
		}
		int32_t vi = max_f;
		if (max_j >= 0 && max_v > max_f) vi = max_v;
		if (lane == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; }
		if (max_ii < 0 || (xi - am_x <= (uint64_t)(int64_t)P.max_dist_x && fm < max_f)) max_ii = i, am_x = xi, am_yl = yil, am_sp = spi, fm = max_f, vm = vi;
		// the anchor enters the window at lane 0, everything else moves up a lane
		w.xl = lc_prev_lane(w.xl, xil), w.xh = lc_prev_lane(w.xh, xih), w.yl = lc_prev_lane(w.yl, yil), w.sp = lc_prev_lane(w.sp, spi);
		w.f = lc_prev_lane(w.f, max_f), w.p = lc_prev_lane(w.p, max_j), w.v = lc_prev_lane(w.v, vi);
		if (nfill < 64) ++nfill;
	}
	__syncthreads(); // (every store has landed before the backtrack reads f / p / v)
}

// ---------------- the same with y, priority and marks of every anchor in LDS (round 6; n <= LC_RQ_LDS) ----------------
// [measured, round 6, profiles/r06f_lchain_phases.txt] lc_dp_rmq() below was 43 % of k_lchain once the first pass and the backtrack had lost their round trips: per anchor
// ~20 dependent trips to global memory -- the anchor, the first anchor of its x-group, the priorities of the group that became available (store -> barrier), the two windows'
// first anchors, a block loop over the window for the range minimum and another for the inner candidates, the minimum's record, the candidates' records, the marks
// (store -> barrier -> load), v of the winner, the result (store -> barrier).  The rescue re-chains a read's CHAINED anchors: 250 on average, so everything the two window
// loops read fits the wavefront's LDS: y and the priority of every anchor (the priority of anchor i is stored when f[i] is known -- the loops only look below i0, so an anchor
// whose x-group is still open is never seen early), the marks as 16-bit anchor numbers.  Left in global memory: the anchors' records and f / p / v, read once per anchor for
// the minimum and once for the <= 64 candidates (p, f, v travel with the record; the winner's v comes from the lane that holds it), and the windows' first anchors, re-read
// only when a window moves.  Same values, same ties (-> false: the host's tree), same candidate cap.
__device__ __forceinline__ bool lc_dp_rmq_lds(const mg128_t *__restrict__ a, int32_t n, const lc_rescue_t &R, lc_ws_t W, lc_lds_t *L, int lane)
{
	double *lpri = L->rql.pri;
	int32_t *ly = L->rql.y, *cand_y = L->rql.cand_y;
	uint16_t *lt = L->rql.t, *cand_j = L->rql.cand_j, *sorted_j = L->rql.sorted_j;
	int32_t *f = W.f, *p = W.p, *v = W.v;
	int32_t max_dist = R.max_dist, max_dist_inner = R.max_dist_inner;
	if (max_dist < R.bw) max_dist = R.bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	if (n > R.cap) return false; // the reference then evicts by tree size
	for (int32_t i = lane; i < n; i += 64) lt[i] = 0xffff;
	mga_wave_sync();
	int32_t i0 = 0, st = 0, st_in = 0;
	mg128_t cur = a[0];
	uint64_t x_i0 = cur.x, xs_st = cur.x, xs_in = cur.x; // x of the anchors i0, st, st_in
	uint64_t xs_st1 = n > 1 ? a[1].x : cur.x, xs_in1 = xs_st1; // ... and of st + 1, st_in + 1: a window that moves by one anchor per step (the usual case) waits for nothing
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t xi = cur.x, yi = cur.y;
		const int32_t yi32 = (int32_t)yi;
		if (i + 1 < n) cur = a[i + 1]; // (in flight while this anchor is chained)
		if (i0 < i && x_i0 != xi) i0 = i, x_i0 = xi; // anchors with a smaller x are available now (lchain.c:279-293); their priorities are in LDS already
		// windows (lchain.c:294-312); the trees hold [st, i0) and [st_in, i0)
		while (st < i) { if (xi >> 32 != xs_st >> 32 || xi > xs_st + (uint64_t)(int64_t)max_dist) { ++st; xs_st = xs_st1; xs_st1 = st + 1 < n ? a[st + 1].x : xi; } else break; }
		if (max_dist_inner > 0)
			while (st_in < i) { if (xi >> 32 != xs_in >> 32 || xi > xs_in + (uint64_t)(int64_t)max_dist_inner) { ++st_in; xs_in = xs_in1; xs_in1 = st_in + 1 < n ? a[st_in + 1].x : xi; } else break; }
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1, max_vj = 0;
		// (1) range-minimum query: keys in [(y_i - max_dist, INT32_MAX), (y_i - 1, 0)] (lchain.c:313-316)
		const int32_t ylo = yi32 - max_dist, yhi = yi32 - 1;
		double bp = 0.0;
		int32_t bj = -1;
		bool tie = false;
		for (int32_t j = st + lane; j < i0; j += 64) {
			const int32_t yj = ly[j];
			const double pj = lpri[j];
			if ((yj > ylo && yj < yhi) || (j == 0 && yj == yhi)) {
				if (bj < 0 || pj < bp) bp = pj, bj = j, tie = false;
				else if (pj == bp) tie = true;
			}
		}
		const uint64_t has = __ballot(bj >= 0);
		if (has) {
			double m = bp;
			bool hm = bj >= 0;
			for (int d = 32; d > 0; d >>= 1) {
				const double om = __shfl_xor(m, d);
				const bool oh = __shfl_xor((int)hm, d) != 0;
				if (oh && (!hm || om < m)) m = om, hm = true;
			}
			const uint64_t at_min = __ballot(bj >= 0 && bp == m);
			if (__popcll(at_min) > 1 || __ballot(bj >= 0 && bp == m && tie)) return false; // equal priorities: the AVL shape would decide
			const int32_t jq = __builtin_amdgcn_readlane(bj, (int)__builtin_ctzll(at_min));
			bool exact;
			int32_t width;
			__syncthreads(); // f / p / v of the anchors before this one have landed (their stores had the LDS phase above to do so)
			const mg128_t aq = a[jq];
			const int32_t fq = f[jq], vq = v[jq];
			const int32_t sc = fq + lc_score_simple(xi, yi, aq.x, aq.y, R.pen_gap, R.pen_skip, &exact, &width);
			if (width <= R.bw && sc > max_f) max_f = sc, max_j = jq, max_vj = vq;
			// (2) inner window in descending (y, index) order (lchain.c:321-350)
			if (!exact && max_dist_inner > 0 && st_in < i0 && yi32 > 0) {
				const int32_t ymin = yi32 - max_dist_inner;
				int32_t m_c = 0;
				for (int32_t j0 = st_in; j0 < i0; j0 += 64) {
					const int32_t j = j0 + lane;
					int32_t yj = 0;
					bool c = false;
					if (j < i0) { yj = ly[j]; c = yj <= yhi && yj >= ymin; }
					const uint64_t mc = __ballot(c);
					const int32_t pos = m_c + __popcll(mc & mga_lanemask_lt());
					if (c && pos < 64) cand_j[pos] = (uint16_t)j, cand_y[pos] = yj;
					m_c += __popcll(mc);
				}
				if (m_c > 64) return false; // rank sort below handles one wave of candidates
				mga_wave_sync();
				if (m_c > 0) {
					// rank by descending (y, j): keys are unique; every lane reads the others' keys from their registers
					const int32_t myj = lane < m_c ? (int32_t)cand_j[lane] : -1, myy = lane < m_c ? cand_y[lane] : 0;
					int32_t rank = 0;
					for (int32_t k = 0; k < m_c; ++k) {
						const int32_t ky = __builtin_amdgcn_readlane(myy, k), kj = __builtin_amdgcn_readlane(myj, k);
						rank += (ky > myy || (ky == myy && kj > myj)) ? 1 : 0;
					}
					if (lane < m_c) sorted_j[rank] = (uint16_t)myj;
					mga_wave_sync();
					const int32_t j = lane < m_c ? (int32_t)sorted_j[lane] : -1;
					int32_t sc2 = LC_NONE, pj = -1, vj = 0;
					bool valid = false;
					if (j >= 0) {
						bool ex2;
						int32_t w2;
						const mg128_t aj = a[j];
						const int32_t fj = f[j];
						pj = p[j], vj = v[j];
						sc2 = fj + lc_score_simple(xi, yi, aj.x, aj.y, R.pen_gap, R.pen_skip, &ex2, &w2);
						valid = w2 <= R.bw;
					}
					if (valid && pj >= 0) lt[pj] = (uint16_t)i; // marks only reach candidates with a smaller y, i.e. visited later
					mga_wave_sync();
					const bool hit_t = valid && lt[j] == (uint16_t)i;
					const int32_t pm = lc_scan_max(valid ? sc2 : INT32_MIN, INT32_MIN);
					int32_t exm = lc_prev_lane(pm, INT32_MIN);
					if (exm < max_f) exm = max_f;
					const bool improve = valid && sc2 > exm;
					const uint64_t m_imp = __ballot(improve);
					int32_t n_skip = 0;
					const int cut_lane = lc_skip_replay(improve, hit_t && !improve, R.max_skip, &n_skip);
					const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
					const uint64_t imp_b = m_imp & before;
					if (imp_b) {
						const int bl = 63 - __clzll(imp_b);
						max_f = __builtin_amdgcn_readlane(sc2, bl), max_j = __builtin_amdgcn_readlane(j, bl), max_vj = __builtin_amdgcn_readlane(vj, bl);
					}
				}
				mga_wave_sync();
			}
		}
		int32_t vi = max_f;
		if (max_j >= 0 && max_vj > max_f) vi = max_vj;
		if (lane == 0) {
			f[i] = max_f; p[i] = max_j; v[i] = vi;
			ly[i] = yi32, lpri[i] = -((double)max_f + 0.5 * (double)R.pen_gap * (double)((int32_t)xi + yi32));
		}
		mga_wave_sync(); // (LDS: the next anchor's windows; the stores are waited for where the next record is read)
	}
	__syncthreads();
	return true;
}

// ---------------- RMQ DP of the rescue (lchain.c:275-357); false = this read must be re-chained by the host ----------------
__device__ __forceinline__ bool lc_dp_rmq(const mg128_t *__restrict__ a, int32_t n, const lc_rescue_t &R, lc_ws_t W, lc_lds_t *L, int lane)
{
	int32_t *cand_j = L->rq.cand_j, *cand_y = L->rq.cand_y, *sorted_j = L->rq.sorted_j;
	int32_t *f = W.f, *p = W.p, *v = W.v, *t = W.t;
	double *pri = (double*)W.z; // z is free until the backtrack
	int32_t max_dist = R.max_dist, max_dist_inner = R.max_dist_inner;
	if (max_dist < R.bw) max_dist = R.bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	if (n > R.cap) return false; // the reference then evicts by tree size
	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	__syncthreads();
	int32_t i0 = 0, st = 0, st_in = 0;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t xi = a[i].x, yi = a[i].y;
		const int32_t yi32 = (int32_t)yi;
		// anchors with a smaller x become available (lchain.c:279-293): their priorities are final now
		if (i0 < i && a[i0].x != xi) {
			for (int32_t j = i0 + lane; j < i; j += 64)
				pri[j] = -((double)f[j] + 0.5 * (double)R.pen_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y));
			i0 = i;
			__syncthreads();
		}
		// windows (lchain.c:294-312); the trees hold [st, i0) and [st_in, i0)
		while (st < i) { const uint64_t xs = a[st].x; if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)max_dist) ++st; else break; }
		if (max_dist_inner > 0)
			while (st_in < i) { const uint64_t xs = a[st_in].x; if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)max_dist_inner) ++st_in; else break; }
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1;
		// (1) range-minimum query: keys in [(y_i - max_dist, INT32_MAX), (y_i - 1, 0)] (lchain.c:313-316)
		const int32_t ylo = yi32 - max_dist, yhi = yi32 - 1;
		double bp = 0.0;
		int32_t bj = -1;
		bool tie = false;
		for (int32_t j = st + lane; j < i0; j += 64) {
			const int32_t yj = (int32_t)a[j].y;
			const double pj = pri[j]; // (with its anchor, not behind the window test: one trip per block instead of two)
			if ((yj > ylo && yj < yhi) || (j == 0 && yj == yhi)) {
				if (bj < 0 || pj < bp) bp = pj, bj = j, tie = false;
				else if (pj == bp) tie = true;
			}
		}
		const uint64_t has = __ballot(bj >= 0);
		if (has) {
			double m = bp;
			bool hm = bj >= 0;
			for (int d = 32; d > 0; d >>= 1) {
				const double om = __shfl_xor(m, d);
				const bool oh = __shfl_xor((int)hm, d) != 0;
				if (oh && (!hm || om < m)) m = om, hm = true;
			}
			const uint64_t at_min = __ballot(bj >= 0 && bp == m);
			if (__popcll(at_min) > 1 || __ballot(bj >= 0 && bp == m && tie)) return false; // equal priorities: the AVL shape would decide
			const int32_t jq = __shfl(bj, (int)__builtin_ctzll(at_min));
			bool exact;
			int32_t width;
			const mg128_t aq = a[jq];
			const int32_t sc = f[jq] + lc_score_simple(xi, yi, aq.x, aq.y, R.pen_gap, R.pen_skip, &exact, &width);
			if (width <= R.bw && sc > max_f) max_f = sc, max_j = jq;
			// (2) inner window in descending (y, index) order (lchain.c:321-350)
			if (!exact && max_dist_inner > 0 && st_in < i0 && yi32 > 0) {
				const int32_t ymin = yi32 - max_dist_inner;
				int32_t m_c = 0;
				for (int32_t j0 = st_in; j0 < i0; j0 += 64) {
					const int32_t j = j0 + lane;
					int32_t yj = 0;
					bool c = false;
					if (j < i0) { yj = (int32_t)a[j].y; c = yj <= yhi && yj >= ymin; }
					const uint64_t mc = __ballot(c);
					const int32_t pos = m_c + __popcll(mc & mga_lanemask_lt());
					if (c && pos < 64) cand_j[pos] = j, cand_y[pos] = yj;
					m_c += __popcll(mc);
				}
				if (m_c > 64) return false; // rank sort below handles one wave of candidates
				__syncthreads();
				if (m_c > 0) {
					// rank by descending (y, j): keys are unique
					const int32_t myj = lane < m_c ? cand_j[lane] : -1, myy = lane < m_c ? cand_y[lane] : 0;
					int32_t rank = 0;
					for (int32_t k = 0; k < m_c; ++k) {
						const int32_t ky = cand_y[k], kj = cand_j[k];
						rank += (ky > myy || (ky == myy && kj > myj)) ? 1 : 0;
					}
					if (lane < m_c) sorted_j[rank] = myj;
					__syncthreads();
					const int32_t j = lane < m_c ? sorted_j[lane] : -1;
					int32_t sc2 = LC_NONE, pj = -1;
					bool valid = false;
					if (j >= 0) {
						bool ex2;
						int32_t w2;
						const mg128_t aj = a[j];
						sc2 = f[j] + lc_score_simple(xi, yi, aj.x, aj.y, R.pen_gap, R.pen_skip, &ex2, &w2);
						valid = w2 <= R.bw;
						pj = p[j];
					}
					if (valid && pj >= 0) t[pj] = i; // marks only reach candidates with a smaller y, i.e. visited later
					__syncthreads();
					const bool hit_t = valid && t[j] == i;
					const int32_t pm = lc_scan_max(valid ? sc2 : INT32_MIN, INT32_MIN);
					int32_t exm = lc_prev_lane(pm, INT32_MIN);
					if (exm < max_f) exm = max_f;
					const bool improve = valid && sc2 > exm;
					const uint64_t m_imp = __ballot(improve);
					int32_t n_skip = 0;
					const int cut_lane = lc_skip_replay(improve, hit_t && !improve, R.max_skip, &n_skip);
					const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
					const uint64_t imp_b = m_imp & before;
					if (imp_b) {
						const int bl = 63 - __clzll(imp_b);
						max_f = __shfl(sc2, bl), max_j = __shfl(j, bl);
					}
				}
				__syncthreads();
			}
		}
		int32_t vi = max_f;
		if (max_j >= 0) { const int32_t vj = v[max_j]; if (vj > max_f) vi = vj; }
		if (lane == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; }
		__syncthreads();
	}
	return true;
}

// ---------------- backtrack (lchain.c:27-77) + compact_a (lchain.c:79-112): chains of a[] -> (u, b) ----------------
// In two parts with the klib sort of the chain ends between them, so that a read's two backtracks (first pass, long-join rescue) and its sorts each have ONE call site:
// [measured, round 6] inlined per call the kernel was 59 KB of code and every phase ran 2-5 x slower -- the waves of a CU are spread over all phases and the
// instruction cache did not hold them.
// part 1: z = chain ends with f >= min_sc, in index order; returns their number
__device__ __forceinline__ int32_t lc_chain_ends(int32_t n, int32_t min_sc, lc_ws_t W, int lane)
{
	const int32_t *f = W.f;
	mg128_t *z = W.z;
	int32_t n_z = 0;
	for (int32_t c0 = 0; c0 < n; c0 += 64) {
		const int32_t i = c0 + lane;
		const bool ok = i < n && f[i] >= min_sc;
		const uint64_t m = __ballot(ok);
		if (ok) { mg128_t e; e.x = (uint64_t)(int64_t)f[i]; e.y = (uint64_t)i; z[n_z + __popcll(m & mga_lanemask_lt())] = e; }
		n_z += __popcll(m);
	}
	__syncthreads();
	return n_z;
}

// part 2, behind the klib sort of z by score: the walks from the best end down, then the compaction
__device__ __forceinline__ void lc_walk_compact(const mg128_t *a, int32_t n, int32_t n_z, int32_t min_sc, int32_t min_cnt, int32_t max_drop, lc_ws_t W,
												uint64_t *u, mg128_t *b, /* a may alias b: a is fully read into z before b is written */ int32_t *n_u_, int32_t *n_v_, lc_lds_t *L, int lane, long long &tick_)
{
	lc_lds_t *LP = L;
	int32_t *f = W.f, *p = W.p, *v = W.v, *t = W.t;
	mg128_t *z = W.z;
	*n_u_ = 0, *n_v_ = 0;
	if (n_z == 0) return;
	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	__syncthreads();
	int32_t n_u = 0, n_v = 0;
	// The walk (lchain.c:9-25, 41-60).  Rounds 1-5 gave it to one lane: a chain of dependent loads -- z[k] -> t[e] per chain end, p[i] -> f[i], t[i] per step, three times over
	// the same path (mg_chain_bk_end, the reset of t[], the collection) -- [measured, round 6, profiles/r06f_lchain_phases.txt] 1.6 M cycles per backtrack for 250 chain ends
	// and 256 steps, a third of the kernel.  Now the wavefront keeps a BLOCK of 64 anchors' p / f / t in registers (lane l <-> anchor top - l): a step reads its successor
	// from a lane (p[i] < i: the path only descends, so the block slides down and is reloaded when the path leaves it -- one trip per 64 anchors instead of three per step);
	// the path goes to v[] as it is walked (the collection IS the path's head up to the best cut), the marks t[] = 1 of the kept part are set afterwards by all lanes.
	// t[] = 2 is never materialised: a path cannot meet itself, and every anchor the reference marks 2 is 0 or 1 again when its walk ends.
	// Chain ends are taken 64 at a time: the ones already marked are skipped by a ballot, the marks re-read after every walk that set any.
	unsigned long long c_walk = 0, c_step = 0;
	for (int32_t k0 = n_z - 1; k0 >= 0; k0 -= 64) {
		const int32_t k = k0 - lane;
		int32_t ze = 0, zsc = 0, te = 1;
		if (k >= 0) { const mg128_t zk = z[k]; ze = (int32_t)zk.y, zsc = (int32_t)zk.x; te = t[ze]; }
		uint64_t m_free = __ballot(te == 0);
		LC_TICK(13);
		while (m_free) {
			const int lw = (int)__builtin_ctzll(m_free);
			const int32_t e = __builtin_amdgcn_readlane(ze, lw), zs = __builtin_amdgcn_readlane(zsc, lw);
			++c_walk;
			int32_t top = e, pb = -1, fb = 0, tb = 0;
			{ const int32_t j = top - lane; if (j >= 0) pb = p[j], fb = f[j], tb = t[j]; }
			int32_t i = e, cnt = 0, best = 0, best_cnt = 0;
			for (;;) {
				if (lane == 0) v[n_v + cnt] = i;
				++cnt;
				const int32_t ni = __builtin_amdgcn_readlane(pb, top - i);
				if (ni < 0) { if (zs > best) best = zs, best_cnt = cnt; break; } // (a drop at the chain's start ends the loop like its end does)
				if (top - ni > 63) { top = ni; const int32_t j = top - lane; pb = -1, fb = 0, tb = 0; if (j >= 0) pb = p[j], fb = f[j], tb = t[j]; }
				const int32_t sc = zs - __builtin_amdgcn_readlane(fb, top - ni);
				if (sc > best) best = sc, best_cnt = cnt;
				else if (best - sc > max_drop) break;
				i = ni;
				if (__builtin_amdgcn_readlane(tb, top - i) != 0) break;
			}
			c_step += cnt;
			LC_TICK(14);
			const uint64_t m_rest = lw == 63 ? 0 : m_free & (~0ULL << (lw + 1));
			if (best_cnt > 0) {
				__syncthreads(); // the path is in v[]
				for (int32_t q = lane; q < best_cnt; q += 64) t[v[n_v + q]] = 1;
				if (best >= min_sc && best_cnt >= min_cnt) { if (lane == 0) u[n_u] = (uint64_t)best << 32 | (uint64_t)best_cnt; ++n_u; n_v += best_cnt; }
				if (m_rest) {
					__syncthreads(); // the marks are set
					if (k >= 0) te = t[ze];
					m_free = __ballot(te == 0) & m_rest;
				} else m_free = 0;
			} else m_free = m_rest;
			LC_TICK(15);
		}
		__syncthreads();
	}
	LC_COUNT(8, n); LC_COUNT(9, n_z); LC_COUNT(10, c_walk); LC_COUNT(11, c_step); LC_COUNT(12, 1);
	__syncthreads();
	LC_TICK(7);
	if (n_u == 0) return;
	// NB: v[] was overwritten from index 0 by the walk (n_v <= anchors visited), as in the reference
	{
		int32_t k0 = 0;
		for (int32_t c = 0; c < n_u; ++c) {
			const int32_t ni = (int32_t)u[c];
			for (int32_t jj = lane; jj < ni; jj += 64) z[k0 + jj] = a[v[k0 + (ni - jj - 1)]];
			k0 += ni;
		}
	}
	__syncthreads();
	// w[c] = (first anchor x, start<<32|c), klib-sorted by x; stored in the f/p area as mg128 (n_u <= n/min_cnt, 16n bytes available)
	mg128_t *w = (mg128_t*)f;
	if (lane == 0) {
		int32_t k0 = 0;
		for (int32_t c = 0; c < n_u; ++c) { w[c].x = z[k0].x; w[c].y = (uint64_t)k0 << 32 | (uint64_t)c; k0 += (int32_t)u[c]; }
	}
	__syncthreads();
	if (n_u <= 64) klib_rank_sort64(w, n_u); else klib_sort128x(w, n_u, t, &L->big); // (t[] is free by now and large enough for the range stack)
	uint64_t *u2 = (uint64_t*)v; // n_u * 8 bytes <= n * 4 bytes when min_cnt >= 2; guarded by the host wrapper
	if (lane == 0) for (int32_t c = 0; c < n_u; ++c) u2[c] = u[(int32_t)w[c].y];
	__syncthreads();
	{
		int32_t k0 = 0;
		for (int32_t c = 0; c < n_u; ++c) {
			const int32_t cnt = (int32_t)u2[c], src = (int32_t)(w[c].y >> 32);
			for (int32_t jj = lane; jj < cnt; jj += 64) b[k0 + jj] = z[src + jj];
			k0 += cnt;
		}
	}
	__syncthreads();
	for (int32_t c = lane; c < n_u; c += 64) u[c] = u2[c];
	__syncthreads();
	*n_u_ = n_u, *n_v_ = n_v;
}

// ---------------- first-pass DP for TWO reads per wavefront (round 5) ----------------
// lc_dp() gives a read all 64 lanes, but the predecessor scan of an anchor visits 26 predecessors on average before the skip heuristic cuts it ([measured, hchain.c's counters on
// 600 bench-like reads] 11 % of the anchors visit <= 16, 87 % 17-32, none more) and the kernel is bound by the instructions a wavefront issues per anchor, vector and scalar
// alike (SQ counters, round 4: 111 k vector + 95 k scalar instructions per read, 72 % of a wave's cycles waiting): half of every vector instruction's lanes and all of the scalar
// stream serve one read.  Here lanes 0-31 take one read and lanes 32-63 another, SIMT style: the same code, every "scalar" a per-lane value that is uniform inside its group of
// 32, blocks of 32 predecessors, scans / ballots / shuffles confined to the group (rows of 16 + one row broadcast; masks of 32 bits).  The replay of the sequential heuristics
// is the one of lc_dp() -- it carries max_f, the skip counter and the cut from block to block, so the block size cannot change a result.
#define LC_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#define LC_SCAN32(name, OP) \
	__device__ __forceinline__ int32_t name(int32_t v, const int32_t ident) \
	{ \
		int32_t t; \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false); v = OP(v, t); /* row_bcast:15 into rows 1 and 3: the upper row of either half */ \
		return v; \
	}
LC_SCAN32(lc_scan32_add, LC_OP_ADD)
LC_SCAN32(lc_scan32_min, LC_OP_MIN)
LC_SCAN32(lc_scan32_max, LC_OP_MAX)
__device__ __forceinline__ uint32_t lcg_ballot(bool x, int grp) { return (uint32_t)(__ballot(x) >> (grp * 32)); } // the group's 32 bits (lanes of the other group that are off contribute nothing anyway)
__device__ __forceinline__ int32_t lcg_prev_lane(int32_t v, int32_t first, int32_t m_first) // lane l <- v[l - 1] inside the group; its first lane <- first (select by mask: a lane that branches around a DPP move is not read by it)
{
	const int32_t t = __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
	return (m_first & first) | (~m_first & t);
}
__device__ __forceinline__ int lcg_skip_replay(bool improve, bool hit, int32_t max_skip, int32_t *n_skip, int grp) // lc_skip_replay() over a group's block of 32; returns the cut's lane in the group (32: none)
{
	const int32_t d = improve ? -1 : hit ? 1 : 0;
	const int32_t S = lc_scan32_add(d, 0) + *n_skip;
	const int32_t M = lc_scan32_min(S, 0x7fffffff);
	const int32_t nk = S - (M < 0 ? M : 0);
	const uint32_t m_cut = lcg_ballot(hit && !improve && nk > max_skip, grp);
	if (m_cut) return (int)__builtin_ctz(m_cut);
	*n_skip = __shfl(nk, grp * 32 + 31);
	return 32;
}

__device__ __forceinline__ void lc_dp2(const mg128_t *__restrict__ a, int32_t n, mga_lchain_par_t P, lc_ws_t W, int lane)
{
	int32_t *f = W.f, *p = W.p, *v = W.v, *t = W.t;
	const int grp = lane >> 5, gl = lane & 31, g0 = grp * 32;
	const int32_t m_first = gl == 0 ? -1 : 0;
	for (int32_t i = gl; i < n; i += 32) t[i] = 0;
	LC_FENCE();
	int32_t st = 0, max_ii = -1;
	for (int32_t i = 0; i < n; ++i) {
		{ // a leading run of anchors without any predecessor in reach (see lc_dp)
			const int32_t k = i + gl;
			bool iso = false;
			uint64_t yk = 0;
			if (k < n) {
				const uint64_t xk = a[k].x;
				yk = a[k].y;
				if (k == 0) iso = true;
				else { const uint64_t xp = a[k - 1].x; iso = xk >> 32 != xp >> 32 || xk > xp + (uint64_t)(int64_t)P.max_dist_x; }
			}
			const uint32_t m = lcg_ballot(iso, grp);
			const int r = (~m) ? (int)__builtin_ctz(~m) : 32;
			if (r > 0) {
				if (gl < r) { const int32_t sp = (int32_t)(yk >> 32 & 0xff); f[k] = sp, p[k] = -1, v[k] = sp; }
				max_ii = i + r - 1;
				if (st < i + r - 1) st = i + r - 1;
				i += r - 1;
				LC_FENCE();
				continue;
			}
		}
		const uint64_t xi = a[i].x, yi = a[i].y;
		uint64_t xs0 = st < i ? a[st].x : 0;
		mg128_t am; am.x = am.y = 0;
		int32_t fm = 0, vm = 0;
		if (max_ii >= 0) am = a[max_ii], fm = f[max_ii], vm = v[max_ii];
		while (st < i) { // lchain.c:171
			if (xi >> 32 != xs0 >> 32 || xi > xs0 + (uint64_t)(int64_t)P.max_dist_x) { ++st; if (st < i) xs0 = a[st].x; } else break;
		}
		if (i - st > P.max_iter) st = i - P.max_iter;
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1, max_v = 0, n_skip = 0, end_j = st - 1;
		bool cut = false;
		for (int32_t j0 = i - 1; j0 >= st && !cut; j0 -= 32) {
			const int32_t j = j0 - gl;
			const bool act = j >= st;
			int32_t sc = LC_NONE, pj = -1, vj = 0;
			if (act) {
				const mg128_t aj = a[j];
				const int32_t fj = f[j];
				pj = p[j], vj = v[j];
				sc = lc_score(xi, yi, aj.x, aj.y, P);
				if (sc != LC_NONE) sc += fj; else pj = -1;
			}
			const bool valid = sc != LC_NONE;
			if (valid && pj >= 0) t[pj] = i; // lchain.c:188 (harmless beyond the cut: only compared against this i)
			LC_FENCE();
			const bool hit_t = valid && t[j] == i;
			const int32_t pm = lc_scan32_max(valid ? sc : INT32_MIN, INT32_MIN);
			int32_t ex = lcg_prev_lane(pm, INT32_MIN, m_first);
			if (ex < max_f) ex = max_f;
			const bool improve = valid && sc > ex;
			const uint32_t m_imp = lcg_ballot(improve, grp);
			const int cut_lane = lcg_skip_replay(improve, hit_t && !improve, P.max_skip, &n_skip, grp);
			const uint32_t before = cut_lane == 32 ? ~0u : (1u << cut_lane) - 1u;
			const uint32_t imp_b = m_imp & before;
			if (imp_b) {
				const int bl = 31 - (int)__builtin_clz(imp_b);
				max_f = __shfl(sc, g0 + bl), max_v = __shfl(vj, g0 + bl), max_j = j0 - bl;
			}
			if (cut_lane < 32) { cut = true; end_j = j0 - cut_lane; }
			LC_FENCE();
		}
		// lchain.c:191-196: best-scoring anchor within reach, recomputed when it fell out of range
		if (max_ii < 0 || xi - am.x > (uint64_t)(int64_t)P.max_dist_x) {
			int32_t bf = INT32_MIN, bj = -1;
			for (int32_t j = i - 1 - gl; j >= st; j -= 32) { const int32_t fj = f[j]; if (bf < fj) bf = fj, bj = j; } // descending j per lane: first max kept
			for (int d = 16; d > 0; d >>= 1) {
				const int32_t of = __shfl_xor(bf, d), oj = __shfl_xor(bj, d);
				if (of > bf || (of == bf && oj > bj)) bf = of, bj = oj; // ties: the larger j was met first
			}
			max_ii = bj;
			if (max_ii >= 0) am = a[max_ii], fm = f[max_ii], vm = v[max_ii];
		}
		if (max_ii >= 0 && max_ii < end_j) { // lchain.c:197-201
			const int32_t tmp = lc_score(xi, yi, am.x, am.y, P);
			if (tmp != LC_NONE && max_f < tmp + fm) max_f = tmp + fm, max_j = max_ii, max_v = vm;
		}
		int32_t vi = max_f;
		if (max_j >= 0 && max_v > max_f) vi = max_v;
		if (gl == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; }
		if (max_ii < 0 || (xi - am.x <= (uint64_t)(int64_t)P.max_dist_x && fm < max_f)) max_ii = i;
		LC_FENCE();
	}
}

// per read: where its anchors and workspace lie, its parameters (map-algo.c:383-386: the reference gap depends on the read's length when -F is given)
struct lc_read_t { const mg128_t *a; int32_t n; lc_ws_t W; uint64_t *u; mg128_t *b; int64_t off; mga_lchain_par_t P; };
__device__ __forceinline__ void lc_read_setup(int r, const mg128_t *a_all, const int64_t *a_off, mga_lchain_par_t P, const lc_rescue_t &R, const int64_t *q_off,
											   uint64_t *u_all, mg128_t *b_all, int32_t *ws_i32, mg128_t *ws_z, lc_read_t *o)
{
	const int64_t off = a_off[r];
	const int32_t n = (int32_t)(a_off[r + 1] - off);
	o->off = off, o->n = n, o->a = a_all + off;
	o->W.f = ws_i32 + off * 4, o->W.p = o->W.f + n, o->W.v = o->W.p + n, o->W.t = o->W.v + n; // 4 int32 per anchor
	o->W.z = ws_z + off;                                                                   // 1 mg128 per anchor
	o->u = u_all + off, o->b = b_all + off;
	if (R.frag_len > 0 && q_off) {
		const int32_t g = R.frag_len - (int32_t)(q_off[r + 1] - q_off[r]);
		P.max_dist_x = g > R.frag_min_gap ? g : R.frag_min_gap;
	}
	if (P.max_dist_x < P.bw) P.max_dist_x = P.bw;
	if (P.max_dist_y < P.bw) P.max_dist_y = P.bw;
	o->P = P;
}

// everything behind the first-pass DP of read r: backtrack + compaction, the long-join rescue (map-algo.c:407-417), the read's counts and flag.  The whole wavefront, one read.
__device__ __forceinline__ void lc_read_finish(int r, const lc_read_t &X, const lc_rescue_t &R, const int64_t *__restrict__ q_off, int32_t *__restrict__ d_nu, int32_t *__restrict__ d_nb,
							   int32_t *__restrict__ d_flag, mg128_t *__restrict__ ws_keep, lc_lds_t *L, int lane, long long &tick_)
{
	lc_lds_t *LP = L;
	const mg128_t *a = X.a;
	const int32_t n = X.n;
	const lc_ws_t W = X.W;
	uint64_t *u = X.u;
	mg128_t *b = X.b;
	const int64_t off = X.off;
	const mga_lchain_par_t &P = X.P;
	int32_t n_u = 0, n_v = 0;
	mg128_t *keep = (mg128_t*)((char*)ws_keep + off * 24); // the first-pass result (16 B/anchor + 8 B/chain), should the host have to take over; scratch of the sorts while it holds nothing
	uint64_t *keep_u = 0;
	const mg128_t *src = a;
	int32_t n_src = n, min_sc = P.min_sc, min_cnt = P.min_cnt, max_drop = P.bw;
	// step 0: the first pass's backtrack; 1: the long-join rescue's sort + DP (map-algo.c:407-417); 2: the rescue's backtrack.  One sort, one walk, whatever the step.
	for (int step = 0;;) {
		mg128_t *sa, *stmp;
		int32_t sn;
		if (step == 1) sa = b, sn = n_v, stmp = W.z; // all chained anchors, by x (n_v = sum of the chain sizes); z is free between the backtracks
		else { sa = W.z, sn = __builtin_amdgcn_readfirstlane(lc_chain_ends(n_src, min_sc, W, lane)), stmp = keep; LC_TICK(5); }
		lc_sort(sa, sn, stmp, W.t, L); // (t[] is free here -- re-zeroed by the walk -- and large enough for the range stack)
		if (step == 1) {
			__syncthreads();
			LC_TICK(2);
			const bool rq_ok = n_v <= LC_RQ_LDS ? lc_dp_rmq_lds(b, n_v, R, W, L, lane) : lc_dp_rmq(b, n_v, R, W, L, lane);
			if (g_lc_prof_on && lane == 0) { const int q_ = n_v <= LC_RQ_LDS ? 16 : 17; LP->prof[q_] += 1, LP->prof[q_ + 2] += (unsigned long long)n_v, LP->prof[q_ + 4] += (unsigned long long)((long long)clock64() - tick_); }
			LC_TICK(3);
			if (!rq_ok) {
				__syncthreads();
				for (int32_t i = lane; i < n_v; i += 64) b[i] = keep[i];
				for (int32_t i = lane; i < n_u; i += 64) u[i] = keep_u[i];
				if (lane == 0 && d_flag) d_flag[r] = 2;
				break;
			}
			// mg_lchain_rmq backtracks with max_drop = its bw (lchain.c:267,359); the anchors are read from b, the result overwrites b; keep is no longer needed: the rescue's chains stand
			step = 2, src = b, n_src = n_v, min_sc = R.min_sc, min_cnt = R.min_cnt, max_drop = R.bw;
			continue;
		}
		LC_TICK(6);
		lc_walk_compact(src, n_src, sn, min_sc, min_cnt, max_drop, W, u, b, &n_u, &n_v, L, lane, tick_);
		n_u = __builtin_amdgcn_readfirstlane(n_u), n_v = __builtin_amdgcn_readfirstlane(n_v);
		if (step == 2) { LC_TICK(4); if (lane == 0 && d_flag) d_flag[r] = 1; break; }
		LC_TICK(1);
		if (!(R.enabled && n_u > 1 && q_off)) break;
		const int32_t qlen = (int32_t)(q_off[r + 1] - q_off[r]);
		const int32_t st = (int32_t)b[0].y, en = (int32_t)b[(int32_t)u[0] - 1].y;
		const int32_t unc = qlen - (en - st);
		if (!(unc > R.rescue_size || (float)unc > (float)qlen * R.rescue_ratio)) break;
		if (n_v > LC_RESCUE_DEV_MAX) {
			// an ultra-long read (hundreds of kb and up): the rescue's per-anchor window searches on ONE wavefront would take longer than the host's
			// tree over the same anchors ([measured] round 1: 2 x 5 Mbp reads 2.2 s in here against 0.9 s for the whole reference job), so the read
			// goes the way of the tied ones -- first-pass chains out, flag 2, the host re-chains it with the RMQ tree (and chains it through the graph)
			if (lane == 0 && d_flag) d_flag[r] = 2;
			break;
		}
		keep_u = (uint64_t*)(keep + n_v);
		for (int32_t i = lane; i < n_v; i += 64) keep[i] = b[i];
		for (int32_t i = lane; i < n_u; i += 64) keep_u[i] = u[i];
		__syncthreads();
		step = 1;
	}
	if (lane == 0) { d_nu[r] = n_u; d_nb[r] = n_v; }
}

// d_flag[r]: 0 = chains of the first pass; 1 = long-join rescue applied on the device; 2 = rescue due, left to the host
template<int WPE> // resident waves per SIMD the register allocation is held to (measurement knob MGA_LC_WPE; see mga_dev_lchain)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) k_lchain(int n_reads, const mg128_t *__restrict__ a_all, const int64_t *__restrict__ a_off, mga_lchain_par_t P,
											   lc_rescue_t R, const int64_t *__restrict__ q_off,
											   uint64_t *__restrict__ u_all, mg128_t *__restrict__ b_all, int32_t *__restrict__ d_nu, int32_t *__restrict__ d_nb,
											   int32_t *__restrict__ d_flag, int32_t *__restrict__ ws_i32, mg128_t *__restrict__ ws_z, mg128_t *__restrict__ ws_keep,
											   const int32_t *__restrict__ order, int dp_win)
{
	__shared__ lc_lds_t L;
	lc_lds_t *LP = &L;
	if (threadIdx.x < 32) L.prof[threadIdx.x] = 0;
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_reads) return;
	const int r = order ? __builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int)blockIdx.x; // workgroups are dispatched in index order: the reads with the most anchors first (mapper.c)
	if (lane == 0 && d_flag) d_flag[r] = 0;
	lc_read_t X;
	lc_read_setup(r, a_all, a_off, P, R, q_off, u_all, b_all, ws_i32, ws_z, &X);
	if (X.n == 0) { if (lane == 0) d_nu[r] = 0, d_nb[r] = 0; return; }
	long long tick_ = g_lc_prof_on ? (long long)clock64() : 0;
	if (dp_win) lc_dp_w(X.a, X.n, X.P, X.W, &L, lane); else lc_dp(X.a, X.n, X.P, X.W, lane);
	LC_TICK(0);
	lc_read_finish(r, X, R, q_off, d_nu, d_nb, d_flag, ws_keep, &L, lane, tick_);
	if (g_lc_prof_on) { mga_wave_sync(); if (lane < 32 && L.prof[lane]) atomicAdd(&g_lc_prof[lane], L.prof[lane]); }
}

// the same, TWO reads per wavefront in the first-pass DP (lc_dp2), one after the other in everything behind it
__global__ void __launch_bounds__(64) k_lchain2(int n_reads, const mg128_t *__restrict__ a_all, const int64_t *__restrict__ a_off, mga_lchain_par_t P,
												lc_rescue_t R, const int64_t *__restrict__ q_off,
												uint64_t *__restrict__ u_all, mg128_t *__restrict__ b_all, int32_t *__restrict__ d_nu, int32_t *__restrict__ d_nb,
												int32_t *__restrict__ d_flag, int32_t *__restrict__ ws_i32, mg128_t *__restrict__ ws_z, mg128_t *__restrict__ ws_keep,
												const int32_t *__restrict__ order)
{
	__shared__ lc_lds_t L;
	lc_lds_t *LP = &L;
	if (threadIdx.x < 32) L.prof[threadIdx.x] = 0;
	const int lane = threadIdx.x, grp = lane >> 5;
	const int k0 = 2 * (int)blockIdx.x;
	if (k0 >= n_reads) return;
	const int ra = order ? __builtin_amdgcn_readfirstlane(order[k0]) : k0;
	const int rb = k0 + 1 < n_reads ? (order ? __builtin_amdgcn_readfirstlane(order[k0 + 1]) : k0 + 1) : -1; // (neighbours in the launch order: about the same number of anchors)
	if (lane == 0 && d_flag) { d_flag[ra] = 0; if (rb >= 0) d_flag[rb] = 0; }
	long long tick_ = g_lc_prof_on ? (long long)clock64() : 0;
	{ // the two first-pass DPs side by side: each half of the wavefront sets up and runs its own read
		const int my = grp == 0 ? ra : rb;
		if (my >= 0) {
			lc_read_t X;
			lc_read_setup(my, a_all, a_off, P, R, q_off, u_all, b_all, ws_i32, ws_z, &X);
			if (X.n > 0) lc_dp2(X.a, X.n, X.P, X.W, lane);
		}
	}
	__syncthreads(); // (both halves are through: what follows is the whole wavefront's, read by read)
	LC_TICK(0);
	for (int q = 0; q < 2; ++q) {
		const int r = q == 0 ? ra : rb;
		if (r < 0) break;
		lc_read_t X;
		lc_read_setup(r, a_all, a_off, P, R, q_off, u_all, b_all, ws_i32, ws_z, &X);
		if (X.n == 0) { if (lane == 0) d_nu[r] = 0, d_nb[r] = 0; continue; }
		lc_read_finish(r, X, R, q_off, d_nu, d_nb, d_flag, ws_keep, &L, lane, tick_);
		__syncthreads();
	}
	if (g_lc_prof_on) { mga_wave_sync(); if (lane < 32 && L.prof[lane]) atomicAdd(&g_lc_prof[lane], L.prof[lane]); }
}

// stage test: the sorts of this file on their own (tests/test_gpu_stages.py: test_device_klib_sort)
__global__ void __launch_bounds__(64) k_sort128x(int n, mg128_t *a_all, const int64_t *__restrict__ a_off, mg128_t *tmp_all, int32_t *stk_all)
{
	__shared__ lc_lds_t L;
	if ((int)blockIdx.x >= n) return;
	const int64_t off = a_off[blockIdx.x];
	const int64_t m = a_off[blockIdx.x + 1] - off;
	if (m > 0x7fffffff) return;
	lc_sort(a_all + off, (int32_t)m, tmp_all + off, stk_all + 3 * off + 8 * (int64_t)blockIdx.x, &L);
}
extern "C" int mga_dev_sort128x(mga_sctx_t *sc, int n, mg128_t *d_a, const int64_t *d_a_off, mg128_t *d_tmp, int32_t *d_stk)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_sort128x, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_a, d_a_off, d_tmp, d_stk);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" size_t mga_dev_lchain_ws_bytes(int64_t total_anchors) { return (size_t)(total_anchors + 16) * 56; }

extern "C" int mga_dev_lchain(mga_sctx_t *sc, int n, const mg128_t *d_a, const int64_t *d_a_off, const mga_lchain_par_t *par, const mga_rescue_par_t *resc,
							  const int64_t *d_q_off, uint64_t *d_u, mg128_t *d_b, int32_t *d_nu, int32_t *d_nb, int32_t *d_flag, void *d_ws, size_t ws_bytes, int64_t total_anchors)
{
	const int32_t *d_order = sc->lc_order; // the launch order is consumed by THIS call whatever its outcome (an early return used to leave it for the next call on the context)
	sc->lc_order = 0;
	if (n <= 0) return 0;
	if (par->min_cnt < 2) { mga_set_error("lchain: min_cnt >= 2 required by the workspace layout (got %d)", par->min_cnt); return -1; }
	if (ws_bytes < mga_dev_lchain_ws_bytes(total_anchors)) { mga_set_error("lchain: workspace too small"); return -1; }
	int32_t *ws_i32 = (int32_t*)d_ws;
	mg128_t *ws_z = (mg128_t*)((char*)d_ws + (size_t)(total_anchors + 8) * 16);
	mg128_t *ws_keep = (mg128_t*)((char*)d_ws + (size_t)(total_anchors + 8) * 32); // b copy (16 B/anchor) + u copy (<= 8 B/anchor / min_cnt)
	lc_rescue_t R;
	memset(&R, 0, sizeof R);
	if (resc && resc->enabled && d_q_off && d_flag) {
		if (resc->min_cnt < 2) { mga_set_error("lchain rescue: min_cnt >= 2 required"); return -1; }
		R.enabled = 1, R.max_dist = resc->max_dist, R.max_dist_inner = resc->max_dist_inner, R.bw = resc->bw, R.max_skip = resc->max_skip, R.cap = resc->cap;
		R.min_cnt = resc->min_cnt, R.min_sc = resc->min_sc, R.pen_gap = resc->chn_pen_gap, R.pen_skip = resc->chn_pen_skip;
		R.rescue_size = resc->rescue_size, R.rescue_ratio = resc->rescue_ratio;
	}
	if (resc && d_q_off) R.frag_len = resc->frag_len, R.frag_min_gap = resc->frag_min_gap;
	{
		static int prof_on = -1;
		if (prof_on < 0) { const char *e = getenv("MGA_LC_PROF"); prof_on = e && atoi(e) > 0; if (prof_on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lc_prof_on), &prof_on, sizeof(int)); }
		if (prof_on) { // print what the previous launches accumulated
			unsigned long long h[32];
			if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lc_prof), sizeof h) == hipSuccess)
				fprintf(stderr, "[lc-prof] Mcycles so far: dp %.1f backtrack %.1f rescue-sort %.1f rescue-dp %.1f rescue-backtrack %.1f | inside both backtracks: ends %.1f sort %.1f walk %.1f (their rest = compaction); %llu backtracks: anchors %llu ends %llu walks %llu steps %llu; walk = ends' marks %.1f + steps %.1f + marks %.1f; rescue DPs in LDS %llu (%llu anchors, %.1f Mcycles), in memory %llu (%llu anchors, %.1f Mcycles)\n", h[0] * 1e-6, h[1] * 1e-6, h[2] * 1e-6, h[3] * 1e-6, h[4] * 1e-6, h[5] * 1e-6, h[6] * 1e-6, h[7] * 1e-6, h[12], h[8], h[9], h[10], h[11], h[13] * 1e-6, h[14] * 1e-6, h[15] * 1e-6, h[16], h[18], h[20] * 1e-6, h[17], h[19], h[21] * 1e-6);
		}
	}
	mga_prof_begin(sc->stream, MGA_K_LCHAIN);
	{
		// [measured, round 5, bench workload, isolated pass / pipelined step, two interleaved repetitions] two reads per wavefront in the first-pass DP: 79.8 ms and 3.59-3.66
		// Gbp/s against 67.0 ms and 3.74-3.76 with one -- bit-identical (every chaining test and the e2e sweep pass in both forms), but a launch lasts as long as its longest
		// wavefront, and a wavefront that backtracks and rescues TWO reads one after the other behind a DP that lasts as long as the longer of the two is 1.5 x the longest read.
		// Kept as MGA_LC_PAIR=1 (the parity tests run it), not the default.
		const char *e_pair = getenv("MGA_LC_PAIR");
		// (also measured and not kept, `git log -p` has it: the first-pass DP over 16-byte {f, p, v, t} records in z[] -- ~9 vector-memory instructions per anchor step instead of
		// ~14, bit-identical, 66.5 vs 66.8 ms: the kernel is not bound by the memory instructions a CU takes either.  A launch is its longest read's chain of dependent trips.)
		const char *e_win = getenv("MGA_LC_WIN"); // 0: the first-pass DP over global memory (lc_dp), the form of rounds 1-5; default: the last 64 anchors in registers (lc_dp_w)
		const int dp_win = !(e_win && *e_win && atoi(e_win) == 0);
		const char *e_wpe = getenv("MGA_LC_WPE");
		const int wpe = e_wpe && *e_wpe ? atoi(e_wpe) : 6; // [measured, round 6, profiles/r06f_lchain_phases.txt] 6 waves (80 VGPRs, 8 spilled): 36.6 ms, 7 (72, 21 spilled): 37.0, 5: slower by 2
#define LC_LAUNCH(W_) hipLaunchKernelGGL(k_lchain<W_>, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_a, d_a_off, *par, R, d_q_off, d_u, d_b, d_nu, d_nb, d_flag, ws_i32, ws_z, ws_keep, d_order, dp_win)
		if (!(e_pair && atoi(e_pair) > 0)) { if (wpe <= 5) LC_LAUNCH(5); else if (wpe == 6) LC_LAUNCH(6); else LC_LAUNCH(7); }
		else hipLaunchKernelGGL(k_lchain2, dim3((n + 1) / 2), dim3(64), 0, (hipStream_t)sc->stream, n, d_a, d_a_off, *par, R, d_q_off, d_u, d_b, d_nu, d_nb, d_flag, ws_i32, ws_z, ws_keep, d_order);
	}
	mga_prof_end(sc->stream, MGA_K_LCHAIN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
