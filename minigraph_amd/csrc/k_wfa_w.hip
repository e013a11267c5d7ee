// k_wfa_w.hip -- WINDOWED exact 2-piece affine WFA (round 3): the gap filler's fast tiers for alignments scoring < 256.
//
// Why a window is exact.  miniwfa's exact mode (miniwfa.c:380-435) widens its band by one diagonal per score on both sides, so a
// problem of score S costs ~S^2 cells although its optimal paths stay near the line between the start diagonal 0 and the end
// diagonal e = ql - tl.  Moving n diagonals costs at least g(n) = min(o1 + n e1, o2 + n e2) = min(4 + 2n, 15 + n): a path that
// visits diagonal d scores at least g(|d|) + g(|e - d|).  Fix a window [lo, hi] of diagonals and let B be that bound for the
// first diagonals outside it (lo - 1, hi + 1; none if the window reaches the matrix edge -tl / ql).  If the alignment found
// INSIDE the window scores S < B, then no cell outside the window lies on ANY optimal path, and every cell that does lie on
// one has the same value, the same winning predecessor and the same tie-break bits as in the reference's wider band: a pruned
// predecessor can only lower a value (furthest-reaching values are monotone in their inputs, extension included), never raise
// it, and a predecessor that wins or ties at a cell of an optimal path is itself on an optimal path, hence inside the window
// and exact by induction over the score.  The traceback (miniwfa.c:329-377) only ever visits cells of optimal paths, so score
// and CIGAR are the reference's.  A problem that reaches score B undecided leaves the tier (MGA_WFA_RETRY_TIER) and is run
// again in a wider window; the windowed tiers stop at score 256, below the reference's first band trimming (miniwfa.c:139-169,
// :420), so its band is still the complete set of reachable diagonals there.  [measured, 359 519 gaps of the bench workload,
// CPU model of this file against the reference's own mwf_wfa_auto: 0 differences; 61.5 % of the gaps decide in 16 diagonals,
// 80 % in 32, 91.5 % in 64 -- the reference's band for the same gaps averages 2 S + 1 = 80.]
//
// What that buys on a 64-lane wavefront: FOUR problems per wave in 16-lane groups (tier W0), TWO in 32-lane groups (W1), one
// in 64 lanes (W2) and one in 2-4 slots of 64 lanes (W3-W5) -- no multi-wave tiers, no barrier, no band bookkeeping (the window
// is computed whole; cells the reference's band has not reached hold "unreachable" + small -- 0 + small under the bias of
// round 5, WFW_BIAS below -- which loses every comparison exactly like the NEG_INF of the reference's padding).
//
// Forward pass (k_wfa_fw): lane l of a group owns diagonal lo + l (+ 64 j for slot j).  H of the last 17 scores, E1/F1 (3),
// E2/F2 (2) live in VGPRs indexed by age as in k_wfa_r.hip; neighbours come from DPP row / wave shifts that stay inside the
// group.  Sequences are staged in LDS (four byte-shifted copies: any 8 bases are one aligned ds_read2_b32).  The groups of a wave
// run in lockstep but on independent problems: a group that finishes (or gives up) is retired and refilled from the work
// queue while the others carry on.  Traceback bytes never touch LDS: four scores are packed into one dword per lane and stored
// with one coalesced global_store_dword per four steps into the problem's own region of a scratch pool.
// Traceback (k_wfa_tb): one LANE per problem walks its region backwards (miniwfa.c:329-377) ONCE, writing the operators downwards
// from the end of a slot of the CIGAR pool sized by the score (one reservation per wavefront), and leaves the same mga_wfa_res_t the
// register tiers of k_wfa_r.hip leave.
#include <type_traits>
#include <stdlib.h>
#include "mga_dev.h"
#include "dev_common.h"
#include "wfa_window.h"

// (round 5) Cells are stored BIASED by WFW_BIAS: "unreachable" is 0 -- what a DPP shift with bound_ctrl feeds a group's edge lanes, what a cleared register holds and what `& okv`
// leaves of a diagonal outside the matrix -- and loses every comparison against a reached cell (>= WFW_BIAS - 1 = offset -1); an unreachable cell grows by at most one per step
// (< 300 in a window's life).  Before: NEG_INF = -2^30 had to be moved into the destination of every one of the eight neighbour shifts of a step.
#define WFW_BIAS 0x2000

extern "C" int32_t mga_wfw_window(int32_t W, int32_t tl, int32_t ql, int32_t *lo) { return wfw_window(W, tl, ql, lo, WFW_SMAX); } // (tests)

// furthest diagonal any path of score <= s can have reached: max n with wfw_gap(n) <= s
__device__ __forceinline__ int32_t wfw_reach(int32_t s) { return s < 6 ? 0 : max((s - 4) >> 1, s - 15); }

__device__ __forceinline__ int32_t wfw_max(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t wfw_sel(int32_t mask, int32_t a, int32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32: a where mask is all ones

// neighbour diagonals inside a group of G lanes.  from_left: lane l <- src[l-1]; the group's first lane gets "unreachable" (0), or with EDGE the previous slot's last lane
template<int G, bool EDGE> __device__ __forceinline__ int32_t wfw_from_left(int32_t edge, int32_t src, int32_t nm_first)
{
	if (G == 16) return __builtin_amdgcn_update_dpp(0, src, 0x111, 0xf, 0xf, true);       // row_shr:1, bound_ctrl: lane 0 of every row of 16 reads 0
	if (EDGE) return __builtin_amdgcn_update_dpp(edge, src, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 keeps `edge`
	const int32_t v = __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true);         // wave_shr:1, lane 0 reads 0
	return G == 32 ? v & nm_first : v;                                                     // (lane 32 must not see lane 31)
}
template<int G, bool EDGE> __device__ __forceinline__ int32_t wfw_from_right(int32_t edge, int32_t src, int32_t nm_last)
{
	if (G == 16) return __builtin_amdgcn_update_dpp(0, src, 0x101, 0xf, 0xf, true);       // row_shl:1
	if (EDGE) return __builtin_amdgcn_update_dpp(edge, src, 0x130, 0xf, 0xf, false);      // wave_shl:1
	const int32_t v = __builtin_amdgcn_update_dpp(0, src, 0x130, 0xf, 0xf, true);
	return G == 32 ? v & nm_last : v;
}

// per-lane problem state: 0 idle, 1 running, 2 reached the end cell (this lane holds it), 3 gives up (score bound / lengths)
enum { WFW_IDLE = 0, WFW_RUN = 1, WFW_DONE = 2, WFW_BAIL = 3 };

template<int G, int J, int SEQCAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(J == 1 ? 6 : 1, 8))) k_wfa_fw(const int *__restrict__ n_items_p, int cap, const int32_t *__restrict__ list, const mga_wfa_prob_t *__restrict__ prob,
											  const char *__restrict__ tseq, const char *__restrict__ qseq, mga_wfa_res_t *__restrict__ res,
											  char *__restrict__ tb, long long tb_stride, int *__restrict__ counter, mga_wfa_retry_t rt)
{
	static_assert(G == 16 || G == 32 || G == 64, "group size");
	static_assert(J == 1 || G == 64, "several slots per lane only with one problem per wave");
	constexpr int P = 64 / G;      // problems per wavefront
	constexpr int W = G * J;       // diagonals of the window = dwords per traceback row
	constexpr int SEQS = SEQCAP + 16;
	constexpr int QCH = P == 4 ? 32 : P == 2 ? 16 : J == 1 ? 4 : J == 2 ? 2 : 1; // items drawn from the queue at a time: many where problems are small and millions, one where each is
	                                                                           // hundreds of steps (a list is sorted longest first: eight in a row to ONE wavefront is a long tail)
	constexpr int MROWS = SEQCAP / 32 + 3; // words of a lane's match mask (+ guard rows that are read, never counted)
	// T: one copy.  Q: four copies, copy c shifted left by c bytes (position p of copy c, stored at c SEQS + 4 + p, is Q[p + c]), so that the four query bytes
	// opposite ANY target dword are one aligned dword read.
	__shared__ __attribute__((aligned(16))) uint8_t Tb[P][SEQS], Qb[P][4 * SEQS];
	// Match masks: bit b of Mk[w][64 j + lane] says T[32 w + b] == Q[d + 32 w + b] for the diagonal d of (slot j, lane) -- built once per problem (a lane keeps its
	// diagonal for the problem's life); extending a cell is then a count of trailing ones in a 32-bit window of the mask instead of a loop of byte compares
	// ([measured] the compare loop was ~60 of ~165 vector instructions of a step).  A row is 64 J dwords: lane l always hits bank l, whatever its word.
	__shared__ uint32_t Mk[MROWS][64 * J];
	const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
	const int n_items = min(*n_items_p, cap); // (the list's length is read here: rungs below this one appended to it during the same sweep)
	uint8_t *const Tg = Tb[grp], *const Qg = Qb[grp];
	const int32_t nm_first = gl == 0 ? 0 : -1, nm_last = gl == G - 1 ? 0 : -1; // (all ones except in a group's first / last lane)
	const uint64_t gmask = (G == 64 ? ~0ULL : ((1ULL << G) - 1ULL)) << (grp * G);
#define WFW_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

	int32_t st = WFW_IDLE, pi = -1, tl = 0, ql = 0, lo = 0, e = 0, s = 0, bnd = 0, lst = 0, ph = 0;
	int32_t okv[J];               // all ones where this lane's diagonal exists in the matrix (-tl <= d <= ql)
	int32_t fc[J];                // what the end cell looks like in this lane: tl - 1 on diagonal ql - tl, a value no offset takes elsewhere (and while the group is idle)
	uint32_t acc[J];              // traceback bytes of the last (up to) four steps
	uint32_t *tbp = 0;            // this lane's dword in the current traceback row (slot j: + 64 j)
	int32_t it_cur = 0;           // the problem's place in the work list = its traceback region
	int32_t H[J][18], E1[J][3], F1[J][3], E2[J][2], F2[J][2];
	int32_t t = 0;                // steps this wavefront has made (all groups advance together)
	int32_t q_next = 0, q_end = 0, q_base = 0; // the wavefront's reservoir of items; lane l holds the descriptor of item q_base + l
	__shared__ int32_t dsc_pi[QCH];
	__shared__ mga_wfa_prob_t dsc_pb[QCH]; // (descriptors of the reservoir's items, in LDS: seven VGPRs fewer than one per lane)
	bool q_empty = false, q_drained = false;
#pragma unroll
	for (int j = 0; j < J; ++j) {
		okv[j] = 0, acc[j] = 0, fc[j] = 0x7fffffff;
#pragma unroll
		for (int a = 0; a < 18; ++a) H[j][a] = 0;
#pragma unroll
		for (int a = 0; a < 3; ++a) E1[j][a] = F1[j][a] = 0;
#pragma unroll
		for (int a = 0; a < 2; ++a) E2[j][a] = F2[j][a] = 0;
	}

	// H of the last 16 scores is shifted by TWO registers every second step (k_wfa_r.hip): the step body exists twice, P = t & 1.
	// P = 0: age a is H[a + 2], the new slice goes to H[1]; P = 1: age a is H[a + 1], the new slice goes to H[0], then H[a] = H[a - 2].
	// The parity is the WAVE's, not the problem's: a problem that starts on an odd step puts its score-0 cell where age 0 is then.
#define HA(j_, a_) H[j_][(a_) + 2 - P]
	auto step = [&](auto Pc) __attribute__((always_inline)) -> bool { // false: the queue is empty and every group is idle
		constexpr int P = decltype(Pc)::value;
		// ---- retire the groups that are through, refill idle groups from the queue
		if (__ballot(st >= WFW_DONE || (st == WFW_IDLE && !q_empty))) {
			const uint64_t m_done = __ballot(st == WFW_DONE);
			const bool g_done = (m_done & gmask) != 0;
			if (g_done || st == WFW_BAIL) { // (a group gives up as a whole: s and bnd are the same in all of its lanes; one lane reaching the end cell in the same step wins)
				if (g_done) { // the rest of the last traceback row (bytes of steps t & ~3 .. t - 1)
					if (t & 3) {
						const int32_t rr = wfw_reach(s + 1), d0 = lo + gl;
#pragma unroll
						for (int j = 0; j < J; ++j) { const int32_t d = d0 + 64 * j; if ((d < 0 ? -d : d) <= rr) tbp[64 * j] = acc[j] << (8 * (4 - (t & 3))); }
					}
					if (st == WFW_DONE) {
						mga_wfa_res_t r;
						r.score = s, r.n_cigar = 0, r.cig_off = (int64_t)(uintptr_t)(tb + (long long)it_cur * tb_stride), r.status = MGA_WFA_TB, r.pad = lst | ph << 4 | W << 8, r.n_iter = 0;
						res[pi] = r;
					}
				} else if (gl == 0) {
					mga_wfa_res_t r;
					r.score = -1, r.n_cigar = 0, r.cig_off = 0, r.status = MGA_WFA_RETRY_TIER, r.pad = 0, r.n_iter = 0;
					res[pi] = r;
					mga_wfa_give_up(rt, pi); // next tier's work list
				}
				st = WFW_IDLE;
#pragma unroll
				for (int j = 0; j < J; ++j) fc[j] = 0x7fffffff; // (the retired problem's cells stay in the registers: they must not look like an end cell)
			}
			if (!q_empty) {
				const uint64_t m_idle = __ballot(st == WFW_IDLE && gl == 0);
				if (m_idle) {
					// The wavefront draws QCH items at a time from the launch's queue (ONE atomic: a counter bumped per problem by 7 000 resident waves was the
					// whole cost of the 16-lane tier, [measured] 13 ns per problem = the rate one address takes atomics at) and reads their descriptors at once,
					// one per lane; a refill then costs no dependent global round trip but the sequence bytes.
					if (q_next == q_end && !q_drained) {
						int32_t b = 0;
						if (lane == 0) b = atomicAdd(counter, QCH);
						b = __builtin_amdgcn_readfirstlane(b);
						q_base = q_next = b, q_end = b + QCH < n_items ? b + QCH : n_items;
						if (q_end <= q_next) q_end = q_next, q_drained = true;
						if (lane < q_end - q_next) {
							const int32_t p_ = list ? list[b + lane] : b + lane;
							dsc_pi[lane] = p_;
							dsc_pb[lane] = prob[p_];
						}
						WFW_LDS_FENCE();
					}
					const int32_t avail = q_end - q_next, n_idle = (int32_t)__popcll(m_idle);
					const int32_t rank = (int32_t)__popcll(m_idle & ((1ULL << (grp * G)) - 1ULL));
					const bool fill = st == WFW_IDLE && rank < avail;
					const int32_t item = q_next + rank, src = fill ? item - q_base : 0;
					q_next += n_idle < avail ? n_idle : avail;
					if (q_drained && q_next == q_end) q_empty = true;
					int32_t ntl = 0, nql = 0;
					const char *ts = tseq, *qs = qseq;
					if (fill) {
						const mga_wfa_prob_t pb = dsc_pb[src];
						pi = dsc_pi[src];
						ntl = pb.tl, nql = pb.ql, ts = tseq + pb.t_off, qs = qseq + pb.q_off;
						it_cur = item;
						tbp = (uint32_t*)(tb + (long long)item * tb_stride) + gl;
						ph = t & 3, s = 0, lst = 0, e = nql - ntl;
						if (ntl > SEQCAP || nql > SEQCAP) bnd = 0, ntl = nql = 0; // too long for this tier's LDS: gives up at once
						else { bnd = wfw_window(W, ntl, nql, &lo, WFW_SMAX); if (bnd > W + 30) bnd = W + 30; } // (W + 30 bounds every window's B; the traceback rows are sized for it)
						tl = ntl, ql = nql;
						st = WFW_RUN;
					}
					// stage the sequences of the refilled groups, a dword per lane and trip (the sequence buffers are padded: reading a few bytes past an end is fine,
					// and what lies beyond tl / ql never reaches a mask)
					WFW_LDS_FENCE(); // (the previous problem's reads of these LDS bytes are complete: same wave, in order)
					for (int32_t m0 = 0; __ballot(fill && 4 * m0 < ntl); m0 += G) {
						const int32_t m = m0 + gl;
						if (fill && 4 * m < ntl) { uint32_t v; __builtin_memcpy(&v, ts + 4 * m, 4); *(uint32_t*)(Tg + 4 * m) = v; }
					}
					for (int32_t m0 = 0; __ballot(fill && 4 * m0 < nql); m0 += G) {
						const int32_t m = m0 + gl;
						if (fill && 4 * m < nql) {
#pragma unroll
							for (int c = 0; c < 4; ++c) {
								uint32_t v;
								__builtin_memcpy(&v, qs + 4 * m + c, 4);
								*(uint32_t*)(Qg + c * SEQS + 4 + 4 * m) = v;
								// position -4..-1 of copy c: the c first bases at its top (a target dword opposite query positions -3..0 sees Q[0] there; what lies before is masked)
								if (m == 0 && c > 0) { uint32_t v0; __builtin_memcpy(&v0, qs, 4); *(uint32_t*)(Qg + c * SEQS) = v0 << (8 * (4 - c)); }
							}
						}
					}
					WFW_LDS_FENCE();
					// match masks of the refilled groups' diagonals, 32 target positions per word, four per compare block (SDWA byte compares, the carry shifts the bit in)
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t d = lo + gl + 64 * j;
						const int32_t kmin = d < 0 ? -d : 0, kmax = min(ntl, nql - d); // positions with 0 <= d + k < ql
						const uint8_t *qc = Qg + (d & 3) * SEQS + 4 + (d & ~3);         // query dword opposite target dword 0 (each copy has a 4-byte front pad; further out: masked below)
						for (int32_t w = 0; __ballot(fill && w <= (ntl >> 5)); ++w) {  // (one word beyond the last base: bit tl is a built, cleared bit -- a run stops there)
							uint32_t bits = 0;
#pragma unroll
							for (int q8 = 7; q8 >= 0; --q8) {
								const uint32_t t4 = *(const uint32_t*)(Tg + 32 * w + 4 * q8), q4 = *(const uint32_t*)(qc + 32 * w + 4 * q8);
								asm volatile("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
											 : "+v"(bits) : "v"(t4), "v"(q4) : "vcc");
							}
							const int32_t b_lo = kmin - 32 * w, b_hi = kmax - 32 * w; // valid bits of this word: [b_lo, b_hi)
							const uint32_t m_hi = b_hi >= 32 ? ~0u : b_hi <= 0 ? 0u : (1u << b_hi) - 1u, m_lo = b_lo <= 0 ? ~0u : b_lo >= 32 ? 0u : ~0u << b_lo;
							if (fill && w <= (ntl >> 5)) Mk[w][64 * j + lane] = bits & m_hi & m_lo;
						}
					}
					WFW_LDS_FENCE();
					const int32_t nmf = fill ? 0 : -1;
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t d = lo + gl + 64 * j;
#pragma unroll
						for (int a = 0; a < 18; ++a) H[j][a] &= nmf;
#pragma unroll
						for (int a = 0; a < 3; ++a) E1[j][a] &= nmf, F1[j][a] &= nmf;
#pragma unroll
						for (int a = 0; a < 2; ++a) E2[j][a] &= nmf, F2[j][a] &= nmf;
						if (fill) {
							acc[j] = 0;
							okv[j] = (d >= -tl && d <= ql) ? -1 : 0;
							fc[j] = d == e ? tl - 1 + WFW_BIAS : 0x7fffffff;
							if (d == 0) HA(j, 0) = WFW_BIAS - 1; // score 0: H[d = 0] = -1 (miniwfa.c:103-119)
						}
					}
				}
			}
		}
		if (!__ballot(st == WFW_RUN)) return false;
		// ---- extension of slice s (miniwfa.c:399-411); the end cell lies on the unique diagonal ql - tl
		const int32_t rs = (J > 1) ? wfw_reach(s) : 0; // (J > 1: one problem per wave, s is uniform)
#pragma unroll
		for (int j = 0; j < J; ++j) {
			if (J > 1) { const int32_t b0 = __builtin_amdgcn_readfirstlane(lo) + 64 * j; if (b0 > rs || b0 + 63 < -rs) continue; } // no diagonal of the slot is reachable yet
			const int32_t k0 = HA(j, 0), tp = k0 + (1 - WFW_BIAS);
			const bool val = st == WFW_RUN && (uint32_t)tp <= (uint32_t)tl; // -1 <= offset < tl (a cell before the query's start or beyond its end has no set bit to count)
			const int32_t wi = val ? tp >> 5 : 0;
			const uint32_t *mp = &Mk[wi][64 * j + lane];
			uint32_t inv = ~__builtin_amdgcn_alignbit(mp[64 * J], mp[0], tp); // ones of the mask from position tp on, as zeros (the shift is taken mod 32)
			inv = val ? inv : 1u; // (round 5) a cell that cannot be extended sees "no match at once": its run is 0 without a select, and it never asks for another window
			int32_t n = inv ? (int32_t)__builtin_ctz(inv) : 32;
			for (int32_t wj = wi + 1; __ballot(inv == 0u); ++wj) { // a run of 32 or more matches (3 % of the steps of a 10 %-error read): next window
				const uint32_t *mq = &Mk[wj < MROWS - 2 ? wj : MROWS - 2][64 * j + lane];
				const uint32_t v = ~__builtin_amdgcn_alignbit(mq[64 * J], mq[0], tp);
				n += inv == 0u ? (v ? (int32_t)__builtin_ctz(v) : 32) : 0;
				inv = inv == 0u ? v : inv;
			}
			const int32_t k = k0 + n;
			HA(j, 0) = k;
			if (k == fc[j]) { // the end cell: offset tl - 1 on diagonal e (then d + k == ql - 1; k0 + 1 <= tl holds there, and fc is out of reach in every other lane)
				st = WFW_DONE;
				lst = n == 0 ? (int32_t)(acc[j] & 7u) : 0; // it was entered by a gap state and not extended: the traceback starts in that state (miniwfa.c:406-407)
			}
		}
		if (st == WFW_RUN && s + 1 >= bnd) st = WFW_BAIL; // (uniform inside a group; the lane that just reached the end cell keeps WFW_DONE, and that wins)
		// ---- slice s + 1 (miniwfa.c:281-308).  Computed for every group; a group that is through ignores it.
		const int32_t rn = (J > 1) ? wfw_reach(s + 1) : 0;
		int32_t nH[J], nE1[J], nF1[J], nE2[J], nF2[J];
#pragma unroll
		for (int j = 0; j < J; ++j) {
			if (J > 1) {
				const int32_t b0 = __builtin_amdgcn_readfirstlane(lo) + 64 * j;
				if (b0 > rn || b0 + 63 < -rn) { nH[j] = nE1[j] = nF1[j] = nE2[j] = nF2[j] = 0; continue; }
			}
			// predecessors: score s+1-p is age p-1 now (ages are shifted at the end of the step)
#define WFW_L(R, a) (j > 0 ? wfw_from_left<G, true>(__builtin_amdgcn_readlane(R[j > 0 ? j - 1 : 0][a], 63), R[j][a], nm_first) : wfw_from_left<G, false>(0, R[j][a], nm_first))
#define WFW_R(R, a) (j < J - 1 ? wfw_from_right<G, true>(__builtin_amdgcn_readlane(R[j < J - 1 ? j + 1 : j][a], 0), R[j][a], nm_last) : wfw_from_right<G, false>(0, R[j][a], nm_last))
			const int32_t ho1l = WFW_L(H, 5 + 2 - P), e1l = WFW_L(E1, 1), ho2l = WFW_L(H, 15 + 2 - P), e2l = WFW_L(E2, 0);
			const int32_t ho1r = WFW_R(H, 5 + 2 - P), f1r = WFW_R(F1, 1), ho2r = WFW_R(H, 15 + 2 - P), f2r = WFW_R(F2, 0);
#undef WFW_L
#undef WFW_R
			const int32_t hx1 = HA(j, 3) + 1;
			const int32_t vE1 = wfw_max(ho1l, e1l), vE2 = wfw_max(ho2l, e2l);
			const int32_t vF1 = wfw_max(ho1r, f1r) + 1, vF2 = wfw_max(ho2r, f2r) + 1;
			const uint32_t bits = (ho1l < e1l ? 0x08u : 0u) | (ho2l < e2l ? 0x20u : 0u) | (ho1r < f1r ? 0x10u : 0u) | (ho2r < f2r ? 0x40u : 0u);
			const int32_t ee = wfw_max(vE1, vE2), ff = wfw_max(vF1, vF2), hh = wfw_max(ee, ff);
			const uint32_t ze = vE1 >= vE2 ? 1u : 3u, zf = vF1 >= vF2 ? 2u : 4u;
			uint32_t z = ee >= ff ? ze : zf;
			z = hx1 >= hh ? 0u : z;
			const int32_t vH = wfw_max(hx1, hh);
			acc[j] = acc[j] << 8 | (bits | z);
			nH[j] = vH & okv[j], nE1[j] = vE1 & okv[j], nF1[j] = vF1 & okv[j], nE2[j] = vE2 & okv[j], nF2[j] = vF2 & okv[j]; // (a diagonal outside the matrix stays unreachable)
		}
#pragma unroll
		for (int j = 0; j < J; ++j) { // age shift (H: every second step, by two)
			HA(j, -1) = nH[j];
			if (P == 1) {
#pragma unroll
				for (int a = 17; a > 1; --a) H[j][a] = H[j][a - 2];
			}
			E1[j][2] = E1[j][1]; E1[j][1] = E1[j][0]; E1[j][0] = nE1[j];
			F1[j][2] = F1[j][1]; F1[j][1] = F1[j][0]; F1[j][0] = nF1[j];
			E2[j][1] = E2[j][0]; E2[j][0] = nE2[j];
			F2[j][1] = F2[j][0]; F2[j][0] = nF2[j];
		}
		if ((t & 3) == 3) { // a row of traceback dwords is full: one coalesced store, next row
			if (st != WFW_IDLE) { // (lanes of a group that gives up in the very step another lane of it reaches the end cell must store too)
				// round 5: only the diagonals a path of the row's newest score (s + 1) can have reached are stored -- the cells beyond hold NEG_INF, lie on no path, and the
				// walk never reads them (it fails loudly if it ever did: uninitialised bytes cannot pass for a path for long).  [measured, round 4] the rungs wrote 48 GB of
				// traceback per 125 000 reads, a byte per cell of the whole window; the reachable part is about half of it in the rungs of 64 diagonals and more.
				const int32_t rr = wfw_reach(s + 1), d0 = lo + gl;
#pragma unroll
				for (int j = 0; j < J; ++j) { const int32_t d = d0 + 64 * j; if ((d < 0 ? -d : d) <= rr) tbp[64 * j] = acc[j]; }
			}
			tbp += W;
		}
		if (st == WFW_RUN) ++s;
		++t;
		return true;
	};
	for (;;) {
		if (!step(std::integral_constant<int, 0>())) break;
		if (!step(std::integral_constant<int, 1>())) break;
	}
#undef HA
}

// ---- PACKED forward pass for the rungs of 128 diagonals and more (round 5) ---------------------------------------------------------------
// k_wfa_fw gives every diagonal of the window a lane and a 32-bit register per slice; the rungs of 128 / 192 / 256 diagonals therefore run 2 / 3 / 4 "slots" of 64 diagonals,
// each with its own copy of the ~110 vector instructions of a step, and they are bound by vector-instruction ISSUE (the microbenchmark's rate is reached or passed: DESIGN 4).
// Offsets inside a window stay below 512 + 1 and the "unreachable" value only has to lose every comparison, so a furthest-reaching offset fits 16 bits: here a lane holds TWO
// NEIGHBOURING diagonals (2 l, 2 l + 1 of a set of 128) in the halves of one register and the recurrence -- the five maxima, the + 1s, the tie-break bits as sign masks of
// packed differences, the age shifts, the validity masks -- runs on v_pk_* instructions: once per 128 diagonals.  What stays per diagonal is the extension along the match mask
// (two bit-field extracts, two mask windows, one repack).  Neighbour exchange: diagonal 2 l - 1 is the high half of lane l - 1, 2 l + 2 the low half of lane l + 1 -- a DPP wave
// shift and a v_alignbit per operand.  One problem per wavefront, its scalars in SGPRs; traceback rows, result record and the walk (k_wfa_tb) are those of k_wfa_fw.
typedef short wfp_pk2 __attribute__((ext_vector_type(2)));
// Cells are stored BIASED by 0x2000 per half: "unreachable" is 0 -- what a DPP shift with bound_ctrl feeds the lanes at the wave's edge, what a fresh register holds, and what
// `& okv` leaves of a diagonal outside the matrix -- and loses every comparison against a reached cell (>= 0x1fff = offset -1).  An unreachable cell grows by at most one per
// step (< 300), a reached one stays below 0x2000 + 513: no half ever carries into its neighbour and every difference of two cells fits 16 bits.
#define WFP_BIAS 0x2000
#define WFP_ONE 0x00010001u
__device__ __forceinline__ uint32_t wfp_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(wfp_pk2, a), __builtin_bit_cast(wfp_pk2, b))); }
__device__ __forceinline__ uint32_t wfp_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (wfp_pk2)(__builtin_bit_cast(wfp_pk2, a) + __builtin_bit_cast(wfp_pk2, b))); }
__device__ __forceinline__ uint32_t wfp_lt(uint32_t a, uint32_t b) // all ones in every half where a < b
{
	const wfp_pk2 d = __builtin_bit_cast(wfp_pk2, a) - __builtin_bit_cast(wfp_pk2, b);
	return __builtin_bit_cast(uint32_t, (wfp_pk2)(d >> (wfp_pk2)(15)));
}
__device__ __forceinline__ uint32_t wfp_sel(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
// left neighbours of a lane's two diagonals: (high half of lane l - 1, own low half).  EDGE: lane 0's comes from `edge` (the previous set's lane 63); else it is unreachable (0)
template<bool EDGE> __device__ __forceinline__ uint32_t wfp_from_left(uint32_t edge, uint32_t x)
{
	const uint32_t t = EDGE ? (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x138, 0xf, 0xf, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true); // wave_shr:1
	return __builtin_amdgcn_alignbit(x, t, 16);
}
// right neighbours: (own high half, low half of lane l + 1); lane 63's comes from `edge` (the next set's lane 0) or is unreachable
template<bool EDGE> __device__ __forceinline__ uint32_t wfp_from_right(uint32_t edge, uint32_t x)
{
	const uint32_t t = EDGE ? (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x130, 0xf, 0xf, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true); // wave_shl:1
	return __builtin_amdgcn_alignbit(t, x, 16);
}

// (defined with k_wfa_tb below)
__device__ __forceinline__ uint32_t wfw_tb_byte(const uint32_t *__restrict__ reg, int32_t W, int32_t ph, int32_t sc, int32_t idx);
template<bool WRITE>
__device__ __forceinline__ int32_t wfw_trace(int32_t tl, int32_t ql, const char *__restrict__ ts, const char *__restrict__ qs, int32_t S, int32_t last,
											 const uint32_t *__restrict__ reg, int32_t W, int32_t ph, int32_t lo, uint32_t *__restrict__ out, int32_t n_total);
#define WFP_POOL_BLK 1024 // CIGAR operators a wavefront of k_wfa_fwp takes from the pool at a time for the walks it runs itself (MGA_WFA_FUSE_SLACK in mga_dev.h covers the abandoned tails)

template<int W, int SEQCAP>
__global__ void __launch_bounds__(64) k_wfa_fwp(const int *__restrict__ n_items_p, int cap, const int32_t *__restrict__ list, const mga_wfa_prob_t *__restrict__ prob,
												const char *__restrict__ tseq, const char *__restrict__ qseq, mga_wfa_res_t *__restrict__ res,
												char *__restrict__ tb, long long tb_stride, int *__restrict__ counter, mga_wfa_retry_t rt,
												uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used, int *__restrict__ err, int fuse)
{
	constexpr int JP = (W + 127) / 128; // sets of 128 diagonals (two per lane)
	constexpr int SEQS = SEQCAP + 16;
	constexpr int MROWS = SEQCAP / 32 + 3;
	__shared__ __attribute__((aligned(16))) uint8_t Tg[SEQS], Qg[4 * SEQS];
	__shared__ uint32_t Mk[MROWS][128 * JP]; // mask of diagonal lo + 128 j + 2 lane + h at [word][64 (2 j + h) + lane]: lane l always hits bank l
	const int lane = threadIdx.x;
	const int n_items = min(*n_items_p, cap);
	// ONE lane-0 region per iteration, at the loop's head: it writes the PREVIOUS problem's result and fetches the next work index.  The first version of this kernel had the
	// result write as an `if (lane == 0)` at the loop's end and the fetch as another at its head; the compiler's structurizer joined the two across the back edge into an inner
	// loop in which lanes 1..63 -- with nothing to do in either -- went straight back to reading the work index while lane 0 was still parked at the inner loop's exit: they
	// saw the old index for ever (gpurun's limit, 30 GPU-minutes; a __syncthreads() does not help, for a workgroup of one wavefront it compiles to nothing).
	bool prev = false, done = false; // (uniform)
	int32_t s = 0, lst = 0, pi = 0, item = 0;
	// Round 6 (fuse, MGA_WFA_FUSE_TB=1, not the default -- see mga_dev_wfa_win): the wavefront walks its own alignment as soon as the forward pass ends -- the rows it reads are the ones it just wrote (L2), target and query are the
	// problem it holds -- and leaves the finished result; k_wfa_tb then finds nothing of this rung to do.  [measured, round 6] k_wfa_tb spent 7.3 of its 18.2 ms per 125 000 reads
	// on the 1.47 M problems of the three packed rungs: launches as long as their longest walk (a lane per problem, ~60 dependent trips for a score of 250).
	// The walk is run by ALL lanes on the same values (same loads, same stores: one transaction each) -- a lane-0 region at the loop's end is what the note above warns about.
	bool walked = false;             // (uniform) this problem's result is final
	int32_t w_ncig = 0, w_ub = 0;
	long long w_slot = 0, w_cells = 0;
	long long blk_off = 0;           // the wavefront's block of the CIGAR pool
	int32_t blk_left = 0;
	for (;;) {
		int32_t item_v = 0;
		if (lane == 0) {
			if (prev) {
				mga_wfa_res_t r;
				if (done && walked) {
					if (w_ncig < 0 || w_ncig > w_ub) { r.status = MGA_WFA_POOL_FULL, r.score = -1, r.n_cigar = 0, r.cig_off = 0, r.pad = 0, r.n_iter = 0; res[pi] = r; atomicAdd(err, 1); } // (pool full, or -- cannot happen -- a walk that left the window: fail loudly)
					else { r.score = s, r.n_cigar = w_ncig, r.cig_off = w_slot + (w_ub - w_ncig), r.status = MGA_WFA_OK, r.pad = 0, r.n_iter = w_cells; res[pi] = r; }
				} else if (done) { r.score = s, r.n_cigar = 0, r.cig_off = (int64_t)(uintptr_t)(tb + (long long)item * tb_stride), r.status = MGA_WFA_TB, r.pad = lst | W << 8, r.n_iter = 0; res[pi] = r; }
				else { r.score = -1, r.n_cigar = 0, r.cig_off = 0, r.status = MGA_WFA_RETRY_TIER, r.pad = 0, r.n_iter = 0; res[pi] = r; mga_wfa_give_up(rt, pi); }
			}
			item_v = atomicAdd(counter, 1);
		}
		item = __builtin_amdgcn_readfirstlane(item_v);
		if (item >= n_items) break;
		prev = true;
		pi = __builtin_amdgcn_readfirstlane(list ? list[item] : item);
		// every scalar of the problem goes through readfirstlane: the compiler then keeps the control flow below in scalar branches (a per-lane copy of a uniform value turns
		// every `if` on it into a divergent one and the loops around them into waterfall / exec-mask loops: the hang of round 4's k_gchain_p1)
		const mga_wfa_prob_t pb = prob[pi];
		const int32_t tl = __builtin_amdgcn_readfirstlane(pb.tl), ql = __builtin_amdgcn_readfirstlane(pb.ql), e = ql - tl;
		const long long t_off = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pb.t_off & 0xffffffffLL)) | (long long)__builtin_amdgcn_readfirstlane((int)(pb.t_off >> 32)) << 32;
		const long long q_off = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pb.q_off & 0xffffffffLL)) | (long long)__builtin_amdgcn_readfirstlane((int)(pb.q_off >> 32)) << 32;
		const char *ts = tseq + t_off, *qs = qseq + q_off;
		int32_t lo = 0, bnd = 0;
		if (tl <= SEQCAP && ql <= SEQCAP) { bnd = wfw_window(W, tl, ql, &lo, WFW_SMAX); if (bnd > W + 30) bnd = W + 30; }
		done = false, walked = false, s = 0, lst = 0;
		uint32_t *tbp = (uint32_t*)(tb + (long long)item * tb_stride) + 2 * lane; // this lane's two dwords in the current row (set j: + 128 j)
		if (bnd > 0) {
			// ---- sequences and match masks (as k_wfa_fw)
			WFW_LDS_FENCE();
			for (int32_t m = lane; 4 * m < tl; m += 64) { uint32_t v; __builtin_memcpy(&v, ts + 4 * m, 4); *(uint32_t*)(Tg + 4 * m) = v; }
			for (int32_t m = lane; 4 * m < ql; m += 64) {
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					uint32_t v;
					__builtin_memcpy(&v, qs + 4 * m + c, 4);
					*(uint32_t*)(Qg + c * SEQS + 4 + 4 * m) = v;
					if (m == 0 && c > 0) { uint32_t v0; __builtin_memcpy(&v0, qs, 4); *(uint32_t*)(Qg + c * SEQS) = v0 << (8 * (4 - c)); }
				}
			}
			WFW_LDS_FENCE();
#pragma unroll
			for (int q = 0; q < 2 * JP; ++q) {
				const int32_t d = lo + 128 * (q >> 1) + 2 * lane + (q & 1);
				const int32_t kmin = d < 0 ? -d : 0, kmax = min(tl, ql - d);
				const uint8_t *qc = Qg + (d & 3) * SEQS + 4 + (d & ~3);
				for (int32_t w = 0; w <= (tl >> 5); ++w) {
					uint32_t bits = 0;
#pragma unroll
					for (int q8 = 7; q8 >= 0; --q8) {
						const uint32_t t4 = *(const uint32_t*)(Tg + 32 * w + 4 * q8), q4 = *(const uint32_t*)(qc + 32 * w + 4 * q8);
						asm volatile("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
									 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
									 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
									 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
									 : "+v"(bits) : "v"(t4), "v"(q4) : "vcc");
					}
					const int32_t b_lo = kmin - 32 * w, b_hi = kmax - 32 * w;
					const uint32_t m_hi = b_hi >= 32 ? ~0u : b_hi <= 0 ? 0u : (1u << b_hi) - 1u, m_lo = b_lo <= 0 ? ~0u : b_lo >= 32 ? 0u : ~0u << b_lo;
					Mk[w][64 * q + lane] = bits & m_hi & m_lo;
				}
			}
			WFW_LDS_FENCE();
			// ---- state: two diagonals per register
			uint32_t H[JP][18], E1[JP][3], F1[JP][3], E2[JP][2], F2[JP][2], okv[JP], accA[JP], accB[JP], fc[JP];
#pragma unroll
			for (int j = 0; j < JP; ++j) {
				const int32_t dA = lo + 128 * j + 2 * lane, dB = dA + 1;
#pragma unroll
				for (int a = 0; a < 18; ++a) H[j][a] = 0;
#pragma unroll
				for (int a = 0; a < 3; ++a) E1[j][a] = F1[j][a] = 0;
#pragma unroll
				for (int a = 0; a < 2; ++a) E2[j][a] = F2[j][a] = 0;
				okv[j] = ((dA >= -tl && dA <= ql && 128 * j + 2 * lane < W) ? 0x0000ffffu : 0u) | ((dB >= -tl && dB <= ql && 128 * j + 2 * lane + 1 < W) ? 0xffff0000u : 0u);
				accA[j] = accB[j] = 0;
				if (dA == 0) H[j][2] = (uint32_t)(WFP_BIAS - 1);       // score 0: H[d = 0] = -1 (miniwfa.c:103-119); age 0 of an even step is H[2]
				if (dB == 0) H[j][2] = (uint32_t)(WFP_BIAS - 1) << 16;
				// what the end cell looks like: offset tl - 1 on diagonal e (one half of one lane); 0xffff in a half never matches
				fc[j] = (dA == e ? (uint32_t)(tl - 1 + WFP_BIAS) : 0xffffu) | (dB == e ? (uint32_t)(tl - 1 + WFP_BIAS) : 0xffffu) << 16;
			}
			bool bail = false;
#define HP(j_, a_) H[j_][(a_) + 2 - P]
			auto step = [&](auto Pc) __attribute__((always_inline)) -> bool { // false: the problem is through (done or bail)
				constexpr int P = decltype(Pc)::value;
				// ---- extension of slice s (miniwfa.c:399-411): the two cells of a lane side by side.  A cell that cannot be extended (unreachable, or at the target's end) gets the
				// window "no match at once" (inv = 1), so that its run is 0 without a select; a run that fills its 32-bit window (3 % of the steps) goes on in ONE loop for both
				const int32_t rs = wfw_reach(s);
				uint32_t dfin[JP], nfin[JP]; // a set's cells against the end cell's pattern (a zero half = reached), and the lengths of its two runs
				bool hit = false;
#pragma unroll
				for (int j = 0; j < JP; ++j) {
					const int32_t b0 = lo + 128 * j;
					dfin[j] = ~0u, nfin[j] = 0;
					if (b0 > rs || b0 + 127 < -rs) continue;
					const uint32_t x = HP(j, 0);
					const int32_t tpA = (int32_t)(x & 0xffffu) + (1 - WFP_BIAS), tpB = (int32_t)(x >> 16) + (1 - WFP_BIAS);
					const bool valA = (uint32_t)tpA <= (uint32_t)tl, valB = (uint32_t)tpB <= (uint32_t)tl; // -1 <= offset < tl
					const int32_t wiA = valA ? tpA >> 5 : 0, wiB = valB ? tpB >> 5 : 0;
					const uint32_t *mA = &Mk[wiA][64 * (2 * j) + lane], *mB = &Mk[wiB][64 * (2 * j + 1) + lane];
					uint32_t invA = ~__builtin_amdgcn_alignbit(mA[128 * JP], mA[0], tpA), invB = ~__builtin_amdgcn_alignbit(mB[128 * JP], mB[0], tpB); // (the shift is taken mod 32)
					invA = valA ? invA : 1u, invB = valB ? invB : 1u;
					uint32_t nA = invA ? (uint32_t)__builtin_ctz(invA) : 32u, nB = invB ? (uint32_t)__builtin_ctz(invB) : 32u;
					for (int32_t it = 1; __ballot(invA == 0u || invB == 0u); ++it) {
						const int32_t wa = wiA + it < MROWS - 2 ? wiA + it : MROWS - 2, wb = wiB + it < MROWS - 2 ? wiB + it : MROWS - 2;
						const uint32_t *qA = &Mk[wa][64 * (2 * j) + lane], *qB = &Mk[wb][64 * (2 * j + 1) + lane];
						const uint32_t vA = ~__builtin_amdgcn_alignbit(qA[128 * JP], qA[0], tpA), vB = ~__builtin_amdgcn_alignbit(qB[128 * JP], qB[0], tpB);
						nA += invA == 0u ? (vA ? (uint32_t)__builtin_ctz(vA) : 32u) : 0u, nB += invB == 0u ? (vB ? (uint32_t)__builtin_ctz(vB) : 32u) : 0u;
						invA = invA == 0u ? vA : invA, invB = invB == 0u ? vB : invB;
					}
					nfin[j] = nA | nB << 16;
					const uint32_t xn = x + nfin[j]; // (no carry between the halves: offsets stay below 0x2000 + tl)
					HP(j, 0) = xn;
					dfin[j] = xn ^ fc[j];
					hit = hit || (dfin[j] & 0xffffu) == 0u || dfin[j] < 0x10000u;
				}
				const uint64_t m_fin = __ballot(hit);
				if (m_fin) { // the end cell (it lies on ONE diagonal).  Entered by a gap state and not extended: the traceback starts in that state (miniwfa.c:406-407)
					int32_t flst = 0;
#pragma unroll
					for (int j = 0; j < JP; ++j) {
						if ((dfin[j] & 0xffffu) == 0u) flst = (nfin[j] & 0xffffu) == 0u ? (int32_t)(accA[j] & 7u) : 0;
						else if (dfin[j] < 0x10000u) flst = (nfin[j] >> 16) == 0u ? (int32_t)(accB[j] & 7u) : 0;
					}
					lst = __shfl(flst, (int)__builtin_ctzll(m_fin));
					done = true;
					return false;
				}
				if (s + 1 >= bnd) { bail = true; return false; }
				// ---- slice s + 1 (miniwfa.c:281-308), two diagonals per instruction
				const int32_t rn = wfw_reach(s + 1);
				uint32_t nH[JP], nE1[JP], nF1[JP], nE2[JP], nF2[JP];
#pragma unroll
				for (int j = 0; j < JP; ++j) {
					const int32_t b0 = lo + 128 * j;
					if (b0 > rn || b0 + 127 < -rn) { nH[j] = nE1[j] = nF1[j] = nE2[j] = nF2[j] = 0u; continue; }
#define WFP_L(R, a) (j > 0 ? wfp_from_left<true>((uint32_t)__builtin_amdgcn_readlane((int)R[j > 0 ? j - 1 : 0][a], 63), R[j][a]) : wfp_from_left<false>(0u, R[j][a]))
#define WFP_R(R, a) (j < JP - 1 ? wfp_from_right<true>((uint32_t)__builtin_amdgcn_readlane((int)R[j < JP - 1 ? j + 1 : j][a], 0), R[j][a]) : wfp_from_right<false>(0u, R[j][a]))
					const uint32_t ho1l = WFP_L(H, 5 + 2 - P), e1l = WFP_L(E1, 1), ho2l = WFP_L(H, 15 + 2 - P), e2l = WFP_L(E2, 0);
					const uint32_t ho1r = WFP_R(H, 5 + 2 - P), f1r = WFP_R(F1, 1), ho2r = WFP_R(H, 15 + 2 - P), f2r = WFP_R(F2, 0);
#undef WFP_L
#undef WFP_R
					const uint32_t hx1 = wfp_add(HP(j, 3), WFP_ONE);
					const uint32_t vE1 = wfp_max(ho1l, e1l), vE2 = wfp_max(ho2l, e2l);
					const uint32_t vF1 = wfp_add(wfp_max(ho1r, f1r), WFP_ONE), vF2 = wfp_add(wfp_max(ho2r, f2r), WFP_ONE);
					const uint32_t bits = (wfp_lt(ho1l, e1l) & 0x00080008u) | (wfp_lt(ho2l, e2l) & 0x00200020u) | (wfp_lt(ho1r, f1r) & 0x00100010u) | (wfp_lt(ho2r, f2r) & 0x00400040u);
					const uint32_t ee = wfp_max(vE1, vE2), ff = wfp_max(vF1, vF2), hh = wfp_max(ee, ff);
					const uint32_t ze = (wfp_lt(vE1, vE2) & 0x00020002u) | WFP_ONE;              // vE1 >= vE2 ? 1 : 3
					const uint32_t zf = (wfp_lt(vF1, vF2) & 0x00060006u) ^ 0x00020002u;           // vF1 >= vF2 ? 2 : 4
					uint32_t z = wfp_sel(wfp_lt(ee, ff), zf, ze);                                  // ee >= ff ? ze : zf
					z &= wfp_lt(hx1, hh);                                                          // hx1 >= hh ? 0 : z
					const uint32_t vH = wfp_max(hx1, hh), bz = bits | z;
					accA[j] = accA[j] << 8 | (bz & 0xffu), accB[j] = accB[j] << 8 | (bz >> 16);
					nH[j] = vH & okv[j], nE1[j] = vE1 & okv[j], nF1[j] = vF1 & okv[j], nE2[j] = vE2 & okv[j], nF2[j] = vF2 & okv[j]; // (a diagonal outside the matrix stays unreachable)
				}
#pragma unroll
				for (int j = 0; j < JP; ++j) { // age shift (H: every second step, by two)
					HP(j, -1) = nH[j];
					if (P == 1) {
#pragma unroll
						for (int a = 17; a > 1; --a) H[j][a] = H[j][a - 2];
					}
					E1[j][2] = E1[j][1]; E1[j][1] = E1[j][0]; E1[j][0] = nE1[j];
					F1[j][2] = F1[j][1]; F1[j][1] = F1[j][0]; F1[j][0] = nF1[j];
					E2[j][1] = E2[j][0]; E2[j][0] = nE2[j];
					F2[j][1] = F2[j][0]; F2[j][0] = nF2[j];
				}
				if ((s & 3) == 3) { // a row of traceback dwords is full (one problem per wavefront: the step count IS the score); reachable diagonals only
					const int32_t rr = wfw_reach(s + 1);
#pragma unroll
					for (int j = 0; j < JP; ++j) {
						const int32_t dA = lo + 128 * j + 2 * lane, dB = dA + 1, ad = min(dA < 0 ? -dA : dA, dB < 0 ? -dB : dB);
						if (ad <= rr && 128 * j + 2 * lane < W) { tbp[128 * j] = accA[j]; if (128 * j + 2 * lane + 1 < W) tbp[128 * j + 1] = accB[j]; }
					}
					tbp += W;
				}
				++s;
				return true;
			};
			for (;;) {
				if (!step(std::integral_constant<int, 0>())) break;
				if (!step(std::integral_constant<int, 1>())) break;
			}
#undef HP
			if (done) {
				if (s & 3) { // the rest of the last traceback row (bytes of steps s & ~3 .. s - 1)
					const int32_t rr = wfw_reach(s + 1);
#pragma unroll
					for (int j = 0; j < JP; ++j) {
						const int32_t dA = lo + 128 * j + 2 * lane, dB = dA + 1, ad = min(dA < 0 ? -dA : dA, dB < 0 ? -dB : dB);
						if (ad <= rr && 128 * j + 2 * lane < W) { tbp[128 * j] = accA[j] << (8 * (4 - (s & 3))); if (128 * j + 2 * lane + 1 < W) tbp[128 * j + 1] = accB[j] << (8 * (4 - (s & 3))); }
					}
				}
				if (fuse) {
					__syncthreads(); // (one wavefront: a wait for the rows' stores, no barrier)
					const int32_t ub = (s >> 1) + 4; // most operators an alignment of this score can have (k_wfa_tb)
					if (blk_left < ub) {
						unsigned long long b_ = 0;
						if (lane == 0) b_ = atomicAdd(pool_used, (unsigned long long)WFP_POOL_BLK);
						blk_off = (long long)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b_ & 0xffffffffULL)) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b_ >> 32)) << 32);
						blk_left = WFP_POOL_BLK;
					}
					w_slot = blk_off, w_ub = ub;
					blk_off += ub, blk_left -= ub;
					w_ncig = -1;
					if (w_slot + ub <= pool_cap)
						w_ncig = wfw_trace<true>(tl, ql, ts, qs, s, lst, (const uint32_t*)(tb + (long long)item * tb_stride), W, 0, lo, pool + w_slot, ub);
					// the reference's cell count for this alignment (n_iter, miniwfa.c:421): its band at score s is the reachable diagonals +- 1, clipped to the matrix
					w_cells = 0;
					for (int32_t q = 0; q < s; ++q) { const int32_t w_ = wfw_reach(q) + 1; w_cells += min(w_, tl) + min(w_, ql) + 1; }
					walked = true;
				}
			}
			(void)bail;
		}
	}
}

// ---- PACKED forward pass, SEVERAL problems per wavefront (round 5): the rungs of 32 / 64 diagonals as groups of 16 / 32 lanes with two diagonals per lane -- k_wfa_fw's
// retire / refill machinery (the groups of a wavefront run in lockstep on independent problems) around k_wfa_fwp's step.  Twice the problems per wavefront for a step that
// costs a quarter more; a refill builds two masks per lane instead of one, which is why the 16-diagonal rung (problems of ~15 steps) is not here.
template<int G, int SEQCAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_wfa_fwq(const int *__restrict__ n_items_p, int cap, const int32_t *__restrict__ list, const mga_wfa_prob_t *__restrict__ prob,
											  const char *__restrict__ tseq, const char *__restrict__ qseq, mga_wfa_res_t *__restrict__ res,
											  char *__restrict__ tb, long long tb_stride, int *__restrict__ counter, mga_wfa_retry_t rt)
{
	static_assert(G == 16 || G == 32, "group size");
	constexpr int P = 64 / G;      // problems per wavefront
	constexpr int W = 2 * G;       // diagonals of the window = dwords per traceback row
	constexpr int SEQS = SEQCAP + 16;
	constexpr int QCH = P == 4 ? 32 : 16;
	constexpr int MROWS = SEQCAP / 32 + 3;
	__shared__ __attribute__((aligned(16))) uint8_t Tb[P][SEQS], Qb[P][4 * SEQS];
	__shared__ uint32_t Mk[MROWS][128]; // mask of a lane's diagonal h (lo + 2 gl + h) at [word][64 h + lane]
	const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
	const int n_items = min(*n_items_p, cap);
	uint8_t *const Tg = Tb[grp], *const Qg = Qb[grp];
	const uint32_t nm_first = gl == 0 ? 0u : ~0u, nm_last = gl == G - 1 ? 0u : ~0u;
	const uint64_t gmask = ((1ULL << G) - 1ULL) << (grp * G);

	int32_t st = WFW_IDLE, pi = -1, tl = 0, ql = 0, lo = 0, e = 0, s = 0, bnd = 0, ph = 0;
	uint32_t okv = 0, accA = 0, accB = 0, fc = ~0u; // fc: what the end cell looks like in this lane (offset tl - 1 + bias in the half of diagonal ql - tl, 0xffff elsewhere and while idle)
	uint32_t dfin = ~0u, nfin = 0;                   // the last extension: cells against fc, lengths of the two runs (read when the group retires)
	uint32_t *tbp = 0;
	int32_t it_cur = 0;
	uint32_t H[18], E1[3], F1[3], E2[2], F2[2];
	int32_t t = 0;
	int32_t q_next = 0, q_end = 0, q_base = 0;
	__shared__ int32_t dsc_pi[QCH];
	__shared__ mga_wfa_prob_t dsc_pb[QCH];
	bool q_empty = false, q_drained = false;
#pragma unroll
	for (int a = 0; a < 18; ++a) H[a] = 0;
#pragma unroll
	for (int a = 0; a < 3; ++a) E1[a] = F1[a] = 0;
#pragma unroll
	for (int a = 0; a < 2; ++a) E2[a] = F2[a] = 0;

#define HQ(a_) H[(a_) + 2 - P2]
	auto step = [&](auto Pc) __attribute__((always_inline)) -> bool { // false: the queue is empty and every group is idle
		constexpr int P2 = decltype(Pc)::value;
		// ---- retire the groups that are through, refill idle groups from the queue (as k_wfa_fw)
		if (__ballot(st >= WFW_DONE || (st == WFW_IDLE && !q_empty))) {
			const uint64_t m_done = __ballot(st == WFW_DONE);
			const bool g_done = (m_done & gmask) != 0;
			if (g_done || st == WFW_BAIL) {
				if (g_done) {
					if (t & 3) { // the rest of the last traceback row
						const int32_t rr = wfw_reach(s + 1), dA = lo + 2 * gl, dB = dA + 1;
						if (min(dA < 0 ? -dA : dA, dB < 0 ? -dB : dB) <= rr) { tbp[0] = accA << (8 * (4 - (t & 3))); tbp[1] = accB << (8 * (4 - (t & 3))); }
					}
					if (st == WFW_DONE) {
						// entered by a gap state and not extended: the traceback starts in that state (miniwfa.c:406-407); the state is the byte BEFORE the one the step's recurrence appended
						const int32_t lst = (dfin & 0xffffu) == 0u ? ((nfin & 0xffffu) == 0u ? (int32_t)(accA >> 8 & 7u) : 0) : ((nfin >> 16) == 0u ? (int32_t)(accB >> 8 & 7u) : 0);
						mga_wfa_res_t r;
						r.score = s, r.n_cigar = 0, r.cig_off = (int64_t)(uintptr_t)(tb + (long long)it_cur * tb_stride), r.status = MGA_WFA_TB, r.pad = lst | ph << 4 | W << 8, r.n_iter = 0;
						res[pi] = r;
					}
				} else if (gl == 0) {
					mga_wfa_res_t r;
					r.score = -1, r.n_cigar = 0, r.cig_off = 0, r.status = MGA_WFA_RETRY_TIER, r.pad = 0, r.n_iter = 0;
					res[pi] = r;
					mga_wfa_give_up(rt, pi);
				}
				st = WFW_IDLE, fc = ~0u;
			}
			if (!q_empty) {
				const uint64_t m_idle = __ballot(st == WFW_IDLE && gl == 0);
				if (m_idle) {
					if (q_next == q_end && !q_drained) {
						int32_t b = 0;
						if (lane == 0) b = atomicAdd(counter, QCH);
						b = __builtin_amdgcn_readfirstlane(b);
						q_base = q_next = b, q_end = b + QCH < n_items ? b + QCH : n_items;
						if (q_end <= q_next) q_end = q_next, q_drained = true;
						if (lane < q_end - q_next) {
							const int32_t p_ = list ? list[b + lane] : b + lane;
							dsc_pi[lane] = p_;
							dsc_pb[lane] = prob[p_];
						}
						WFW_LDS_FENCE();
					}
					const int32_t avail = q_end - q_next, n_idle = (int32_t)__popcll(m_idle);
					const int32_t rank = (int32_t)__popcll(m_idle & ((1ULL << (grp * G)) - 1ULL));
					const bool fill = st == WFW_IDLE && rank < avail;
					const int32_t item = q_next + rank, src = fill ? item - q_base : 0;
					q_next += n_idle < avail ? n_idle : avail;
					if (q_drained && q_next == q_end) q_empty = true;
					int32_t ntl = 0, nql = 0;
					const char *ts = tseq, *qs = qseq;
					if (fill) {
						const mga_wfa_prob_t pb = dsc_pb[src];
						pi = dsc_pi[src];
						ntl = pb.tl, nql = pb.ql, ts = tseq + pb.t_off, qs = qseq + pb.q_off;
						it_cur = item;
						tbp = (uint32_t*)(tb + (long long)item * tb_stride) + 2 * gl;
						ph = t & 3, s = 0, e = nql - ntl;
						if (ntl > SEQCAP || nql > SEQCAP) bnd = 0, ntl = nql = 0;
						else { bnd = wfw_window(W, ntl, nql, &lo, WFW_SMAX); if (bnd > W + 30) bnd = W + 30; }
						tl = ntl, ql = nql;
						st = WFW_RUN;
					}
					WFW_LDS_FENCE();
					for (int32_t m0 = 0; __ballot(fill && 4 * m0 < ntl); m0 += G) {
						const int32_t m = m0 + gl;
						if (fill && 4 * m < ntl) { uint32_t v; __builtin_memcpy(&v, ts + 4 * m, 4); *(uint32_t*)(Tg + 4 * m) = v; }
					}
					for (int32_t m0 = 0; __ballot(fill && 4 * m0 < nql); m0 += G) {
						const int32_t m = m0 + gl;
						if (fill && 4 * m < nql) {
#pragma unroll
							for (int c = 0; c < 4; ++c) {
								uint32_t v;
								__builtin_memcpy(&v, qs + 4 * m + c, 4);
								*(uint32_t*)(Qg + c * SEQS + 4 + 4 * m) = v;
								if (m == 0 && c > 0) { uint32_t v0; __builtin_memcpy(&v0, qs, 4); *(uint32_t*)(Qg + c * SEQS) = v0 << (8 * (4 - c)); }
							}
						}
					}
					WFW_LDS_FENCE();
#pragma unroll
					for (int h = 0; h < 2; ++h) { // match masks of the refilled groups' diagonals (as k_wfa_fw), two per lane
						const int32_t d = lo + 2 * gl + h;
						const int32_t kmin = d < 0 ? -d : 0, kmax = min(ntl, nql - d);
						const uint8_t *qc = Qg + (d & 3) * SEQS + 4 + (d & ~3);
						for (int32_t w = 0; __ballot(fill && w <= (ntl >> 5)); ++w) {
							uint32_t bits = 0;
#pragma unroll
							for (int q8 = 7; q8 >= 0; --q8) {
								const uint32_t t4 = *(const uint32_t*)(Tg + 32 * w + 4 * q8), q4 = *(const uint32_t*)(qc + 32 * w + 4 * q8);
								asm volatile("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
											 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
											 : "+v"(bits) : "v"(t4), "v"(q4) : "vcc");
							}
							const int32_t b_lo = kmin - 32 * w, b_hi = kmax - 32 * w;
							const uint32_t m_hi = b_hi >= 32 ? ~0u : b_hi <= 0 ? 0u : (1u << b_hi) - 1u, m_lo = b_lo <= 0 ? ~0u : b_lo >= 32 ? 0u : ~0u << b_lo;
							if (fill && w <= (ntl >> 5)) Mk[w][64 * h + lane] = bits & m_hi & m_lo;
						}
					}
					WFW_LDS_FENCE();
					const uint32_t nmf = fill ? 0u : ~0u;
#pragma unroll
					for (int a = 0; a < 18; ++a) H[a] &= nmf;
#pragma unroll
					for (int a = 0; a < 3; ++a) E1[a] &= nmf, F1[a] &= nmf;
#pragma unroll
					for (int a = 0; a < 2; ++a) E2[a] &= nmf, F2[a] &= nmf;
					if (fill) {
						const int32_t dA = lo + 2 * gl, dB = dA + 1;
						accA = accB = 0;
						okv = ((dA >= -tl && dA <= ql) ? 0x0000ffffu : 0u) | ((dB >= -tl && dB <= ql) ? 0xffff0000u : 0u);
						fc = (dA == e ? (uint32_t)(tl - 1 + WFP_BIAS) : 0xffffu) | (dB == e ? (uint32_t)(tl - 1 + WFP_BIAS) : 0xffffu) << 16;
						if (dA == 0) HQ(0) = (uint32_t)(WFP_BIAS - 1);       // score 0: H[d = 0] = -1 (miniwfa.c:103-119)
						if (dB == 0) HQ(0) = (uint32_t)(WFP_BIAS - 1) << 16;
					}
				}
			}
		}
		if (!__ballot(st == WFW_RUN)) return false;
		// ---- extension of slice s (miniwfa.c:399-411): the two cells of a lane side by side (k_wfa_fwp)
		{
			const uint32_t x = HQ(0);
			const int32_t tpA = (int32_t)(x & 0xffffu) + (1 - WFP_BIAS), tpB = (int32_t)(x >> 16) + (1 - WFP_BIAS);
			const bool valA = st == WFW_RUN && (uint32_t)tpA <= (uint32_t)tl, valB = st == WFW_RUN && (uint32_t)tpB <= (uint32_t)tl;
			const int32_t wiA = valA ? tpA >> 5 : 0, wiB = valB ? tpB >> 5 : 0;
			const uint32_t *mA = &Mk[wiA][lane], *mB = &Mk[wiB][64 + lane];
			uint32_t invA = ~__builtin_amdgcn_alignbit(mA[128], mA[0], tpA), invB = ~__builtin_amdgcn_alignbit(mB[128], mB[0], tpB);
			invA = valA ? invA : 1u, invB = valB ? invB : 1u;
			uint32_t nA = invA ? (uint32_t)__builtin_ctz(invA) : 32u, nB = invB ? (uint32_t)__builtin_ctz(invB) : 32u;
			for (int32_t it = 1; __ballot(invA == 0u || invB == 0u); ++it) {
				const int32_t wa = wiA + it < MROWS - 2 ? wiA + it : MROWS - 2, wb = wiB + it < MROWS - 2 ? wiB + it : MROWS - 2;
				const uint32_t *qA = &Mk[wa][lane], *qB = &Mk[wb][64 + lane];
				const uint32_t vA = ~__builtin_amdgcn_alignbit(qA[128], qA[0], tpA), vB = ~__builtin_amdgcn_alignbit(qB[128], qB[0], tpB);
				nA += invA == 0u ? (vA ? (uint32_t)__builtin_ctz(vA) : 32u) : 0u, nB += invB == 0u ? (vB ? (uint32_t)__builtin_ctz(vB) : 32u) : 0u;
				invA = invA == 0u ? vA : invA, invB = invB == 0u ? vB : invB;
			}
			nfin = nA | nB << 16;
			const uint32_t xn = x + nfin;
			HQ(0) = xn;
			dfin = xn ^ fc;
			if ((dfin & 0xffffu) == 0u || dfin < 0x10000u) st = WFW_DONE; // the end cell (fc matches nothing in any other lane, nor in an idle group)
		}
		if (st == WFW_RUN && s + 1 >= bnd) st = WFW_BAIL;
		// ---- slice s + 1 (miniwfa.c:281-308), two diagonals per instruction; neighbours stay inside the group
		{
#define WFQ_L(x_) __builtin_amdgcn_alignbit((x_), G == 16 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x_), 0x111, 0xf, 0xf, true) : ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x_), 0x138, 0xf, 0xf, true) & nm_first), 16)
#define WFQ_R(x_) __builtin_amdgcn_alignbit(G == 16 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x_), 0x101, 0xf, 0xf, true) : ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x_), 0x130, 0xf, 0xf, true) & nm_last), (x_), 16)
			const uint32_t ho1l = WFQ_L(HQ(5)), e1l = WFQ_L(E1[1]), ho2l = WFQ_L(HQ(15)), e2l = WFQ_L(E2[0]);
			const uint32_t ho1r = WFQ_R(HQ(5)), f1r = WFQ_R(F1[1]), ho2r = WFQ_R(HQ(15)), f2r = WFQ_R(F2[0]);
#undef WFQ_L
#undef WFQ_R
			const uint32_t hx1 = wfp_add(HQ(3), WFP_ONE);
			const uint32_t vE1 = wfp_max(ho1l, e1l), vE2 = wfp_max(ho2l, e2l);
			const uint32_t vF1 = wfp_add(wfp_max(ho1r, f1r), WFP_ONE), vF2 = wfp_add(wfp_max(ho2r, f2r), WFP_ONE);
			const uint32_t bits = (wfp_lt(ho1l, e1l) & 0x00080008u) | (wfp_lt(ho2l, e2l) & 0x00200020u) | (wfp_lt(ho1r, f1r) & 0x00100010u) | (wfp_lt(ho2r, f2r) & 0x00400040u);
			const uint32_t ee = wfp_max(vE1, vE2), ff = wfp_max(vF1, vF2), hh = wfp_max(ee, ff);
			const uint32_t ze = (wfp_lt(vE1, vE2) & 0x00020002u) | WFP_ONE;              // vE1 >= vE2 ? 1 : 3
			const uint32_t zf = (wfp_lt(vF1, vF2) & 0x00060006u) ^ 0x00020002u;           // vF1 >= vF2 ? 2 : 4
			uint32_t z = wfp_sel(wfp_lt(ee, ff), zf, ze);                                  // ee >= ff ? ze : zf
			z &= wfp_lt(hx1, hh);                                                          // hx1 >= hh ? 0 : z
			const uint32_t vH = wfp_max(hx1, hh), bz = bits | z;
			accA = accA << 8 | (bz & 0xffu), accB = accB << 8 | (bz >> 16);
			HQ(-1) = vH & okv;
			if (P2 == 1) {
#pragma unroll
				for (int a = 17; a > 1; --a) H[a] = H[a - 2];
			}
			E1[2] = E1[1]; E1[1] = E1[0]; E1[0] = vE1 & okv;
			F1[2] = F1[1]; F1[1] = F1[0]; F1[0] = vF1 & okv;
			E2[1] = E2[0]; E2[0] = vE2 & okv;
			F2[1] = F2[0]; F2[0] = vF2 & okv;
		}
		if ((t & 3) == 3) { // a row of traceback dwords is full: reachable diagonals only, next row
			if (st != WFW_IDLE) {
				const int32_t rr = wfw_reach(s + 1), dA = lo + 2 * gl, dB = dA + 1;
				if (min(dA < 0 ? -dA : dA, dB < 0 ? -dB : dB) <= rr) { tbp[0] = accA; tbp[1] = accB; }
			}
			tbp += W;
		}
		if (st == WFW_RUN) ++s;
		++t;
		return true;
	};
	for (;;) {
		if (!step(std::integral_constant<int, 0>())) break;
		if (!step(std::integral_constant<int, 1>())) break;
	}
#undef HQ
}

// ---- traceback: one lane per problem (miniwfa.c:329-377) ----------------------------------------------------------------------------

// byte of cell (score sc, window index idx) in a problem's region: rows of W dwords, four scores per dword, the earliest in the top byte.
// [measured and not kept, round 5: the same dwords in TILES of 4 diagonals x 4 rows, so that a 64-byte sector holds 16 scores of 4 neighbouring diagonals instead of 4 scores of
// 16 -- the walk of k_wfa_tb moves a diagonal at a time.  k_wfa_tb fetched 27.9 GB per 125 000 reads instead of 29.0 and ran 17.8 ms instead of 18.1, the forward kernels 1.6 ms
// longer (16-byte pieces per wave store): profiles/r05j_pmc_*_tiled_traceback.txt.  What the walk fetches is not mainly traceback: 15 M problems x (descriptor + result + the two
// sequences + ~4 rows, every one of them a 128-byte line of its own since the lists are sorted by length, not by address) = 1.9 KB per problem.]
__device__ __forceinline__ uint32_t wfw_tb_byte(const uint32_t *__restrict__ reg, int32_t W, int32_t ph, int32_t sc, int32_t idx)
{
	const int32_t p = ph + sc - 1;
	return reg[(size_t)(p >> 2) * W + idx] >> (8 * (3 - (p & 3))) & 0xffu;
}

// walks the alignment from the end cell to the start; WRITE: operators go to out[n_total - 1 - k] (the walk finds them last to first).  Returns their number.
template<bool WRITE>
__device__ __forceinline__ int32_t wfw_trace(int32_t tl, int32_t ql, const char *__restrict__ ts, const char *__restrict__ qs, int32_t S, int32_t last,
											 const uint32_t *__restrict__ reg, int32_t W, int32_t ph, int32_t lo, uint32_t *__restrict__ out, int32_t n_total)
{
	int32_t i = ql - 1, k = tl - 1, sc = S, n = 0, cur_op = -1, cur_len = 0;
#define PUSH(op_, len_) do { \
		if (cur_op == (op_)) cur_len += (len_); \
		else { \
			if (cur_op >= 0) { if (WRITE && n < n_total) out[n_total - 1 - n] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n; } \
			cur_op = (op_), cur_len = (len_); \
		} \
	} while (0)
	while (i >= 0 && k >= 0) {
		if (last == 0) { // a run of matches, eight bases at a time
			int32_t run = 0;
			while (i >= 7 && k >= 7) {
				uint64_t x, y;
				__builtin_memcpy(&x, qs + i - 7, 8);
				__builtin_memcpy(&y, ts + k - 7, 8);
				const uint64_t c = x ^ y;
				const int32_t m = c ? (int32_t)(__builtin_clzll(c) >> 3) : 8;
				i -= m, k -= m, run += m;
				if (m < 8) goto run_done;
			}
			while (i >= 0 && k >= 0 && qs[i] == ts[k]) --i, --k, ++run;
run_done:
			if (run > 0) PUSH(7, run);
			if (i < 0 || k < 0) break;
		}
		if ((uint32_t)((i - k) - lo) >= (uint32_t)W || sc <= 0) return -1; // cannot happen: the walk stays on cells the forward pass computed (fail loudly, never walk out of the region)
		const uint32_t x = wfw_tb_byte(reg, W, ph, sc, (i - k) - lo);
		const int32_t state = last == 0 ? (int32_t)(x & 7) : last;
		const int32_t ext = state > 0 ? (int32_t)(x >> (state + 2) & 1) : 0;
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

__global__ void __launch_bounds__(256) k_wfa_tb(const int *__restrict__ n_p, int cap, const int32_t *__restrict__ list, const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq,
												const char *__restrict__ qseq, mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used, int *__restrict__ err, int W_rung)
{
	const int n = min(*n_p, cap); // the problems of ONE rung's work list (own + arrivals), right after its forward pass: the regions are reused by the next rung
	const int it = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
	const int i = it < n ? list[it] : 0;
	mga_wfa_res_t r;
	r.status = MGA_WFA_OK;
	if (it < n) r = res[i];
	// (round 5: the NEXT rung's forward pass may run next to this walk and a problem that gave up here is on both rungs' lists -- its result then belongs to the other
	// rung's window, says so in `pad` (written with `status` by one 16-byte store), and is left to that rung's walk)
	const bool mine = it < n && r.status == MGA_WFA_TB && (r.pad >> 8) == W_rung;
	if (!__ballot(mine)) return;
	mga_wfa_prob_t pb;
	pb.tl = pb.ql = 0, pb.t_off = pb.q_off = 0;
	if (mine) pb = prob[i];
	const int32_t W = r.pad >> 8, ph = r.pad >> 4 & 3, last = r.pad & 7;
	const uint32_t *reg = (const uint32_t*)(uintptr_t)r.cig_off;
	const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
	// ONE walk: the operators come out last to first, so they are written downwards from the end of a slot sized for the most an alignment of this score can have -- an
	// edit run costs at least 4 (a mismatch), match runs are one more than edit runs, plus the gap that is left when one sequence runs out: <= 2 (S / 4) + 2 -- and the
	// alignment's CIGAR is the slot's tail.  (A counting walk before the reservation read every traceback sector twice: 51 GB per 125 k reads.)
	int32_t lo = 0, n_cig = 0;
	const int32_t ub = mine ? (r.score >> 1) + 4 : 0;
	int32_t incl = ub; // one reservation per wavefront
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
	const int32_t tot = __shfl(incl, 63);
	unsigned long long base = 0;
	if (lane == 63 && tot > 0) base = atomicAdd(pool_used, (unsigned long long)tot);
	base = __shfl(base, 63);
	if (!mine) return;
	const long long slot = (long long)base + (incl - ub);
	if ((long long)(base + (unsigned long long)tot) <= pool_cap) {
		(void)wfw_window(W, pb.tl, pb.ql, &lo, WFW_SMAX);
		n_cig = wfw_trace<true>(pb.tl, pb.ql, ts, qs, r.score, last, reg, W, ph, lo, pool + slot, ub);
	} else n_cig = -1;
	if (n_cig < 0 || n_cig > ub) { // (pool full, or -- cannot happen -- a walk that left the window / more operators than the score allows: fail loudly)
		r.status = MGA_WFA_POOL_FULL, r.score = -1, r.n_cigar = 0, r.cig_off = 0;
		res[i] = r;
		atomicAdd(err, 1);
		return;
	}
	const long long off = slot + (ub - n_cig);
	// the reference's cell count for this alignment (n_iter, miniwfa.c:421): its band at score s is the reachable diagonals +- 1, clipped to the matrix
	long long cells = 0;
	for (int32_t s = 0; s < r.score; ++s) { const int32_t w = wfw_reach(s) + 1; cells += min(w, pb.tl) + min(w, pb.ql) + 1; }
	r.n_cigar = n_cig, r.cig_off = off, r.status = MGA_WFA_OK, r.pad = 0, r.n_iter = cells;
	res[i] = r;
}

// ---- host drivers ---------------------------------------------------------------------------------------------------------

struct wfw_tier_t { int W, n_wg; };
static const wfw_tier_t g_wtier[MGA_WFW_N] = {
	// window, resident workgroups (one wavefront each)
	{  16, 8192 },   // four problems per wavefront; scores < 46 (bound of a centred 16-diagonal window)
	{  32, 8192 },   // two per wavefront; < 62
	{  64, 8192 },   // < 94
	{ 128, 6144 },   // < 158
	{ 192, 6144 },   // < 222
	{ 256, 4096 },   // < 256
};
// traceback rows per problem: four scores each, scores < min(W + 30, 256), + the phase of the first row (<= 3) and the step after the last score
static int wfw_rows(int W) { const int smax = W + 30 < WFW_SMAX ? W + 30 : WFW_SMAX; return (smax + 3) / 4 + 2; }

extern "C" int64_t mga_dev_wfa_win_tb_stride(int wt) { return (int64_t)wfw_rows(g_wtier[wt].W) * g_wtier[wt].W * 4; }

extern "C" int mga_dev_wfa_win(mga_sctx_t *sc, const int *d_n, int n, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
							   mga_wfa_res_t *d_res, char *d_tb, int wt, int slot, mga_wfa_retry_t rt,
							   uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int *d_err)  /* (launched on the context's own stream) */
{
	// MGA_WFA_FUSE_TB=1: a wavefront of the packed rungs walks its own alignment behind its forward pass instead of leaving it to k_wfa_tb.  [measured, round 6,
	// profiles/r06z_fuse_tb.txt, 125 000 reads, isolated] k_wfa_tb 18.1 -> 11.0 ms, but the three rungs 26.5 + 20.9 + 8.5 -> 34.3 + 25.8 + 10.2 ms: at 2-4 waves per SIMD nothing
	// hides a walk's ~60 dependent trips, and the wavefront holds its registers and LDS while it waits.  Not the default; the tests run both forms.
	const char *e_fu = getenv("MGA_WFA_FUSE_TB");
	const int fuse = (e_fu && *e_fu ? atoi(e_fu) : 0) && d_pool != 0 && d_pool_used != 0 && d_err != 0;
	if (n <= 0) return 0;
	if (wt < 0 || wt >= MGA_WFW_N) { mga_set_error("wfa_win: bad tier %d", wt); return -1; }
	const wfw_tier_t &T = g_wtier[wt];
	// MGA_WFA_PACKED=<mask> (read per launch): which rungs run a packed kernel (two diagonals per lane): bit 0 / 1 / 2 the rungs of 128 / 192 / 256 diagonals (k_wfa_fwp, one
	// problem per wavefront), bit 3 / 4 the rungs of 64 / 32 diagonals (k_wfa_fwq, two / four problems per wavefront).  Same results either way (tests/test_gpu_stages.py)
	const char *e_pk = getenv("MGA_WFA_PACKED");
	const int pk_mask = e_pk && *e_pk ? atoi(e_pk) : 7;
	const bool fwq = (wt == 2 && (pk_mask & 8)) || (wt == 1 && (pk_mask & 16));
	const int per = (fwq ? 2 : 1) * (64 / (T.W < 64 ? T.W : 64)); // problems per wavefront
	int wgs = (n + per * 4 - 1) / (per * 4);
	// MGA_WFA_GRID_PCT=<p>: persistent grids of p % of the table's size -- the rest of the wave slots stay free for the kernels of the OTHER chunks in the pipeline
	// (their launches otherwise only get the chip in this one's tail); a tuning knob, the result does not depend on the grid
	const char *e_pct = getenv("MGA_WFA_GRID_PCT"); // (read per launch: six launches per chunk)
	const int grid_pct = e_pct && atoi(e_pct) > 0 ? atoi(e_pct) : 100;
	const int cap_wg = (int)((long long)T.n_wg * grid_pct / 100) > 64 ? (int)((long long)T.n_wg * grid_pct / 100) : 64;
	if (wgs > cap_wg) wgs = cap_wg;
	if (wgs < 1) wgs = 1;
	hipStream_t st = (hipStream_t)mga_wfa_stream(sc, slot);
	int *d_counter = (int*)((char*)sc->wfa_cnt.p + 64 * slot);
	const long long stride = (long long)mga_dev_wfa_win_tb_stride(wt);
	mga_prof_begin(st, MGA_K_WFAW0 + wt);
#define LAUNCH(GG, JJ, SEQ) hipLaunchKernelGGL((k_wfa_fw<GG, JJ, SEQ>), dim3(wgs), dim3(64), 0, st, d_n, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_tb, stride, d_counter, rt)
	// (default mask 7, measured per 125 000 reads, profiles/r05n_packed_sweep.txt: 128: 45.9 -> 26.9 ms, 192: 22.8 -> 21.1 ms, 256: 12.5 -> 8.5 ms)
#define LAUNCHQ(GG, SEQ) hipLaunchKernelGGL((k_wfa_fwq<GG, SEQ>), dim3(wgs), dim3(64), 0, st, d_n, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_tb, stride, d_counter, rt)
#define LAUNCHP(WW, SEQ) hipLaunchKernelGGL((k_wfa_fwp<WW, SEQ>), dim3(wgs), dim3(64), 0, st, d_n, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_tb, stride, d_counter, rt, d_pool, (long long)pool_cap, d_pool_used, d_err, fuse)
	if (wt == 0) LAUNCH(16, 1, 128);
	else if (wt == 1) { if (pk_mask & 16) LAUNCHQ(16, 192); else LAUNCH(32, 1, 192); }
	else if (wt == 2) { if (pk_mask & 8) LAUNCHQ(32, 256); else LAUNCH(64, 1, 256); }
	else if (wt == 3) { if (pk_mask & 1) LAUNCHP(128, 384); else LAUNCH(64, 2, 384); }
	else if (wt == 4) { if (pk_mask & 2) LAUNCHP(192, 384); else LAUNCH(64, 3, 384); }
	else { if (pk_mask & 4) LAUNCHP(256, 512); else LAUNCH(64, 4, 512); }
#undef LAUNCH
#undef LAUNCHP
#undef LAUNCHQ
	mga_prof_end(st, MGA_K_WFAW0 + wt);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int mga_dev_wfa_traceback(mga_sctx_t *sc, void *stream, const int *d_n, int cap, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq, mga_wfa_res_t *d_res,
									 uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int *d_err, int wt)
{
	if (cap <= 0) return 0;
	if (wt < 0 || wt >= MGA_WFW_N) { mga_set_error("wfa_traceback: bad tier %d", wt); return -1; }
	hipStream_t st = (hipStream_t)(stream ? stream : sc->stream);
	mga_prof_begin(st, MGA_K_WFATB);
	hipLaunchKernelGGL(k_wfa_tb, dim3((cap + 255) / 256), dim3(256), 0, st, d_n, cap, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, d_err, g_wtier[wt].W);
	mga_prof_end(st, MGA_K_WFATB);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
