// k_rmq.hip -- forward pass of the RMQ chainer (mg_lchain_rmq, reference lchain.c:252-357) over the (segment, strand) RUNS of a contig's anchors: the primary chainer of -x asm.
//
// The reference inserts every anchor into an AVL tree keyed by (query position, index), erases it when it falls out of the target window, and per anchor i asks the tree for
// the minimum-priority key with y in (y_i - max_dist, y_i - 1); on a non-exact answer it also walks the keys of an inner window in descending (y, index) order with the
// order-dependent skip heuristic of the first-pass DP.  The tree is emptied whenever the target (segment, strand) changes (lchain.c:294,304), so runs of whole groups are
// independent (rmq.c: mga_lchain_rmq_fwd takes them one by one on host threads: 49 of the 52 host CPU-seconds of a 500 Mbp -x asm job, round 4).
//
// Here: ONE WAVEFRONT PER RUN, thousands of runs in flight.  Nothing of the tree survives -- only what its answers depend on:
//   * the range-minimum query is a wave-wide arg-min over the window [st, i0) (priorities and query positions as flat arrays, 12 bytes per candidate, coalesced).  The tree's
//     answer IS the arg-min whenever the minimum is unique; two candidates with the same (double) priority would be told apart by the AVL tree's shape, which no
//     order-independent formulation reproduces: the wave DETECTS the tie, gives the run up (status 1), and the host runs its exact tree over that run (rmq.c).
//     [measured, round 5, MGA_RQ_TIE_STATS=1 on 4 x 3 Mbp contigs vs a 20 Mbp graph: 0 tied queries in 4 076 743, 0 of 2 462 runs]
//   * the tree-size cap (cap_rmq_size) is the window's length: the tree holds exactly the anchors [st, i0);
//   * the inner walk only asks ORDER questions: the candidates (<= RQ_CAND_MAX, else status 2) are rank-sorted by (y, index) in LDS and the skip heuristic is replayed from
//     ballot masks 64 candidates at a time, exactly like the long-join rescue of k_lchain.hip (whose single block of 64 this generalises).
// Same float / double arithmetic as the reference, no FMA contraction.  Output: f, p (absolute indices into the read's anchor array, -1: none), v -- what
// mg_chain_backtrack (hchain.c) consumes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_lcscan.h"

#define RQ_CAND_MAX 512

struct rq_par_t { int32_t max_dist, max_dist_inner, bw, max_skip, cap; float pen_gap, pen_skip; };
struct rq_run_t { long long beg, end, base; }; // a run [beg, end) of the chunk's anchor array; base = first anchor of the run's READ (p[] and the (y, index) keys count from there)

// one run [beg, end) of a read's x-sorted anchors; indices are positions in a[] (the CHUNK's array: several reads' runs share a launch), `base` = the read's first anchor.
// Returns 0, or why the host has to redo the run.
__device__ int rq_run(const mg128_t *__restrict__ a, int32_t beg, int32_t end, int32_t base, const rq_par_t &R, int32_t *__restrict__ f, long long *__restrict__ p, int32_t *__restrict__ v,
					  int32_t *__restrict__ t, double *__restrict__ pri, int32_t *__restrict__ ys, int32_t *cand_j, int32_t *cand_y, int32_t *sorted_j, int lane)
{
	int32_t max_dist = R.max_dist, max_dist_inner = R.max_dist_inner;
	if (max_dist < R.bw) max_dist = R.bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	int32_t i0 = beg, st = beg, st_in = beg;
	for (int32_t i = beg; i < end; ++i) {
		const uint64_t xi = a[i].x, yi = a[i].y;
		const int32_t yi32 = (int32_t)yi;
		// anchors with a smaller x become available (lchain.c:279-293): their priorities are final now
		if (i0 < i && a[i0].x != xi) {
			for (int32_t j = i0 + lane; j < i; j += 64) {
				const mg128_t aj = a[j];
				pri[j] = -((double)f[j] + 0.5 * (double)R.pen_gap * (double)((int32_t)aj.x + (int32_t)aj.y));
				ys[j] = (int32_t)aj.y;
			}
			i0 = i;
			__syncthreads();
		}
		// windows (lchain.c:294-312): the outer tree holds [st, i0), the inner one [st_in, i0); the size cap is the window's length
		while (st < i) {
			const uint64_t xs = a[st].x;
			if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)max_dist || (i0 > st ? i0 - st : 0) > R.cap) ++st; else break;
		}
		if (max_dist_inner > 0)
			while (st_in < i) {
				const uint64_t xs = a[st_in].x;
				if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)max_dist_inner || (st_in < i0 ? i0 - st_in : 0) > R.cap) ++st_in; else break;
			}
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1;
		// (1) range-minimum query: keys in [(y_i - max_dist, INT32_MAX), (y_i - 1, 0)] (lchain.c:313-316)
		const int32_t ylo = yi32 - max_dist, yhi = yi32 - 1;
		double bp = 0.0;
		int32_t bj = -1;
		bool tie = false;
		// (four blocks of 64 candidates per trip: eight independent loads in flight instead of a round trip per block -- [measured, round 5] 7.6 us per anchor with one)
#define RQ_CAND(j_, yj_, pj_) do { if (((yj_) > ylo && (yj_) < yhi) || ((j_) == base && (yj_) == yhi)) { /* (the key (y_i - 1, 0): index 0 of the READ's array) */ \
				if (bj < 0 || (pj_) < bp) bp = (pj_), bj = (j_), tie = false; else if ((pj_) == bp) tie = true; } } while (0)
		int32_t j = st + lane;
		for (; j + 192 < i0; j += 256) {
			const int32_t y0 = ys[j], y1 = ys[j + 64], y2 = ys[j + 128], y3 = ys[j + 192];
			const double p0 = pri[j], p1 = pri[j + 64], p2 = pri[j + 128], p3 = pri[j + 192];
			RQ_CAND(j, y0, p0); RQ_CAND(j + 64, y1, p1); RQ_CAND(j + 128, y2, p2); RQ_CAND(j + 192, y3, p3);
		}
		for (; j < i0; j += 64) { const int32_t yj = ys[j]; const double pj = pri[j]; RQ_CAND(j, yj, pj); }
#undef RQ_CAND
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
			if (__popcll(at_min) > 1 || __ballot(bj >= 0 && bp == m && tie)) return 1; // equal priorities: the AVL shape would decide
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
					if (j < i0) { yj = ys[j]; c = yj <= yhi && yj >= ymin; }
					const uint64_t mc = __ballot(c);
					const int32_t pos = m_c + (int32_t)__popcll(mc & mga_lanemask_lt());
					if (c && pos < RQ_CAND_MAX) cand_j[pos] = j, cand_y[pos] = yj;
					m_c += (int32_t)__popcll(mc);
				}
				if (m_c > RQ_CAND_MAX) return 2; // more candidates than the rank sort below holds
				__syncthreads();
				if (m_c > 0) {
					// rank by descending (y, j): keys are unique
					for (int32_t c0 = 0; c0 < m_c; c0 += 64) {
						const int32_t me = c0 + lane;
						const int32_t myj = me < m_c ? cand_j[me] : -1, myy = me < m_c ? cand_y[me] : 0;
						int32_t rank = 0;
						for (int32_t k = 0; k < m_c; ++k) {
							const int32_t ky = cand_y[k], kj = cand_j[k];
							rank += (ky > myy || (ky == myy && kj > myj)) ? 1 : 0;
						}
						if (me < m_c) sorted_j[rank] = myj;
					}
					__syncthreads();
					int32_t n_skip = 0;
					for (int32_t c0 = 0; c0 < m_c; c0 += 64) { // 64 candidates at a time, in visiting order = lane order; max_f, max_j and the skip counter carry over
						const int32_t me = c0 + lane;
						const int32_t j = me < m_c ? sorted_j[me] : -1;
						int32_t sc2 = LC_NONE;
						long long pj = -1;
						bool valid = false;
						if (j >= 0) {
							bool ex2;
							int32_t w2;
							const mg128_t aj = a[j];
							sc2 = f[j] + lc_score_simple(xi, yi, aj.x, aj.y, R.pen_gap, R.pen_skip, &ex2, &w2);
							valid = w2 <= R.bw;
							pj = p[j];
						}
						if (valid && pj >= 0) t[base + pj] = i; // (p[] counts from the read's first anchor) marks only reach candidates with a smaller y, i.e. visited later (this block or a later one)
						__syncthreads();
						const bool hit_t = valid && t[j] == i;
						const int32_t pm = lc_scan_max(valid ? sc2 : INT32_MIN, INT32_MIN);
						int32_t exm = lc_prev_lane(pm, INT32_MIN);
						if (exm < max_f) exm = max_f;
						const bool improve = valid && sc2 > exm;
						const uint64_t m_imp = __ballot(improve);
						const int cut_lane = lc_skip_replay(improve, hit_t && !improve, R.max_skip, &n_skip);
						const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
						const uint64_t imp_b = m_imp & before;
						if (imp_b) {
							const int bl = 63 - (int)__clzll(imp_b);
							max_f = __shfl(sc2, bl), max_j = __shfl(j, bl);
						}
						if (cut_lane < 64) break;
					}
				}
				__syncthreads();
			}
		}
		int32_t vi = max_f;
		if (max_j >= 0) { const int32_t vj = v[max_j]; if (vj > max_f) vi = vj; }
		if (lane == 0) { f[i] = max_f; p[i] = max_j >= 0 ? (long long)(max_j - base) : -1LL; v[i] = vi; }
		__syncthreads();
	}
	return 0;
}

// persistent wavefronts, runs drawn from a queue in the order the host lists them (longest first); status[r] = 0, or 1 (tied priorities) / 2 (inner window too large): host
__global__ void __launch_bounds__(64) k_rmq_fwd(int n_order, const rq_run_t *__restrict__ runs, const int32_t *__restrict__ order, const mg128_t *__restrict__ a, rq_par_t R,
												 int32_t *__restrict__ f, long long *__restrict__ p, int32_t *__restrict__ v, int32_t *__restrict__ t, double *__restrict__ pri,
												 int32_t *__restrict__ ys, int32_t *__restrict__ status, int *__restrict__ counter)
{
	__shared__ int32_t cand_j[RQ_CAND_MAX], cand_y[RQ_CAND_MAX], sorted_j[RQ_CAND_MAX];
	const int lane = threadIdx.x;
	// ONE lane-0 region per iteration, at the loop's head: it writes the previous run's status and fetches the next work index.  (As two regions -- the status at the loop's end,
	// the fetch at its head -- the compiler's structurizer may join them across the back edge into an inner loop that lanes 1..63 skip: k_wfa_w.hip's k_wfa_fwp hung that way.)
	int r = -1, rc = 0;
	for (;;) {
		int k = 0;
		if (lane == 0) {
			if (r >= 0) status[r] = rc;
			k = atomicAdd(counter, 1);
		}
		k = __builtin_amdgcn_readfirstlane(k);
		if (k >= n_order) break;
		r = order ? __builtin_amdgcn_readfirstlane(order[k]) : k;
		const rq_run_t run = runs[r];
		rc = rq_run(a, (int32_t)run.beg, (int32_t)run.end, (int32_t)run.base, R, f, p, v, t, pri, ys, cand_j, cand_y, sorted_j, lane);
		__syncthreads();
	}
}

// forward pass over the runs d_runs[d_order[0 .. n_order)] of a CHUNK's anchor array d_a (n_total < 2^31 anchors; the runs of several reads may share a launch).  The caller has
// zeroed d_t over the runs' reads; d_pri (doubles) and d_ys (int32) are scratch of n_total entries.
extern "C" int mga_dev_rmq_fwd(mga_sctx_t *sc, int64_t n_total, const mg128_t *d_a, int n_order, const void *d_runs, const int32_t *d_order, int max_dist, int max_dist_inner, int bw,
							   int max_skip, int cap, float pen_gap, float pen_skip, int32_t *d_f, int64_t *d_p, int32_t *d_v, int32_t *d_t, double *d_pri, int32_t *d_ys,
							   int32_t *d_status, int *d_counter)
{
	if (n_total <= 0 || n_order <= 0) return 0;
	if (n_total >= 0x7fffffffLL) { mga_set_error("rmq_fwd: more than 2^31 anchors in one chunk"); return -1; }
	hipStream_t st = (hipStream_t)sc->stream;
	rq_par_t R;
	R.max_dist = max_dist, R.max_dist_inner = max_dist_inner, R.bw = bw, R.max_skip = max_skip, R.cap = cap, R.pen_gap = pen_gap, R.pen_skip = pen_skip;
	static int n_wg = 0;
	if (n_wg == 0) { const char *e = getenv("MGA_RMQ_WAVES"); n_wg = e && atoi(e) > 0 ? atoi(e) : 8192; }
	MGA_HIP_CHECK(hipMemsetAsync(d_counter, 0, 4, st));
	mga_prof_begin(sc->stream, MGA_K_LCHAIN);
	hipLaunchKernelGGL(k_rmq_fwd, dim3(n_order < n_wg ? n_order : n_wg), dim3(64), 0, st, n_order, (const rq_run_t*)d_runs, d_order, d_a, R, d_f, (long long*)d_p, d_v, d_t, d_pri, d_ys, d_status, d_counter);
	mga_prof_end(sc->stream, MGA_K_LCHAIN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
