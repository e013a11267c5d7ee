// k_wfa_sched.hip -- device-side scheduling of the gap-filling problems over the WFA capacity tiers.
//
// Every problem starts in the cheapest tier its length suggests; a problem that outgrows its tier is appended by
// the kernel itself to the next tier's work list (mga_wfa_retry_t).  The host only reads a handful of counters
// per pass -- it never walks the (millions of) problems.
//   pass 0:  counting sort of the problem ids by (tier, decreasing length): k_wfa_bin_count -> k_wfa_bin_scan ->
//            k_wfa_bin_scatter.  Longest-first inside a tier starts the slow problems early instead of leaving
//            them as the tail of the launch.
//   pass p:  the retry lists written during pass p-1, one tier up (double-buffered work lists).
// All tiers of a pass run concurrently on their own streams (mga_wfa_fork / mga_wfa_join).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <time.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "mga_dev.h"
#include "wfachain.h"
#include "dev_common.h"

#define WFS_NBIN (MGA_WFA_N_SLOT * 1024)

// The ladder (round 3).  Rungs W0-W5 are the WINDOWED tiers of k_wfa_w.hip (16 / 32 / 64 / 128 / 192 / 256 diagonals, exact for scores below the
// window's bound, < 256 in any case); what they cannot decide -- scores >= 256, sequences beyond 512 bases -- goes to the register tiers with the
// reference's full band (k_wfa_r.hip: 512 / 1024 / 2048 diagonals) and the HBM tiers (k_wfa.hip).  MGA_WFA_LADDER=old brings back the round-2 ladder
// (register tiers 64 .. 2048, HBM) for A/B measurements; the chained fallback's sub-problems always use it (their results must sit in the pool).
// A rung: kind (0 windowed, 1 register / HBM), tier index of that kind, the longest sequence that STARTS there, the rung a problem that gives up goes to.
// First rungs by length, [measured on 359 519 gaps of the bench workload, cost model = steps x lanes-share]: 90 % of the gaps of length L score <= 0.68 L,
// and a gap that gives up has wasted its steps up to the bound, so a gap starts in the narrowest window whose bound most gaps of its length stay under.
struct wfs_rung_t { int kind, idx, maxlen, next; };
struct wfs_ladder_t { int n; wfs_rung_t r[MGA_WFA_N_SLOT]; };
struct wfs_thr_t { int32_t n, n_stable, t[MGA_WFA_N_SLOT]; }; // n_stable: the first rungs whose lists are in problem order instead of longest-first (below)
static const wfs_ladder_t g_ladder_win = { 11, {
	{ 0, 0, 71, 1 }, { 0, 1, 111, 2 }, { 0, 2, 167, 3 }, { 0, 3, 255, 4 }, { 0, 4, 343, 5 }, { 0, 5, 420, 6 },
	// the register tiers are windowed too (their band is the reference's, trimming included, clipped to the window): 512 diagonals decide scores < 542, ...
	{ 1, 4, 780, 7 }, { 1, 5, 1500, 8 }, { 1, 6, 3000, 9 }, { 1, 7, 0x7fffffff, 10 }, { 1, 8, 0x7fffffff, -1 } } };
static const wfs_ladder_t g_ladder_old = { 9, {
	{ 1, 0, 64, 1 }, { 1, 1, 128, 2 }, { 1, 2, 192, 3 }, { 1, 3, 256, 4 }, { 1, 4, 512, 5 }, { 1, 5, 2048, 6 }, { 1, 6, 4096, 7 }, { 1, 7, 0x7fffffff, 8 }, { 1, 8, 0x7fffffff, -1 } } };

static const wfs_ladder_t *wfs_ladder(int force_old)
{
	static wfs_ladder_t L[2];
	static int init = 0;
	if (!init) { // MGA_WFA_THR="a,b,c,...": first-rung length limits, in rung order (tuning aid)
		const char *e = getenv("MGA_WFA_LADDER"), *thr = getenv("MGA_WFA_THR");
		L[0] = g_ladder_win, L[1] = g_ladder_old;
		if (e && strcmp(e, "old") == 0) L[0] = g_ladder_old;
		for (int k = 0; thr && *thr && k < L[0].n; ++k) { L[0].r[k].maxlen = atoi(thr); thr = strchr(thr, ','); if (thr) ++thr; }
		init = 1;
	}
	return &L[force_old ? 1 : 0];
}
__host__ __device__ __forceinline__ int wfs_first_rung(int32_t tl, int32_t ql, const wfs_thr_t &T)
{
	const int32_t m = tl > ql ? tl : ql;
	for (int k = 0; k < T.n - 1; ++k) if (m <= T.t[k]) return k;
	return T.n - 1;
}

extern "C" int mga_wfa_first_tier(int32_t tl, int32_t ql) // (round-2 stage API: first register tier by length)
{
	const wfs_ladder_t *L = &g_ladder_old;
	const int32_t m = tl > ql ? tl : ql;
	for (int k = 0; k < L->n; ++k) if (m <= L->r[k].maxlen) return k;
	return L->n - 1;
}

__global__ void __launch_bounds__(1024) k_wfa_bin_count(int n, const int32_t *__restrict__ ids /* NULL: problems 0..n-1 */, const mga_wfa_prob_t *__restrict__ prob, int32_t *__restrict__ key, int *__restrict__ hist, wfs_thr_t T)
{
	__shared__ int h[WFS_NBIN];
	for (int i = threadIdx.x; i < WFS_NBIN; i += blockDim.x) h[i] = 0;
	__syncthreads();
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int pi = ids ? ids[i] : i;
		const int32_t tl = prob[pi].tl, ql = prob[pi].ql;
		const int rg = wfs_first_rung(tl, ql, T);
		int lb = (tl + ql) >> 3;
		if (lb > 1023) lb = 1023;
		if (rg < T.n_stable) lb = 0; // one bin: the scatter keeps this rung's problems in index order
		const int k = rg << 10 | (1023 - lb);
		key[i] = k;
		atomicAdd(&h[k], 1);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < WFS_NBIN; i += blockDim.x) if (h[i]) atomicAdd(&hist[i], h[i]);
}

// counts -> exclusive offsets in place (the scatter's cursors); tier_off[t] = first list slot of tier t, tier_off[N_TIER] = n
__global__ void __launch_bounds__(1024) k_wfa_bin_scan(int *__restrict__ hist, int *__restrict__ tier_off)
{
	constexpr int PER = WFS_NBIN / 1024;
	__shared__ int wsum[16];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	int v[PER], sum = 0;
#pragma unroll
	for (int i = 0; i < PER; ++i) { v[i] = hist[tid * PER + i]; sum += v[i]; }
	int inc = sum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
	if (lane == 63) wsum[wid] = inc;
	__syncthreads();
	int base = 0;
	for (int w = 0; w < wid; ++w) base += wsum[w];
	int run = base + inc - sum;
#pragma unroll
	for (int i = 0; i < PER; ++i) {
		const int b = tid * PER + i;
		hist[b] = run;
		if ((b & 1023) == 0) tier_off[b >> 10] = run;
		run += v[i];
	}
	if (tid == 1023) tier_off[MGA_WFA_N_SLOT] = run;
}

// tile of 8192 ids per workgroup: LDS histogram of the tile, ONE global atomic per non-empty bin to reserve its slots,
// then LDS atomics hand out the slots (4.8 M global atomics on ~20 hot bins took 14 ms; this takes <1 ms)
struct wfs_shift_t { int32_t s[MGA_WFA_N_SLOT]; }; // rung r's region of the list starts s[r] slots after where the compact order would put it (room for arrivals from below)
// Round 5: the lists of the first `n_stable` rungs (16 / 32 / 64 diagonals: 97 % of the problems, each a few microseconds of a launch that lasts milliseconds, so longest-first
// buys them nothing) keep the problems' INDEX order -- runs of ~4 000 ascending ids per block of 8 192.  Neighbouring lanes of the traceback walk (a lane per list entry) and
// neighbouring refills of the forward kernels then touch neighbouring descriptors, results and sequences: [measured, round 4] the walk fetched 1.9 KB per problem, a 128-byte
// line of its own for each of descriptor, result, two sequences and ~4 traceback rows.  A rung's rank inside the block comes from ballots and per-wave counts, round by round.
__global__ void __launch_bounds__(1024) k_wfa_bin_scatter(int n, const int32_t *__restrict__ ids, const int32_t *__restrict__ key, int *__restrict__ cursor, int32_t *__restrict__ list, wfs_shift_t shift, int n_stable)
{
	__shared__ int cnt[WFS_NBIN], base[WFS_NBIN];
	__shared__ int wcnt[3][16];
	const int t0 = blockIdx.x * 8192, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < WFS_NBIN; i += 1024) cnt[i] = 0;
	__syncthreads();
	int k[8], rank[8];
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const int i = t0 + r * 1024 + threadIdx.x;
		k[r] = i < n ? key[i] : -1;
		rank[r] = -1;
		if (k[r] >= 0) atomicAdd(&cnt[k[r]], 1);
	}
	if (n_stable > 0) { // (uniform) rank of an entry among its block's entries of the same rung, in index order: i = t0 + 1024 r + thread
		int run[3] = { 0, 0, 0 };
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			const int rg = (k[r] >= 0 && (k[r] >> 10) < n_stable) ? k[r] >> 10 : -1;
			uint64_t b[3];
#pragma unroll
			for (int x = 0; x < 3; ++x) { b[x] = __ballot(rg == x); if (lane == 0) wcnt[x][wid] = (int)__popcll(b[x]); }
			__syncthreads();
#pragma unroll
			for (int x = 0; x < 3; ++x) {
				int pre = 0, tot = 0;
				for (int w = 0; w < 16; ++w) { const int c = wcnt[x][w]; tot += c; if (w < wid) pre += c; }
				if (rg == x) rank[r] = run[x] + pre + (int)__popcll(b[x] & mga_lanemask_lt());
				run[x] += tot;
			}
			__syncthreads();
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < WFS_NBIN; i += 1024) if (cnt[i]) base[i] = atomicAdd(&cursor[i], cnt[i]);
	__syncthreads();
#pragma unroll
	for (int r = 0; r < 8; ++r)
		if (k[r] >= 0) {
			const int at = rank[r] >= 0 ? rank[r] : atomicSub(&cnt[k[r]], 1) - 1;
			list[base[k[r]] + shift.s[k[r] >> 10] + at] = ids ? ids[t0 + r * 1024 + threadIdx.x] : t0 + r * 1024 + threadIdx.x;
		}
}

__global__ void __launch_bounds__(256) k_wfa_sum_cells(int n, const mga_wfa_res_t *__restrict__ res, unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long ws[4];
	unsigned long long s = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += (unsigned long long)res[i].n_iter;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

// ---- CIGARs back in PROBLEM order ----------------------------------------------------------------------------
// The kernels append CIGARs to the pool in completion order; the host stitches them per read, i.e. in problem
// order.  Gathering them on the device turns ~120 cache misses per read on the host into one sequential stream.
__global__ void __launch_bounds__(256) k_wfa_ncig(int n, const mga_wfa_res_t *__restrict__ res, int32_t *__restrict__ ncig)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ncig[i] = res[i].status == MGA_WFA_OK ? res[i].n_cigar : 0;
}

// one wavefront per 64 consecutive problems: lanes stride over the group's output range; the owner of an output
// slot is found by a 6-step binary search over the 64 offsets held one per lane
__global__ void __launch_bounds__(256) k_wfa_gather(int n, const mga_wfa_res_t *__restrict__ res, const int64_t *__restrict__ off,
													const uint32_t *__restrict__ pool, uint32_t *__restrict__ ord)
{
	const int lane = threadIdx.x & 63;
	const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
	if (g >= n) return;
	const int me = g + lane < n ? g + lane : n - 1;
	const long long o_l = g + lane < n ? off[g + lane] : off[n];
	const long long c_l = res[me].cig_off;
	const long long o_end = off[g + 64 < n ? g + 64 : n];
	for (long long o0 = __shfl(o_l, 0); o0 < o_end; o0 += 64) { // uniform trip count: the shuffles below need every lane
		const long long o = o0 + lane;
		int lo = 0;
#pragma unroll
		for (int step = 32; step > 0; step >>= 1) {
			const long long v = __shfl(o_l, lo + step);
			if (v <= o) lo += step; // offsets beyond the last problem equal off[n] > o
		}
		const long long src = __shfl(c_l, lo) + (o - __shfl(o_l, lo));
		if (o < o_end) ord[o] = pool[src];
	}
}

extern "C" int mga_dev_wfa_gather(mga_sctx_t *sc, int n, const mga_wfa_res_t *d_res, const uint32_t *d_pool, int32_t *d_ncig, int64_t *d_off, uint32_t *d_ord,
								  int64_t ord_cap, int64_t *h_total)
{
	*h_total = 0;
	if (n <= 0) return 0;
	hipStream_t st = (hipStream_t)sc->stream;
	long long tot = 0;
	hipLaunchKernelGGL(k_wfa_ncig, dim3((n + 255) / 256), dim3(256), 0, st, n, d_res, d_ncig);
	MGA_HIP_CHECK(hipGetLastError());
	if (mga_dev_scan_i32_to_i64(sc, d_ncig, n, d_off) < 0) return -1;
	if (mga_d2h_s(sc, &tot, d_off + n, 8) < 0 || mga_ssync(sc) < 0) return -1;
	if (tot > ord_cap) { mga_set_error("WFA gather: %lld operators exceed the buffer of %lld", tot, (long long)ord_cap); return -1; }
	hipLaunchKernelGGL(k_wfa_gather, dim3((n + 255) / 256), dim3(256), 0, st, n, d_res, (const int64_t*)d_off, d_pool, d_ord);
	MGA_HIP_CHECK(hipGetLastError());
	*h_total = (int64_t)tot;
	return 0;
}

extern "C" int mga_dev_wfa_tier(mga_sctx_t *sc, const int *d_n, int cap, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
								mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt)
{
	if (cap - first <= 0) return 0;
	if (tier < 7) return mga_dev_wfa_reg(sc, d_n, cap, first, slot, stream, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, tier, rt);
	return mga_dev_wfa(sc, d_n, cap, first, slot, stream, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, tier - 7, rt); /* HBM tiers with 4096 / 32768 diagonals */
}

// problems a sweep left undecided (their rung's list was full): collected for the next sweep
__global__ void __launch_bounds__(256) k_wfa_collect_open(int n, const mga_wfa_res_t *__restrict__ res, int32_t *__restrict__ list, int *__restrict__ cnt)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool open = i < n && (res[i].status == MGA_WFA_RETRY_TIER || res[i].status == MGA_WFA_PENDING);
	const uint64_t m = __ballot(open);
	if (m == 0) return;
	int base = 0;
	if ((threadIdx.x & 63) == 0) base = atomicAdd(cnt, (int)__popcll(m));
	base = __shfl(base, 0);
	if (open) list[base + (int)__popcll(m & mga_lanemask_lt())] = i;
}
__global__ void __launch_bounds__(256) k_wfa_mark_pending(int n, mga_wfa_res_t *__restrict__ res)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) res[i].status = MGA_WFA_PENDING;
}


// ---- chained fallback ---------------------------------------------------------------------------
// mwf_wfa_auto() (miniwfa.c:824-834): when the exact WFA passes 1e8 wavefront cells, mwf_wfa_chain() (miniwfa.c:776-822) anchors the
// two sequences on shared 13-mers and closes the stretches between the anchors one by one.  The plan (anchors, literal ops, list
// of stretches; wfachain.c) is integer work on two sequences and runs on the host; the stretches -- all of the DP -- go through the
// ladder again as ordinary problems (sub-ranges of the same device sequences, no cell cap: miniwfa.c:831), and the stitched
// CIGAR replaces the result of the problem.  Rare by construction (a gap of tens of kilobases at > 10 % divergence).
static int wfs_fallback(mga_sctx_t *sc, int n_fb, const int32_t *d_fb, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
						mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used)
{
	struct job_t { int32_t pi; mga_wfa_prob_t pb; mga_wfa_res_t r0; mga_wc_plan_t plan; int64_t sub0; };
	if (sc->wfa_uncapped) { mga_set_error("WFA fallback: nested cell cap"); return -1; } // cannot happen: the nested ladder has no cap
	std::vector<int32_t> ids(n_fb);
	std::vector<job_t> job(n_fb);
	std::vector<mga_wfa_prob_t> sub;
	std::vector<char> tbuf, qbuf;
	mga_wc_par_t par;
	mga_wc_par_default(&par);
	int ret = -1;
	if (mga_ssync(sc) < 0 || mga_d2h(ids.data(), d_fb, (size_t)n_fb * 4) < 0) return -1;
	std::sort(ids.begin(), ids.end()); // the kernels append in completion order; the result does not depend on it, the pool layout would
	for (int j = 0; j < n_fb; ++j) memset(&job[j].plan, 0, sizeof(mga_wc_plan_t));
	for (int j = 0; j < n_fb; ++j) {
		job_t &J = job[j];
		J.pi = ids[j], J.sub0 = (int64_t)sub.size();
		if (mga_d2h(&J.pb, d_prob + J.pi, sizeof(mga_wfa_prob_t)) < 0 || mga_d2h(&J.r0, d_res + J.pi, sizeof(mga_wfa_res_t)) < 0) goto done;
		tbuf.resize((size_t)J.pb.tl + 1); qbuf.resize((size_t)J.pb.ql + 1);
		if (mga_d2h(tbuf.data(), d_tseq + J.pb.t_off, (size_t)J.pb.tl) < 0 || mga_d2h(qbuf.data(), d_qseq + J.pb.q_off, (size_t)J.pb.ql) < 0) goto done;
		if (mga_wfa_chain_plan(&par, J.pb.tl, tbuf.data(), J.pb.ql, qbuf.data(), &J.plan) < 0) { mga_set_error("WFA fallback: out of memory"); goto done; }
		for (int32_t i = 0; i < J.plan.n; ++i) {
			const mga_wc_el_t &e = J.plan.el[i];
			if (e.sub) sub.push_back(mga_wfa_prob_t{ J.pb.t_off + e.x0, J.pb.q_off + e.y0, e.tl, e.ql });
		}
	}
	{
		const int n_sub = (int)sub.size();
		std::vector<mga_wfa_res_t> sres((size_t)n_sub);
		if (n_sub > 0) {
			if (mga_dbuf_reserve(&sc->fb_prob, (size_t)n_sub * sizeof(mga_wfa_prob_t)) < 0 || mga_dbuf_reserve(&sc->fb_res, (size_t)n_sub * sizeof(mga_wfa_res_t)) < 0) goto done;
			if (mga_h2d(sc->fb_prob.p, sub.data(), (size_t)n_sub * sizeof(mga_wfa_prob_t)) < 0) goto done;
			sc->wfa_uncapped = 1;
			const int rc = mga_dev_wfa_solve(sc, n_sub, (const mga_wfa_prob_t*)sc->fb_prob.p, d_tseq, d_qseq, (mga_wfa_res_t*)sc->fb_res.p, d_pool, pool_cap, d_pool_used, 0, 0, 0);
			sc->wfa_uncapped = 0;
			if (rc < 0 || mga_ssync(sc) < 0 || mga_d2h(sres.data(), sc->fb_res.p, (size_t)n_sub * sizeof(mga_wfa_res_t)) < 0) goto done;
		}
		unsigned long long used = 0;
		if (mga_d2h(&used, d_pool_used, 8) < 0) goto done;
		std::vector<uint32_t> cig, out;
		std::vector<const uint32_t*> cptr;
		std::vector<int32_t> cn;
		for (int j = 0; j < n_fb; ++j) {
			job_t &J = job[j];
			const int64_t s1 = j + 1 < n_fb ? job[j + 1].sub0 : (int64_t)n_sub;
			int64_t tot = 0, iter = J.r0.n_iter, score = J.plan.score;
			for (int64_t k = J.sub0; k < s1; ++k) tot += sres[k].n_cigar;
			cig.resize((size_t)tot + 1); cptr.clear(); cn.clear();
			tot = 0;
			for (int64_t k = J.sub0; k < s1; ++k) {
				if (sres[k].status != MGA_WFA_OK) { mga_set_error("WFA fallback: sub-problem failed (status %d)", sres[k].status); goto done; }
				if (sres[k].n_cigar > 0 && mga_d2h(cig.data() + tot, d_pool + sres[k].cig_off, (size_t)sres[k].n_cigar * 4) < 0) goto done;
				cptr.push_back(cig.data() + tot); cn.push_back(sres[k].n_cigar);
				tot += sres[k].n_cigar, iter += sres[k].n_iter, score += sres[k].score;
			}
			const int64_t cap = tot + J.plan.n + 1;
			out.resize((size_t)cap);
			const int64_t n_out = mga_wfa_chain_stitch(&J.plan, cptr.data(), cn.data(), out.data(), cap);
			if (n_out < 0) { mga_set_error("WFA fallback: stitch overflow"); goto done; }
			if ((int64_t)used + n_out > pool_cap) { mga_set_error("WFA fallback: CIGAR pool of %ld ops exhausted", (long)pool_cap); goto done; }
			mga_wfa_res_t r;
			r.score = (int32_t)score, r.n_cigar = (int32_t)n_out, r.cig_off = (int64_t)used, r.status = MGA_WFA_OK, r.pad = 0, r.n_iter = iter;
			if (n_out > 0 && mga_h2d(d_pool + used, out.data(), (size_t)n_out * 4) < 0) goto done;
			if (mga_h2d(d_res + J.pi, &r, sizeof(r)) < 0) goto done;
			used += (unsigned long long)n_out;
		}
		if (mga_h2d(d_pool_used, &used, 8) < 0) goto done;
	}
	ret = 0;
done:
	for (int j = 0; j < n_fb; ++j) mga_wfa_chain_plan_free(&job[j].plan);
	return ret;
}

// problems launched / given up per rung since the last reset (bench.py, DESIGN.md: where the gaps are decided)
static long long g_rung_n[MGA_WFA_MAX_TIER], g_rung_up[MGA_WFA_MAX_TIER];
extern "C" void mga_wfa_ladder_stats(int64_t *launched, int64_t *given_up, int reset)
{
	for (int t = 0; t < MGA_WFA_MAX_TIER; ++t) {
		launched[t] = __atomic_load_n(&g_rung_n[t], __ATOMIC_RELAXED), given_up[t] = __atomic_load_n(&g_rung_up[t], __ATOMIC_RELAXED);
		if (reset) __atomic_store_n(&g_rung_n[t], 0, __ATOMIC_RELAXED), __atomic_store_n(&g_rung_up[t], 0, __ATOMIC_RELAXED);
	}
}

// One SWEEP of the ladder (round 3): the rungs run ONCE each, in ascending order, on the context's stream.  A problem that gives up on rung t is appended to
// the list of a HIGHER rung, which has not started yet and reads its list's length from device memory when it does -- so every retry is served in the same
// sweep, with no host round trip between the rungs (round 2 ran pass after pass, every rung again for the problems that had climbed to it: seven host
// synchronisations and ~3 launches per rung and chunk, [measured] 79 ms per 125k reads in the 512-diagonal tier alone, most of it launch tails).
// A rung's list has room for its own problems + a share of what the rungs below it run; an overflow (never seen on real reads) only moves the count on, and
// the host then sweeps once more over what is still open.
static int wfs_sweep(mga_sctx_t *sc, const wfs_ladder_t *LD, int arr_pct, int n_list, const int32_t *d_ids /* NULL: problems 0 .. n_list - 1 */,
					 const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq, mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap,
					 unsigned long long *d_pool_used, int *ctl, bool *any_tb, int *n_open)
{
	constexpr int NS = MGA_WFA_N_SLOT;
	constexpr int O_TOFF = WFS_NBIN, O_RC = O_TOFF + 16, O_ERR = O_RC + 32;
	hipStream_t st = (hipStream_t)sc->stream;
	const int NR = LD->n;
	static int dbg = -1;
	if (dbg < 0) { const char *e = getenv("MGA_DEBUG_WFA"); dbg = e && atoi(e) > 0; }
	*n_open = 0;
	// MGA_WFA_LIST_STABLE=<n>: the lists of the first n windowed rungs in problem order (k_wfa_bin_scatter); 0 (default) = every rung longest-first.  [measured, profiles/r05r_list_sweep.txt]
	// n = 3: W16 12.7 -> 12.0, W32 15.1 -> 15.5, the traceback walk 18.0 -> 17.8-18.0 ms per 125 000 reads: the walk is a chain of dependent loads per lane, not their volume.  Only the new ladder's
	// narrow rungs qualify (the old ladder's rungs are multi-wave kernels whose launches do end in their longest problems)
	static int n_stable_env = -1;
	if (n_stable_env < 0) { const char *e = getenv("MGA_WFA_LIST_STABLE"); n_stable_env = e && *e ? atoi(e) : 0; if (n_stable_env > 3) n_stable_env = 3; if (n_stable_env < 0) n_stable_env = 0; }
	const int n_stable = LD->r[0].kind != 0 ? 0 : n_stable_env; // (kind 1: the round-2 ladder, register tiers from the first rung on)
	MGA_HIP_CHECK(hipMemsetAsync(ctl, 0, (size_t)O_ERR * 4, st)); // histogram, offsets, list lengths (the err / fb / cells words behind them are kept)
	{
		int nb = (n_list + 1023) / 1024;
		if (nb > 1024) nb = 1024;
		wfs_thr_t T;
		T.n = NR, T.n_stable = n_stable;
		for (int k = 0; k < NS; ++k) T.t[k] = k < NR ? LD->r[k].maxlen : 0x7fffffff;
		mga_prof_begin(st, MGA_K_SCAN);
		hipLaunchKernelGGL(k_wfa_bin_count, dim3(nb), dim3(1024), 0, st, n_list, d_ids, d_prob, (int32_t*)sc->wfa_key.p, ctl, T);
		hipLaunchKernelGGL(k_wfa_bin_scan, dim3(1), dim3(1024), 0, st, ctl, ctl + O_TOFF);
		mga_prof_end(st, MGA_K_SCAN);
		MGA_HIP_CHECK(hipGetLastError());
	}
	int h[NS + 2], cap[NS], base[NS + 1];
	int *own = sc->wfa_own; // (lives in the context: the copy engine reads it after this frame's locals could be gone)
	if (mga_d2h_s(sc, h, ctl + O_TOFF, (NS + 1) * 4) < 0 || mga_ssync(sc) < 0) return -1;
	int64_t tot_cap = 0;
	{ // room of every rung's list: its own problems + arr_pct % of what the rungs feeding it may run (+ 4096)
		int64_t feed[NS];
		for (int t = 0; t < NS; ++t) own[t] = t < NR ? h[t + 1] - h[t] : 0, feed[t] = 0;
		for (int t = 0; t < NR; ++t) {
			int64_t c = own[t] + (feed[t] > 0 ? feed[t] * arr_pct / 100 + (arr_pct < 100 ? 4096 : 0) : 0);
			if (c > n_list) c = n_list;
			cap[t] = (int)c;
			if (LD->r[t].next >= 0) feed[LD->r[t].next] += c;
		}
		for (int t = NR; t < NS; ++t) cap[t] = 0;
		base[0] = 0;
		for (int t = 0; t < NS; ++t) base[t + 1] = base[t] + cap[t], tot_cap = base[t + 1];
	}
	if (mga_dbuf_reserve(&sc->wfa_list[0], (size_t)tot_cap * 4 + 64) < 0) return -1;
	int32_t *L = (int32_t*)sc->wfa_list[0].p;
	{
		wfs_shift_t sh;
		for (int t = 0; t < NS; ++t) sh.s[t] = t < NR ? base[t] - h[t] : 0;
		mga_prof_begin(st, MGA_K_SCAN);
		hipLaunchKernelGGL(k_wfa_bin_scatter, dim3((n_list + 8191) / 8192), dim3(1024), 0, st, n_list, d_ids, (const int32_t*)sc->wfa_key.p, ctl, L, sh, n_stable);
		mga_prof_end(st, MGA_K_SCAN);
		MGA_HIP_CHECK(hipGetLastError());
	}
	int *rc = ctl + O_RC; // rc[t]: length of rung t's list (own problems, then arrivals); rc[15]: problems beyond the last rung
	MGA_HIP_CHECK(hipMemcpyAsync(rc, own, NS * 4, hipMemcpyHostToDevice, st));
	MGA_HIP_CHECK(hipMemsetAsync(sc->wfa_cnt.p, 0, 4096, st)); // the launches' work-queue counters (one 64-byte line each)
	int64_t tb_bytes = 0; // traceback regions: ONE buffer, the largest windowed rung's -- every rung's walk (k_wfa_tb) runs right behind its forward pass
	for (int t = 0; t < NR; ++t) if (LD->r[t].kind == 0 && cap[t] > 0 && (int64_t)cap[t] * mga_dev_wfa_win_tb_stride(LD->r[t].idx) > tb_bytes) tb_bytes = (int64_t)cap[t] * mga_dev_wfa_win_tb_stride(LD->r[t].idx);
	if (tb_bytes > 0 && mga_dbuf_reserve(&sc->wfa_tbuf[0], (size_t)tb_bytes + 256) < 0) return -1;
	(void)any_tb;
	// The register / HBM rungs hold few, long problems (hundreds to thousands of score steps each: a launch is as long as its longest chain of them, with most of
	// the GPU idle).  Their OWN problems depend on nothing, so they start NOW on a side stream, next to the windowed rungs that fill the machine; what ARRIVES
	// at them from below is run by a second, short launch at the end of the sweep ([measured] 68 ms of the 225 ms of an isolated WFA phase were these tails).
	const char *e_side = getenv("MGA_WFA_SIDE");
	const int use_side = !(e_side && atoi(e_side) == 0); // MGA_WFA_SIDE=0: everything on one stream (bench.py's isolated pass: per-kernel timings that do not overlap)
	hipStream_t side = use_side ? (hipStream_t)sc->tier_stream[0] : st;
	bool forked = !use_side;
	auto rt_of = [&](int t) {
		const int nx = LD->r[t].next;
		// (wfa_fb takes the problems that hit the cell cap: only the HBM tiers count cells)
		return mga_wfa_retry_t{ nx >= 0 ? L + base[nx] : L, rc + (nx >= 0 ? nx : 15), ctl + O_ERR, (int32_t*)sc->wfa_fb.p, ctl + O_ERR + 1, nx >= 0 ? cap[nx] : 0 };
	};
	for (int t = 0; t < NR; ++t) {
		if (LD->r[t].kind != 1 || own[t] <= 0) continue;
		if (!forked) {
			MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_ready, st));
			MGA_HIP_CHECK(hipStreamWaitEvent(side, (hipEvent_t)sc->ev_ready, 0));
			forked = true;
		}
		if (mga_dev_wfa_tier(sc, rc + t, own[t], 0, LD->r[t].idx, side, L + base[t], d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, LD->r[t].idx, rt_of(t)) < 0) return -1;
	}
	// Round 5: a rung's walk (k_wfa_tb: a lane per problem chasing traceback bytes -- bound by memory latency, 2 % of the vector issue slots) runs on a stream of its own
	// NEXT TO the following rung's forward pass (bound by vector issue): two traceback buffers used in turn, the forward pass of rung k waits for the walk of rung
	// k - 2 (the previous user of its buffer), the walk of rung k for its forward pass.  Opt-in (MGA_WFA_TB_SIDE=1): measured, no gain -- see below; the default and the isolated pass (MGA_WFA_SIDE=0) use one buffer, one stream.
	const char *e_tbs = getenv("MGA_WFA_TB_SIDE");
	const bool tb_side = use_side && e_tbs && atoi(e_tbs) > 0; // [measured, round 5, bench workload, three interleaved repetitions] 3.51 / 3.60 / 3.36 Gbp/s with the walk on its own stream against 3.61 / 3.63 without: kept behind MGA_WFA_TB_SIDE=1, not the default
	hipStream_t tbs = tb_side ? (hipStream_t)sc->tier_stream[1] : st;
	if (tb_side && tb_bytes > 0 && mga_dbuf_reserve(&sc->wfa_tbuf[1], (size_t)tb_bytes + 256) < 0) return -1;
	int n_win = 0;
	for (int t = 0; t < NR; ++t) {
		if (LD->r[t].kind != 0 || cap[t] <= 0) continue;
		char *tbuf = (char*)sc->wfa_tbuf[tb_side ? (n_win & 1) : 0].p;
		if (tb_side && n_win >= 2) MGA_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)sc->ev_done[2 + (n_win & 1)], 0)); // the walk that read this buffer two rungs ago
		if (mga_dev_wfa_win(sc, rc + t, cap[t], L + base[t], d_prob, d_tseq, d_qseq, d_res, tbuf, LD->r[t].idx, 9 + LD->r[t].idx, rt_of(t), d_pool, pool_cap, d_pool_used, ctl + O_ERR) < 0) return -1;
		if (tb_side) {
			MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_done[1], st));
			MGA_HIP_CHECK(hipStreamWaitEvent(tbs, (hipEvent_t)sc->ev_done[1], 0));
		}
		if (mga_dev_wfa_traceback(sc, tbs, rc + t, cap[t], L + base[t], d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, ctl + O_ERR, LD->r[t].idx) < 0) return -1;
		if (tb_side) MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_done[2 + (n_win & 1)], tbs));
		++n_win;
	}
	if (tb_side) for (int k = 0; k < 2 && k < n_win; ++k) MGA_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)sc->ev_done[2 + k], 0)); // every walk is through before what follows on the main stream
	if (forked && use_side) {
		MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_done[0], side));
		MGA_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)sc->ev_done[0], 0));
	}
	for (int t = 0; t < NR; ++t) { // arrivals at the register / HBM rungs, in ascending order (a rung's own launch may itself have sent problems up)
		if (LD->r[t].kind != 1 || cap[t] - own[t] <= 0) continue;
		if (mga_dev_wfa_tier(sc, rc + t, cap[t], own[t], 32 + LD->r[t].idx, 0, L + base[t], d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, LD->r[t].idx, rt_of(t)) < 0) return -1;
	}
	int hr[16], herr = 0;
	if (mga_d2h_s(sc, hr, rc, 16 * 4) < 0 || mga_d2h_s(sc, &herr, ctl + O_ERR, 4) < 0 || mga_ssync(sc) < 0) return -1;
	if (dbg) {
		fprintf(stderr, "[wfa] sweep over %d problems:", n_list);
		for (int t = 0; t < NR; ++t) if (hr[t]) fprintf(stderr, " %s%d: %d (+%d, room %d)", LD->r[t].kind == 0 ? "W" : "R", LD->r[t].idx, own[t], hr[t] - own[t], cap[t]);
		fprintf(stderr, "\n");
	}
	if (herr) { mga_set_error("WFA: %d problems failed (CIGAR pool of %ld ops exhausted, iteration cap, or a traceback walked out of its window)", herr, (long)pool_cap); return -1; }
	if (hr[15] > 0) { mga_set_error("%d WFA problems exceed the largest capacity tier", hr[15]); return -1; }
	if (!sc->wfa_uncapped) for (int t = 0; t < NR; ++t) {
		__atomic_fetch_add(&g_rung_n[t], (long long)(hr[t] < cap[t] ? hr[t] : cap[t]), __ATOMIC_RELAXED);
		__atomic_fetch_add(&g_rung_up[t], (long long)(hr[t] - own[t]), __ATOMIC_RELAXED); // (arrivals AT rung t)
	}
	for (int t = 0; t < NR; ++t) if (hr[t] > cap[t]) *n_open += hr[t] - cap[t];
	return 0;
}

extern "C" int mga_dev_wfa_solve(mga_sctx_t *sc, int n, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
								 mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int64_t *cells,
								 void (*bulk_done)(void*), void *bulk_arg)
{
	(void)bulk_done; (void)bulk_arg; // (round 2: early release of the caller's GPU-phase token between concurrent tiers; a sweep is one chain of launches)
	if (cells) *cells = 0;
	if (n <= 0) return 0;
	hipStream_t st = (hipStream_t)sc->stream;
	const wfs_ladder_t *LD = wfs_ladder(sc->wfa_uncapped); // (the fallback's sub-problems: register / HBM tiers only, results in the pool at once)
	// ctl: hist[NBIN] | tier_off[16] | rc[2][16] | err | fb | cells (8 bytes)
	constexpr int O_TOFF = WFS_NBIN, O_RC = O_TOFF + 16, O_ERR = O_RC + 32, O_FB = O_ERR + 1, O_CELLS = O_ERR + 2, N_CTL = O_CELLS + 2;
	if (mga_dbuf_reserve(&sc->wfa_key, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&sc->wfa_fb, (size_t)n * 4 + 64) < 0 || mga_dbuf_reserve(&sc->wfa_ctl, (size_t)N_CTL * 4) < 0 ||
		mga_dbuf_reserve(&sc->wfa_cnt, 4096) < 0) return -1;
	int *ctl = (int*)sc->wfa_ctl.p;
	MGA_HIP_CHECK(hipMemsetAsync(ctl, 0, (size_t)N_CTL * 4, st));
	hipLaunchKernelGGL(k_wfa_mark_pending, dim3((n + 255) / 256), dim3(256), 0, st, n, d_res);
	MGA_HIP_CHECK(hipGetLastError());
	bool any_tb = false;
	int n_open = 0;
	static int arr_pct = -1;
	if (arr_pct < 0) { const char *e = getenv("MGA_WFA_ARRIVALS_PCT"); arr_pct = e ? atoi(e) : 20; } // (tests: 0 leaves a rung's list room for 4096 arrivals only)
	if (wfs_sweep(sc, LD, arr_pct, n, 0, d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, ctl, &any_tb, &n_open) < 0) return -1;
	if (n_open > 0) { // a rung's list overflowed (far more of a chunk's gaps gave up than any read set has shown): what is still open is swept again in slices whose
		// lists have room for EVERYTHING below them (no second overflow), small enough for the traceback regions that implies
		const int SLICE = 32768;
		int *d_cnt = ctl + O_CELLS, k = 0; // (the cells words are free until the end)
		if (mga_dbuf_reserve(&sc->wfa_list[1], (size_t)n * 4 + 64) < 0) return -1;
		MGA_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, st));
		hipLaunchKernelGGL(k_wfa_collect_open, dim3((n + 255) / 256), dim3(256), 0, st, n, (const mga_wfa_res_t*)d_res, (int32_t*)sc->wfa_list[1].p, d_cnt);
		MGA_HIP_CHECK(hipGetLastError());
		if (mga_d2h_s(sc, &k, d_cnt, 4) < 0 || mga_ssync(sc) < 0) return -1;
		MGA_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, st));
		for (int o = 0; o < k; o += SLICE) {
			if (wfs_sweep(sc, LD, 100, k - o < SLICE ? k - o : SLICE, (const int32_t*)sc->wfa_list[1].p + o, d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used, ctl, &any_tb, &n_open) < 0) return -1;
			if (n_open > 0) { mga_set_error("WFA ladder: %d problems open after a sweep with room for all", n_open); return -1; }
		}
	}
	{ // problems the exact pass gave up on (> 1e8 cells): miniwfa's chained fallback (miniwfa.c:829-832)
		int n_fb = 0;
		if (mga_d2h_s(sc, &n_fb, ctl + O_FB, 4) < 0 || mga_ssync(sc) < 0) return -1;
		if (n_fb > 0 && wfs_fallback(sc, n_fb, (const int32_t*)sc->wfa_fb.p, d_prob, d_tseq, d_qseq, d_res, d_pool, pool_cap, d_pool_used) < 0) return -1;
		ctl = (int*)sc->wfa_ctl.p; // the nested ladder may have grown the buffer
	}
	if (cells) {
		unsigned long long c = 0;
		int nb = (n + 255) / 256;
		if (nb > 2048) nb = 2048;
		hipLaunchKernelGGL(k_wfa_sum_cells, dim3(nb), dim3(256), 0, st, n, (const mga_wfa_res_t*)d_res, (unsigned long long*)(ctl + O_CELLS));
		MGA_HIP_CHECK(hipGetLastError());
		if (mga_d2h_s(sc, &c, ctl + O_CELLS, 8) < 0 || mga_ssync(sc) < 0) return -1;
		*cells = (int64_t)c;
	}
	return 0;
}
