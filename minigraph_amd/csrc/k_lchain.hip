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
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_klibsort.h"

#define LC_NONE INT32_MIN

__device__ __forceinline__ float lc_log2(float x) // mgpriv.h:63-71
{
	uint32_t i = __float_as_uint(x);
	float r = (float)((int32_t)(i >> 23 & 255) - 128);
	i &= ~(255U << 23);
	i += 127U << 23;
	const float f = __uint_as_float(i);
	r += (-0.34484843f * f + 2.02466578f) * f - 0.67487759f;
	return r;
}

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

__global__ void __launch_bounds__(64) k_lchain(int n_reads, const mg128_t *__restrict__ a_all, const int64_t *__restrict__ a_off, mga_lchain_par_t P,
											   uint64_t *__restrict__ u_all, mg128_t *__restrict__ b_all, int32_t *__restrict__ d_nu, int32_t *__restrict__ d_nb,
											   int32_t *__restrict__ ws_i32, mg128_t *__restrict__ ws_z)
{
	__shared__ klib_lds_t L;
	const int r = blockIdx.x, lane = threadIdx.x;
	if (r >= n_reads) return;
	const int64_t off = a_off[r];
	const int32_t n = (int32_t)(a_off[r + 1] - off);
	if (n == 0) { if (lane == 0) d_nu[r] = 0, d_nb[r] = 0; return; }
	const mg128_t *a = a_all + off;
	int32_t *f = ws_i32 + off * 4, *p = f + n, *v = p + n, *t = v + n; // 4 int32 per anchor
	mg128_t *z = ws_z + off;                                             // 1 mg128 per anchor
	uint64_t *u = u_all + off;
	mg128_t *b = b_all + off;
	if (P.max_dist_x < P.bw) P.max_dist_x = P.bw;
	if (P.max_dist_y < P.bw) P.max_dist_y = P.bw;
	const int32_t max_drop = P.bw;

	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	__syncthreads();

	// ---------------- DP ----------------
	int32_t st = 0, max_ii = -1;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t xi = a[i].x, yi = a[i].y;
		while (st < i) { // lchain.c:171
			const uint64_t xs = a[st].x;
			if (xi >> 32 != xs >> 32 || xi > xs + (uint64_t)(int64_t)P.max_dist_x) ++st; else break;
		}
		if (i - st > P.max_iter) st = i - P.max_iter;
		int32_t max_f = (int32_t)(yi >> 32 & 0xff), max_j = -1, n_skip = 0, end_j = st - 1;
		bool cut = false;
		for (int32_t j0 = i - 1; j0 >= st && !cut; j0 -= 64) {
			const int32_t j = j0 - lane;
			const bool act = j >= st;
			int32_t sc = LC_NONE, pj = -1;
			if (act) {
				const mg128_t aj = a[j];
				sc = lc_score(xi, yi, aj.x, aj.y, P);
				if (sc != LC_NONE) { sc += f[j]; pj = p[j]; }
			}
			const bool valid = sc != LC_NONE;
			if (valid && pj >= 0) t[pj] = i; // lchain.c:188 (harmless beyond the cut: only compared against this i)
			__syncthreads();
			const bool hit_t = valid && t[j] == i;
			// exclusive prefix max of valid scores in visiting order, seeded with the running max_f
			int32_t pm = valid ? sc : INT32_MIN;
			for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(pm, d); if (lane >= d && y > pm) pm = y; }
			int32_t ex = __shfl_up(pm, 1);
			if (lane == 0) ex = INT32_MIN;
			if (ex < max_f) ex = max_f;
			const bool improve = valid && sc > ex;
			const uint64_t m_imp = __ballot(improve), m_hit = __ballot(hit_t && !improve);
			// replay n_skip over the event lanes (scalar, uniform)
			uint64_t ev = m_imp | m_hit;
			int cut_lane = 64;
			while (ev) {
				const int l = __builtin_ctzll(ev);
				ev &= ev - 1;
				if (m_imp >> l & 1) { if (n_skip > 0) --n_skip; }
				else if (++n_skip > P.max_skip) { cut_lane = l; break; }
			}
			const uint64_t before = cut_lane == 64 ? ~0ULL : (1ULL << cut_lane) - 1ULL;
			const uint64_t imp_b = m_imp & before;
			if (imp_b) {
				const int bl = 63 - __clzll(imp_b);
				max_f = __shfl(sc, bl), max_j = j0 - bl;
			}
			if (cut_lane < 64) { cut = true; end_j = j0 - cut_lane; }
			__syncthreads();
		}
		// lchain.c:191-196: best-scoring anchor within reach, recomputed when it fell out of range
		if (max_ii < 0 || xi - a[max_ii].x > (uint64_t)(int64_t)P.max_dist_x) {
			int32_t bf = INT32_MIN, bj = -1;
			for (int32_t j = i - 1 - lane; j >= st; j -= 64) { const int32_t fj = f[j]; if (bf < fj) bf = fj, bj = j; } // descending j per lane: first max kept
			for (int d = 32; d > 0; d >>= 1) {
				const int32_t of = __shfl_xor(bf, d), oj = __shfl_xor(bj, d);
				if (of > bf || (of == bf && oj > bj)) bf = of, bj = oj; // ties: the larger j was met first
			}
			max_ii = bj;
		}
		if (max_ii >= 0 && max_ii < end_j) { // lchain.c:197-201
			const mg128_t am = a[max_ii];
			const int32_t tmp = lc_score(xi, yi, am.x, am.y, P);
			if (tmp != LC_NONE && max_f < tmp + f[max_ii]) max_f = tmp + f[max_ii], max_j = max_ii;
		}
		int32_t vi = max_f;
		if (max_j >= 0) { const int32_t vj = v[max_j]; if (vj > max_f) vi = vj; }
		if (lane == 0) { f[i] = max_f; p[i] = max_j; v[i] = vi; }
		if (max_ii < 0 || (xi - a[max_ii].x <= (uint64_t)(int64_t)P.max_dist_x && f[max_ii] < max_f)) max_ii = i;
		__syncthreads();
	}

	// ---------------- backtrack (lchain.c:27-77) ----------------
	// z = chain ends with f >= min_sc, in index order, then the klib sort by score
	int32_t n_z = 0;
	for (int32_t c0 = 0; c0 < n; c0 += 64) {
		const int32_t i = c0 + lane;
		const bool ok = i < n && f[i] >= P.min_sc;
		const uint64_t m = __ballot(ok);
		if (ok) { mg128_t e; e.x = (uint64_t)(int64_t)f[i]; e.y = (uint64_t)i; z[n_z + __popcll(m & mga_lanemask_lt())] = e; }
		n_z += __popcll(m);
	}
	__syncthreads();
	if (n_z == 0) { if (lane == 0) d_nu[r] = 0, d_nb[r] = 0; return; }
	klib_sort128x(z, n_z, t, &L); // t[] is free here (re-zeroed below) and large enough for the range stack
	for (int32_t i = lane; i < n; i += 64) t[i] = 0;
	__syncthreads();
	int32_t n_u = 0, n_v = 0;
	if (lane == 0) { // the walk is a chain of dependent loads: one lane
		for (int32_t k = n_z - 1; k >= 0; --k) {
			const int32_t e = (int32_t)z[k].y, zs = (int32_t)z[k].x;
			if (t[e] != 0) continue;
			// mg_chain_bk_end (lchain.c:9-25)
			int32_t i = e, stop = -1, best_i = e, best = 0;
			do {
				t[i] = 2;
				stop = i = p[i];
				const int32_t s = i < 0 ? zs : zs - f[i];
				if (s > best) best = s, best_i = i;
				else if (best - s > max_drop) break;
			} while (i >= 0 && t[i] == 0);
			for (i = e; i >= 0 && i != stop; i = p[i]) t[i] = 0;
			const int32_t cutp = best_i, n_v0 = n_v;
			for (i = e; i != cutp; i = p[i]) v[n_v++] = i, t[i] = 1;
			const int32_t sc = i < 0 ? zs : zs - f[i];
			if (sc >= P.min_sc && n_v > n_v0 && n_v - n_v0 >= P.min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
			else n_v = n_v0;
		}
	}
	n_u = __shfl(n_u, 0), n_v = __shfl(n_v, 0);
	__syncthreads();
	if (n_u == 0) { if (lane == 0) d_nu[r] = 0, d_nb[r] = 0; return; }

	// ---------------- compact_a (lchain.c:79-112) ----------------
	// NB: v[] was overwritten from index 0 by the walk (n_v <= anchors visited), as in the reference
	// tmp anchors in chain order into z-space? z is still needed? no: reuse ws: write to b first (chain order), then reorder via z.
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
	klib_sort128x(w, n_u, t, &L);
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
	if (lane == 0) { d_nu[r] = n_u; d_nb[r] = n_v; }
}

extern "C" size_t mga_dev_lchain_ws_bytes(int64_t total_anchors) { return (size_t)(total_anchors + 16) * 32; }

extern "C" int mga_dev_lchain(mga_sctx_t *sc, int n, const mg128_t *d_a, const int64_t *d_a_off, const mga_lchain_par_t *par,
							  uint64_t *d_u, mg128_t *d_b, int32_t *d_nu, int32_t *d_nb, void *d_ws, size_t ws_bytes, int64_t total_anchors)
{
	if (n <= 0) return 0;
	if (par->min_cnt < 2) { mga_set_error("lchain: min_cnt >= 2 required by the workspace layout (got %d)", par->min_cnt); return -1; }
	if (ws_bytes < mga_dev_lchain_ws_bytes(total_anchors)) { mga_set_error("lchain: workspace too small"); return -1; }
	int32_t *ws_i32 = (int32_t*)d_ws;
	mg128_t *ws_z = (mg128_t*)((char*)d_ws + (size_t)(total_anchors + 8) * 16);
	mga_prof_begin(sc->stream, MGA_K_LCHAIN);
	hipLaunchKernelGGL(k_lchain, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_a, d_a_off, *par, d_u, d_b, d_nu, d_nb, ws_i32, ws_z);
	mga_prof_end(sc->stream, MGA_K_LCHAIN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
