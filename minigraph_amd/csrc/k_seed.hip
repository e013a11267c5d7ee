// k_seed.hip -- seed collection for a batch of reads, one wavefront per read.
//
// Replaces collect_matches() + collect_seed_hits() (reference map-algo.c:58-91,152-192) and the
// mg_idx_get() probes under them (index.c:50-72):
//   pass 1 (k_seed_count)  every lane probes the flat minimizer table for one minimizer (one 16-byte
//          load per probe step); per read: anchor count, kept-minimizer count and rep_len, the union
//          length of query intervals covered by minimizers with occ >= max_occ (map-algo.c:72-79,88),
//          computed from wave ballots instead of the sequential rep_st/rep_en update.
//   pass 2 (k_seed_fill)   hits are expanded to anchors in the reference's order (minimizer order,
//          then ascending graph position), mini_pos[] is written, and the read's anchors are sorted
//          by x with the exact klib permutation (dev_klibsort.h).
// Anchor encoding (map-algo.c:177-186):
//   x = seg<<33 | rev<<32 | rpos      (reverse strand: rpos = seglen - (lastPos + 1 - span) - 1)
//   y = min(occ,255)<<56 | tandem<<42 | span<<32 | qpos      (query segment id 0: single-segment reads)
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_klibsort.h"
#include "mga_idxhash.h"

// probe: occurrence count and slot value of minimizer hash `key`
__device__ __forceinline__ int32_t seed_probe(const mga_didx_t &ix, uint64_t key, uint64_t *val)
{
	uint64_t sl = mga_idx_slot(key, ix.bits);
	for (;;) {
		const mg128_t e = ix.d_tab[sl];
		if (e.x == MGA_IDX_EMPTY) { *val = 0; return 0; }
		if ((e.x & ~MGA_IDX_LIST) == key) {
			*val = e.y;
			return (e.x & MGA_IDX_LIST) ? (int32_t)(uint32_t)e.y : 1;
		}
		sl = (sl + 1) & (ix.n_slots - 1);
	}
}

__global__ void __launch_bounds__(64) k_seed_count(mga_didx_t ix, int n, const mg128_t *__restrict__ mz, const int64_t *__restrict__ mz_off, const int32_t *__restrict__ mz_cnt, int max_occ,
												   int32_t *__restrict__ occ, uint64_t *__restrict__ val,
												   int32_t *__restrict__ d_na, int32_t *__restrict__ d_nmini, int32_t *__restrict__ d_rep)
{
	const int r = blockIdx.x, lane = threadIdx.x;
	if (r >= n) return;
	const int64_t base = mz_off[r];
	const int32_t n_mz = mz_cnt ? mz_cnt[r] : (int32_t)(mz_off[r + 1] - base); // (mz_cnt: the slots of a read are only partly filled)
	int32_t na = 0, nmini = 0, rep_len = 0, en_prev = 0;
	for (int32_t c0 = 0; c0 < n_mz; c0 += 64) {
		const int32_t i = c0 + lane;
		const bool act = i < n_mz;
		int32_t t = 0, en = 0, st = 0;
		if (act) {
			const mg128_t m = mz[base + i];
			uint64_t v;
			t = seed_probe(ix, m.x >> 8, &v);
			occ[base + i] = t, val[base + i] = v;
			en = (int32_t)((uint32_t)m.y >> 1) + 1, st = en - (int32_t)(m.x & 0xff);
		}
		const bool rep = act && t >= max_occ;
		const uint64_t m_rep = __ballot(rep);
		// previous repetitive minimizer's end: nearest repetitive lane below, else the carry
		const uint64_t below = m_rep & mga_lanemask_lt();
		const int src = below ? 63 - __clzll(below) : 0;
		int32_t ep = __shfl(en, src);
		if (!below) ep = en_prev;
		int32_t contrib = 0;
		if (rep) contrib = st > ep ? en - st : en - ep;
		int32_t kept_t = (act && !rep) ? t : 0;
		for (int d = 32; d > 0; d >>= 1) { contrib += __shfl_xor(contrib, d); kept_t += __shfl_xor(kept_t, d); }
		rep_len += contrib, na += kept_t;
		nmini += __popcll(__ballot(act && !rep));
		if (m_rep) en_prev = __shfl(en, 63 - __clzll(m_rep));
	}
	if (lane == 0) { d_na[r] = na; d_nmini[r] = nmini; d_rep[r] = rep_len; }
}

__global__ void __launch_bounds__(64) k_seed_fill(mga_didx_t ix, int n, const mg128_t *__restrict__ mz, const int64_t *__restrict__ mz_off, const int32_t *__restrict__ mz_cnt, int max_occ,
												  const int32_t *__restrict__ occ, const uint64_t *__restrict__ val,
												  const int64_t *__restrict__ a_off, mg128_t *__restrict__ a_all,
												  const int64_t *__restrict__ mini_off, int32_t *__restrict__ mini_all, mg128_t *__restrict__ tmp_all)
{
	__shared__ klib_lds_t L;
	const int r = blockIdx.x, lane = threadIdx.x;
	if (r >= n) return;
	const int64_t base = mz_off[r];
	const int32_t n_mz = mz_cnt ? mz_cnt[r] : (int32_t)(mz_off[r + 1] - base); // (mz_cnt: the slots of a read are only partly filled)
	mg128_t *a = a_all + a_off[r];
	int32_t *mini = mini_all + mini_off[r];
	const int64_t n_a = a_off[r + 1] - a_off[r];
	int32_t na = 0, nmini = 0;
	for (int32_t c0 = 0; c0 < n_mz; c0 += 64) {
		const int32_t i = c0 + lane;
		const bool act = i < n_mz;
		int32_t t = 0;
		mg128_t m; m.x = m.y = 0;
		if (act) { m = mz[base + i]; t = occ[base + i]; }
		const bool kept = act && t < max_occ;
		const uint64_t m_kept = __ballot(kept);
		const int32_t tk = kept ? t : 0;
		const int32_t incl = mga_wave_incl_scan_i32(tk);
		if (kept) {
			const uint64_t key = m.x >> 8;
			const uint32_t q_pos = (uint32_t)m.y, q_span = (uint32_t)(m.x & 0xff);
			mini[nmini + __popcll(m_kept & mga_lanemask_lt())] = (int32_t)(q_pos >> 1);
			if (t > 0) {
				bool tandem = false;
				if (i > 0 && mz[base + i - 1].x >> 8 == key) tandem = true;
				if (i < n_mz - 1 && mz[base + i + 1].x >> 8 == key) tandem = true;
				uint64_t y = (uint64_t)q_span << 32 | (uint64_t)(q_pos >> 1);
				if (tandem) y |= MG_SEED_TANDEM;
				y |= (uint64_t)(t < 255 ? t : 255) << MG_SEED_OCC_SHIFT;
				const uint64_t v = val[base + i];
				const uint64_t *cr = ix.d_pos + (v >> 32);
				mg128_t *o = a + na + (incl - tk);
				for (int32_t k = 0; k < t; ++k) {
					const uint64_t rr = t == 1 ? v : cr[k];
					const uint64_t seg = rr >> 32;
					const int32_t rpos = (int32_t)((uint32_t)rr >> 1);
					uint64_t x;
					if ((rr & 1) == (q_pos & 1)) x = seg << 33 | (uint64_t)(uint32_t)rpos;
					else x = seg << 33 | 1ULL << 32 | (uint64_t)(uint32_t)(ix.d_seg_len[seg] - (rpos + 1 - (int32_t)q_span) - 1);
					o[k].x = x, o[k].y = y;
				}
			}
		}
		na += __shfl(incl, 63);
		nmini += __popcll(m_kept);
	}
	__syncthreads();
	klib_sort128x(a, n_a, (int32_t*)(tmp_all + a_off[r]), &L);
}

extern "C" int mga_dev_seed_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, const int32_t *d_mz_cnt, int max_occ,
								  int32_t *d_occ, uint64_t *d_val, int32_t *d_na, int32_t *d_nmini, int32_t *d_rep_len)
{
	if (n <= 0) return 0;
	mga_prof_begin(sc->stream, MGA_K_SEED_COUNT);
	hipLaunchKernelGGL(k_seed_count, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, *ix, n, d_mz, d_mz_off, d_mz_cnt, max_occ, d_occ, d_val, d_na, d_nmini, d_rep_len);
	mga_prof_end(sc->stream, MGA_K_SEED_COUNT);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int mga_dev_seed_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, const int32_t *d_mz_cnt, int max_occ,
								 const int32_t *d_occ, const uint64_t *d_val, const int64_t *d_a_off, mg128_t *d_a,
								 const int64_t *d_mini_off, int32_t *d_mini, mg128_t *d_tmp)
{
	if (n <= 0) return 0;
	mga_prof_begin(sc->stream, MGA_K_SEED_FILL);
	hipLaunchKernelGGL(k_seed_fill, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, *ix, n, d_mz, d_mz_off, d_mz_cnt, max_occ, d_occ, d_val, d_a_off, d_a, d_mini_off, d_mini, d_tmp);
	mga_prof_end(sc->stream, MGA_K_SEED_FILL);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
