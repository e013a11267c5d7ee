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
	__shared__ union { klib_lds_t big; klib_small_lds_t small; } L;
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
	// radix_sort_128x (map-algo.c:189).  Round 6: up to 1024 anchors -- most 10 kb reads -- the sort runs its sequential part in LDS (dev_klibsort.h: klib_sort128x_small)
	if (n_a <= KLIB_SMALL_CAP) klib_sort128x_small(a, (int32_t)n_a, tmp_all + a_off[r], &L.small);
	else klib_sort128x(a, n_a, (int32_t*)(tmp_all + a_off[r]), &L.big);
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

// ---- long queries (-x asm: contigs of megabases, a handful per batch) ----------------------------------------------
// One wavefront per read leaves the GPU idle when a read has 10^7 minimizers.  Here every minimizer of the batch is a work item:
//   k_seedl_probe   table probe per minimizer; kept occurrence count, kept flag, and rid<<32 | (repetitive ? end : 0)
//   three device scans over the whole (contiguous, exact-offset) minimizer array: the sums give every minimizer its first anchor
//                   slot and its mini_pos slot (and, sampled at the read boundaries, the per-read offsets); the running MAXIMUM of
//                   rid<<32|end gives every repetitive minimizer the end of the previous repetitive one of its read -- ends grow
//                   along a read -- which is all the sequential rep_st/rep_en update of collect_matches (map-algo.c:72-79,88) needs
//   k_seedl_finish  rep_len per read (a few atomics: repetitive minimizers are rare) and the per-read offsets
//   k_seedl_expand  one thread per minimizer writes its anchors.
// The anchors stay in hit order: under MG_M_RMQ the chainer runs on the host, which sorts them there with the reference's exact
// radix permutation (ksortx.c) before it chains.
#include <rocprim/device/device_scan.hpp>

__device__ __forceinline__ int seedl_read_of(const int64_t *__restrict__ mz_off, int n, int64_t i)
{ // last r with mz_off[r] <= i (empty reads share an offset with their successor)
	int lo = 0, hi = n - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (mz_off[mid] <= i) lo = mid; else hi = mid - 1; }
	return lo;
}

__global__ void __launch_bounds__(256) k_seedl_probe(mga_didx_t ix, int n, const mg128_t *__restrict__ mz, const int64_t *__restrict__ mz_off, int64_t n_mz, int max_occ,
													 int32_t *__restrict__ occ, uint64_t *__restrict__ val, int32_t *__restrict__ tk, int32_t *__restrict__ kf, uint64_t *__restrict__ rep_key)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mz; i += (int64_t)gridDim.x * blockDim.x) {
		const mg128_t m = mz[i];
		uint64_t v;
		const int32_t t = seed_probe(ix, m.x >> 8, &v);
		const bool kept = t < max_occ;
		occ[i] = t, val[i] = v;
		tk[i] = kept ? t : 0, kf[i] = kept ? 1 : 0;
		const uint32_t en = ((uint32_t)m.y >> 1) + 1;
		rep_key[i] = (uint64_t)seedl_read_of(mz_off, n, i) << 32 | (kept ? 0u : en);
	}
}

__global__ void __launch_bounds__(256) k_seedl_finish(int n, const mg128_t *__restrict__ mz, const int64_t *__restrict__ mz_off, int64_t n_mz, const int32_t *__restrict__ kf,
													  const uint64_t *__restrict__ rep_max, const int64_t *__restrict__ off_a, const int64_t *__restrict__ off_m,
													  int64_t *__restrict__ a_off, int64_t *__restrict__ mini_off, int32_t *__restrict__ rep_len)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mz; i += (int64_t)gridDim.x * blockDim.x) {
		if (i <= n) a_off[i] = off_a[mz_off[i]], mini_off[i] = off_m[mz_off[i]];
		if (kf[i]) continue;
		const mg128_t m = mz[i];
		const int32_t en = (int32_t)((uint32_t)m.y >> 1) + 1, st = en - (int32_t)(m.x & 0xff);
		const uint64_t me = rep_max[i]; // inclusive: its high half is this minimizer's read
		int32_t ep = 0;
		if (i > 0 && rep_max[i - 1] >> 32 == me >> 32) ep = (int32_t)(uint32_t)rep_max[i - 1];
		atomicAdd(&rep_len[me >> 32], st > ep ? en - st : en - ep);
	}
	if (n_mz <= n) // fewer minimizers than reads: the offsets above did not cover every read
		for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x)
			a_off[i] = off_a[mz_off[i]], mini_off[i] = off_m[mz_off[i]];
}

__global__ void __launch_bounds__(256) k_seedl_expand(mga_didx_t ix, int n, const mg128_t *__restrict__ mz, const int64_t *__restrict__ mz_off, int64_t n_mz, int max_occ,
													  const int32_t *__restrict__ occ, const uint64_t *__restrict__ val, const int64_t *__restrict__ off_a, const int64_t *__restrict__ off_m,
													  mg128_t *__restrict__ a_all, int32_t *__restrict__ mini_all)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mz; i += (int64_t)gridDim.x * blockDim.x) {
		const int32_t t = occ[i];
		if (t >= max_occ) continue;
		const mg128_t m = mz[i];
		const uint64_t key = m.x >> 8;
		const uint32_t q_pos = (uint32_t)m.y, q_span = (uint32_t)(m.x & 0xff);
		mini_all[off_m[i]] = (int32_t)(q_pos >> 1);
		if (t == 0) continue;
		const int r = seedl_read_of(mz_off, n, i);
		const int64_t rb = mz_off[r], re = mz_off[r + 1];
		bool tandem = false; // map-algo.c:168-172: the neighbouring minimizer of the same read has the same hash
		if (i > rb && mz[i - 1].x >> 8 == key) tandem = true;
		if (i < re - 1 && mz[i + 1].x >> 8 == key) tandem = true;
		uint64_t y = (uint64_t)q_span << 32 | (uint64_t)(q_pos >> 1);
		if (tandem) y |= MG_SEED_TANDEM;
		y |= (uint64_t)(t < 255 ? t : 255) << MG_SEED_OCC_SHIFT;
		const uint64_t v = val[i];
		const uint64_t *cr = ix.d_pos + (v >> 32);
		mg128_t *o = a_all + off_a[i];
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

static int seedl_blocks(int64_t n_mz) { int64_t nb = (n_mz + 255) / 256; return (int)(nb > 16384 ? 16384 : nb < 1 ? 1 : nb); }

extern "C" int mga_dev_seed_long_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, int64_t n_mz, int max_occ,
									   int32_t *d_occ, uint64_t *d_val, int32_t *d_tk, int32_t *d_kf, int64_t *d_off_a, int64_t *d_off_m, uint64_t *d_rep_key, uint64_t *d_rep_max,
									   int64_t *d_a_off, int64_t *d_mini_off, int32_t *d_rep_len)
{
	if (n <= 0) return 0;
	hipStream_t st = (hipStream_t)sc->stream;
	MGA_HIP_CHECK(hipMemsetAsync(d_rep_len, 0, (size_t)n * 4, st));
	if (n_mz <= 0) {
		MGA_HIP_CHECK(hipMemsetAsync(d_a_off, 0, (size_t)(n + 1) * 8, st));
		MGA_HIP_CHECK(hipMemsetAsync(d_mini_off, 0, (size_t)(n + 1) * 8, st));
		return 0;
	}
	const int nb = seedl_blocks(n_mz);
	mga_prof_begin(sc->stream, MGA_K_SEED_COUNT);
	hipLaunchKernelGGL(k_seedl_probe, dim3(nb), dim3(256), 0, st, *ix, n, d_mz, d_mz_off, n_mz, max_occ, d_occ, d_val, d_tk, d_kf, d_rep_key);
	mga_prof_end(sc->stream, MGA_K_SEED_COUNT);
	MGA_HIP_CHECK(hipGetLastError());
	if (mga_dev_scan_i32_to_i64(sc, d_tk, n_mz, d_off_a) < 0 || mga_dev_scan_i32_to_i64(sc, d_kf, n_mz, d_off_m) < 0) return -1;
	{
		size_t tmp_bytes = 0;
		MGA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tmp_bytes, d_rep_key, d_rep_max, (size_t)n_mz, rocprim::maximum<uint64_t>(), st));
		if (mga_dbuf_reserve(&sc->scan_tmp, tmp_bytes + 64) < 0) return -1;
		MGA_HIP_CHECK(rocprim::inclusive_scan(sc->scan_tmp.p, tmp_bytes, d_rep_key, d_rep_max, (size_t)n_mz, rocprim::maximum<uint64_t>(), st));
	}
	mga_prof_begin(sc->stream, MGA_K_SEED_COUNT);
	hipLaunchKernelGGL(k_seedl_finish, dim3(nb), dim3(256), 0, st, n, d_mz, d_mz_off, n_mz, (const int32_t*)d_kf, (const uint64_t*)d_rep_max, (const int64_t*)d_off_a, (const int64_t*)d_off_m, d_a_off, d_mini_off, d_rep_len);
	mga_prof_end(sc->stream, MGA_K_SEED_COUNT);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int mga_dev_seed_long_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, const mg128_t *d_mz, const int64_t *d_mz_off, int64_t n_mz, int max_occ,
									  const int32_t *d_occ, const uint64_t *d_val, const int64_t *d_off_a, const int64_t *d_off_m, mg128_t *d_a, int32_t *d_mini)
{
	if (n <= 0 || n_mz <= 0) return 0;
	mga_prof_begin(sc->stream, MGA_K_SEED_FILL);
	hipLaunchKernelGGL(k_seedl_expand, dim3(seedl_blocks(n_mz)), dim3(256), 0, (hipStream_t)sc->stream, *ix, n, d_mz, d_mz_off, n_mz, max_occ, d_occ, d_val, d_off_a, d_off_m, d_a, d_mini);
	mga_prof_end(sc->stream, MGA_K_SEED_FILL);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
