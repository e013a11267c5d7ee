// stage_api.hip -- public stage-level entry points of include/minigraph_amd.h (host pointers in/out).
// These are what the parity tests and foreign-language bindings call; the mapping pipeline itself
// keeps intermediate results in HBM and calls the mga_dev_* launchers directly.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mga_dev.h"
#include "dev_common.h"

extern "C" void mga_ksort_128x(int64_t n, mg128_t *a); // ksortx.c: radix_sort_128x with the reference's exact permutation
extern "C" void mga_host_unpin(void *p);
extern "C" void mga_free(void *p) { mga_host_unpin(p); free(p); } // (a buffer the library handed out may have been page-locked with mga_host_pin)

namespace {
struct dptr { // RAII for a device allocation
	void *p = 0;
	~dptr() { mga_dfree(p); }
	bool alloc(size_t n) { p = mga_dmalloc(n); return p != 0; }
	template<class T> T *as() const { return (T*)p; }
};
}

extern "C" int mga_sketch_batch(int n, const char *seq, const int64_t *off, const uint32_t *rid, int w, int k,
								mg128_t **mz, int64_t **mz_off)
{
	*mz = 0, *mz_off = 0;
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) { *mz_off = (int64_t*)calloc(1, 8); return 0; }
	const int64_t tot = off[n];
	dptr d_seq, d_off, d_rid, d_cnt, d_mzoff, d_mz;
	if (!d_seq.alloc(tot + 64) || !d_off.alloc((n + 1) * 8) || !d_cnt.alloc(n * 4) || !d_mzoff.alloc((n + 1) * 8)) return -1;
	if (rid && !d_rid.alloc(n * 4)) return -1;
	if (mga_h2d(d_seq.p, seq, tot) < 0 || mga_h2d(d_off.p, off, (n + 1) * 8) < 0) return -1;
	if (rid && mga_h2d(d_rid.p, rid, n * 4) < 0) return -1;
	struct planes_guard { mga_sctx_t *sc; ~planes_guard() { sc->sk_planes_src = 0; } } pg{SC}; // (the packed form belongs to d_seq, which dies with this call)
	{ const char *e = getenv("MGA_SKETCH_2BIT"); if (e && atoi(e) > 0 && mga_dev_pack2(SC, d_seq.as<char>(), tot) < 0) return -1; } // the sketch reads bit planes instead of bytes (k_sketch.hip)
	// long sequences are sketched in pieces (k odd: see k_sketch.hip); piece offsets are folded back into per-sequence offsets
	const int32_t PIECE = 1 << 16;
	std::vector<int32_t> items, first_item(n + 1);
	bool pieces = false;
	if (k & 1) for (int i = 0; i < n; ++i) if (off[i + 1] - off[i] > PIECE) pieces = true;
	if (pieces) {
		for (int i = 0; i < n; ++i) {
			const int64_t l = off[i + 1] - off[i];
			int64_t b = 0;
			first_item[i] = (int32_t)(items.size() / 4);
			do { items.push_back(i); items.push_back((int32_t)b); items.push_back((int32_t)(b + PIECE < l ? b + PIECE : l)); items.push_back(0); b += PIECE; } while (b < l);
		}
		first_item[n] = (int32_t)(items.size() / 4);
	}
	const int n_items = pieces ? (int)(items.size() / 4) : n;
	dptr d_items;
	if (pieces) {
		mga_dfree(d_cnt.p); mga_dfree(d_mzoff.p); d_cnt.p = d_mzoff.p = 0;
		if (!d_items.alloc(items.size() * 4) || !d_cnt.alloc((size_t)n_items * 4) || !d_mzoff.alloc((size_t)(n_items + 1) * 8)) return -1;
		if (mga_h2d(d_items.p, items.data(), items.size() * 4) < 0) return -1;
		if (mga_dev_sketch_items(SC, n_items, d_items.as<int32_t>(), d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, d_cnt.as<int32_t>(), 0, 0) < 0) return -1;
	} else if (mga_dev_sketch(SC, n, d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, d_cnt.as<int32_t>(), 0, 0) < 0) return -1;
	if (mga_dev_scan_i32_to_i64(SC, d_cnt.as<int32_t>(), n_items, d_mzoff.as<int64_t>()) < 0) return -1;
	std::vector<int64_t> it_off((size_t)n_items + 1);
	if (mga_ssync(SC) < 0 || mga_d2h(it_off.data(), d_mzoff.p, ((size_t)n_items + 1) * 8) < 0) return -1;
	const int64_t n_mz = it_off[n_items];
	int64_t *h_off = (int64_t*)malloc((n + 1) * 8);
	for (int i = 0; i <= n; ++i) h_off[i] = pieces ? it_off[first_item[i]] : it_off[i];
	if (!d_mz.alloc((size_t)n_mz * 16 + 16)) { free(h_off); return -1; }
	if (pieces) { if (mga_dev_sketch_items(SC, n_items, d_items.as<int32_t>(), d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, 0, d_mzoff.as<int64_t>(), d_mz.as<mg128_t>()) < 0) { free(h_off); return -1; } }
	else if (mga_dev_sketch(SC, n, d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, 0, d_mzoff.as<int64_t>(), d_mz.as<mg128_t>()) < 0) { free(h_off); return -1; }
	mg128_t *h_mz = (mg128_t*)malloc((size_t)n_mz * 16 + 16);
	if (mga_ssync(SC) < 0 || mga_d2h(h_mz, d_mz.p, (size_t)n_mz * 16) < 0) { free(h_off); free(h_mz); return -1; }
	*mz = h_mz, *mz_off = h_off;
	return 0;
}

extern "C" int mga_wfa_batch(int n, const char *tseq, const int64_t *t_off, const char *qseq, const int64_t *q_off,
							 int32_t **score, uint32_t **cigar, int64_t **cig_off)
{
	*score = 0, *cigar = 0, *cig_off = 0;
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) { *cig_off = (int64_t*)calloc(1, 8); return 0; }
	const int64_t tt = t_off[n], tq = q_off[n];
	std::vector<mga_wfa_prob_t> prob(n);
	for (int i = 0; i < n; ++i) {
		prob[i].t_off = t_off[i], prob[i].q_off = q_off[i];
		prob[i].tl = (int32_t)(t_off[i + 1] - t_off[i]), prob[i].ql = (int32_t)(q_off[i + 1] - q_off[i]);
		if (prob[i].tl <= 0 || prob[i].ql <= 0) { mga_set_error("wfa: problem %d has an empty sequence (the caller handles those, galign.c:98-100)", i); return -1; }
	}
	dptr d_t, d_q, d_prob, d_res, d_pool, d_used;
	int64_t pool_cap = (tt + tq) / 4 + n * 4 + 1024 + 40000LL * 512 + MGA_WFA_FUSE_SLACK;
	if (!d_t.alloc(tt + 64) || !d_q.alloc(tq + 64) || !d_prob.alloc((size_t)n * sizeof(mga_wfa_prob_t)) ||
		!d_res.alloc((size_t)n * sizeof(mga_wfa_res_t)) || !d_used.alloc(8)) return -1;
	if (mga_h2d(d_t.p, tseq, tt) < 0 || mga_h2d(d_q.p, qseq, tq) < 0 || mga_h2d(d_prob.p, prob.data(), (size_t)n * sizeof(mga_wfa_prob_t)) < 0) return -1;
	if (mga_dmemset_s(SC, (char*)d_t.p + tt, 0, 64) < 0 || mga_dmemset_s(SC, (char*)d_q.p + tq, 0, 64) < 0) return -1;

	std::vector<mga_wfa_res_t> res(n);
	if (!d_pool.alloc((size_t)pool_cap * 4) || mga_dmemset_s(SC, d_used.p, 0, 8) < 0) return -1;
	if (mga_dev_wfa_solve(SC, n, d_prob.as<mga_wfa_prob_t>(), d_t.as<char>(), d_q.as<char>(), d_res.as<mga_wfa_res_t>(),
						  d_pool.as<uint32_t>(), pool_cap, (unsigned long long*)d_used.p, 0, 0, 0) < 0) return -1;
	if (mga_dsync() < 0 || mga_d2h(res.data(), d_res.p, (size_t)n * sizeof(mga_wfa_res_t)) < 0) return -1;
	unsigned long long used = 0;
	if (mga_dsync() < 0 || mga_d2h(&used, d_used.p, 8) < 0) return -1;
	std::vector<uint32_t> hpool((size_t)used + 1);
	if (used && mga_d2h(hpool.data(), d_pool.p, (size_t)used * 4) < 0) return -1;
	int32_t *h_score = (int32_t*)malloc((size_t)n * 4);
	int64_t *h_off = (int64_t*)malloc((size_t)(n + 1) * 8);
	int64_t tot = 0;
	for (int i = 0; i < n; ++i) { h_off[i] = tot; h_score[i] = res[i].score; if (res[i].status == MGA_WFA_OK) tot += res[i].n_cigar; }
	h_off[n] = tot;
	uint32_t *h_cig = (uint32_t*)malloc((size_t)tot * 4 + 4);
	for (int i = 0; i < n; ++i)
		if (res[i].status == MGA_WFA_OK && res[i].n_cigar)
			memcpy(h_cig + h_off[i], hpool.data() + res[i].cig_off, (size_t)res[i].n_cigar * 4);
	*score = h_score, *cigar = h_cig, *cig_off = h_off;
	return 0;
}

struct mg_idx_bucket_s_view { mga_didx_t dev; }; // first member of the hidden index struct (mga_host.h)

extern "C" int mga_seed_batch(const mg_idx_t *gi, int n, const mg128_t *mz, const int64_t *mz_off, int max_occ,
							  mg128_t **a, int64_t **a_off, int32_t **rep_len, int32_t **mini_pos, int64_t **mini_off)
{
	*a = 0, *a_off = 0, *rep_len = 0, *mini_pos = 0, *mini_off = 0;
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) { *a_off = (int64_t*)calloc(1, 8); *mini_off = (int64_t*)calloc(1, 8); return 0; }
	const mga_didx_t *ix = &((const mg_idx_bucket_s_view*)gi->B)->dev;
	const int64_t n_mz = mz_off[n];
	dptr d_mz, d_mzoff, d_occ, d_val, d_na, d_nmini, d_rep, d_aoff, d_minioff, d_a, d_tmp, d_mini;
	if (!d_mz.alloc((size_t)n_mz * 16 + 16) || !d_mzoff.alloc((n + 1) * 8) || !d_occ.alloc((size_t)n_mz * 4 + 4) || !d_val.alloc((size_t)n_mz * 8 + 8) ||
		!d_na.alloc(n * 4) || !d_nmini.alloc(n * 4) || !d_rep.alloc(n * 4) || !d_aoff.alloc((n + 1) * 8) || !d_minioff.alloc((n + 1) * 8)) return -1;
	if (mga_h2d(d_mz.p, mz, (size_t)n_mz * 16) < 0 || mga_h2d(d_mzoff.p, mz_off, (n + 1) * 8) < 0) return -1;
	// MGA_SEED_LONG=1: the intra-read parallel kernels of the long-query path (k_seed.hip), anchors sorted here like the host chainer does
	const char *e_long = getenv("MGA_SEED_LONG");
	const bool use_long = e_long && atoi(e_long) > 0;
	dptr d_tk, d_kf, d_offa, d_offm, d_rkey, d_rmax;
	if (use_long) {
		if (!d_tk.alloc((size_t)n_mz * 4 + 4) || !d_kf.alloc((size_t)n_mz * 4 + 4) || !d_offa.alloc((size_t)(n_mz + 1) * 8) || !d_offm.alloc((size_t)(n_mz + 1) * 8) ||
			!d_rkey.alloc((size_t)n_mz * 8 + 8) || !d_rmax.alloc((size_t)n_mz * 8 + 8)) return -1;
		if (mga_dev_seed_long_count(SC, ix, n, d_mz.as<mg128_t>(), d_mzoff.as<int64_t>(), n_mz, max_occ, d_occ.as<int32_t>(), d_val.as<uint64_t>(), d_tk.as<int32_t>(), d_kf.as<int32_t>(),
									d_offa.as<int64_t>(), d_offm.as<int64_t>(), d_rkey.as<uint64_t>(), d_rmax.as<uint64_t>(), d_aoff.as<int64_t>(), d_minioff.as<int64_t>(), d_rep.as<int32_t>()) < 0) return -1;
	} else {
		if (mga_dev_seed_count(SC, ix, n, d_mz.as<mg128_t>(), d_mzoff.as<int64_t>(), 0, max_occ, d_occ.as<int32_t>(), d_val.as<uint64_t>(),
							   d_na.as<int32_t>(), d_nmini.as<int32_t>(), d_rep.as<int32_t>()) < 0) return -1;
		if (mga_dev_scan_i32_to_i64(SC, d_na.as<int32_t>(), n, d_aoff.as<int64_t>()) < 0) return -1;
		if (mga_dev_scan_i32_to_i64(SC, d_nmini.as<int32_t>(), n, d_minioff.as<int64_t>()) < 0) return -1;
	}
	int64_t *h_aoff = (int64_t*)malloc((n + 1) * 8), *h_moff = (int64_t*)malloc((n + 1) * 8);
	int32_t *h_rep = (int32_t*)malloc(n * 4);
	if (mga_ssync(SC) < 0 || mga_d2h(h_aoff, d_aoff.p, (n + 1) * 8) < 0 || mga_d2h(h_moff, d_minioff.p, (n + 1) * 8) < 0 || mga_d2h(h_rep, d_rep.p, n * 4) < 0) return -1;
	const int64_t n_a = h_aoff[n], n_m = h_moff[n];
	if (!d_a.alloc((size_t)n_a * 16 + 64) || !d_mini.alloc((size_t)n_m * 4 + 16)) return -1;
	if (use_long) {
		if (mga_dev_seed_long_fill(SC, ix, n, d_mz.as<mg128_t>(), d_mzoff.as<int64_t>(), n_mz, max_occ, d_occ.as<int32_t>(), d_val.as<uint64_t>(), d_offa.as<int64_t>(), d_offm.as<int64_t>(),
								   d_a.as<mg128_t>(), d_mini.as<int32_t>()) < 0) return -1;
	} else {
		if (!d_tmp.alloc((size_t)n_a * 16 + 64)) return -1;
		if (mga_dev_seed_fill(SC, ix, n, d_mz.as<mg128_t>(), d_mzoff.as<int64_t>(), 0, max_occ, d_occ.as<int32_t>(), d_val.as<uint64_t>(),
							  d_aoff.as<int64_t>(), d_a.as<mg128_t>(), d_minioff.as<int64_t>(), d_mini.as<int32_t>(), d_tmp.as<mg128_t>()) < 0) return -1;
	}
	mg128_t *h_a = (mg128_t*)malloc((size_t)n_a * 16 + 16);
	int32_t *h_mini = (int32_t*)malloc((size_t)n_m * 4 + 4);
	if (mga_ssync(SC) < 0 || mga_d2h(h_a, d_a.p, (size_t)n_a * 16) < 0 || mga_d2h(h_mini, d_mini.p, (size_t)n_m * 4) < 0) return -1;
	if (use_long) for (int i = 0; i < n; ++i) mga_ksort_128x(h_aoff[i + 1] - h_aoff[i], h_a + h_aoff[i]); // radix_sort_128x, map-algo.c:189
	*a = h_a, *a_off = h_aoff, *rep_len = h_rep, *mini_pos = h_mini, *mini_off = h_moff;
	return 0;
}

extern "C" int mga_gaf_div_batch(int n, const float *div, char *out)
{
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) return 0;
	dptr d_div, d_out;
	if (!d_div.alloc((size_t)n * 4 + 16) || !d_out.alloc((size_t)n * 8 + 16)) return -1;
	if (mga_h2d(d_div.p, div, (size_t)n * 4) < 0 || mga_dev_gaf_div(SC, n, d_div.as<float>(), d_out.as<char>()) < 0) return -1;
	if (mga_ssync(SC) < 0 || mga_d2h(out, d_out.p, (size_t)n * 8) < 0) return -1;
	return 0;
}

extern "C" int mga_sort128x_batch(int n, mg128_t *a, const int64_t *a_off)
{
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) return 0;
	const int64_t tot = a_off[n];
	dptr d_a, d_aoff, d_tmp, d_stk;
	if (!d_a.alloc((size_t)tot * 16 + 16) || !d_aoff.alloc((n + 1) * 8) || !d_tmp.alloc((size_t)tot * 16 + 16) || !d_stk.alloc((size_t)tot * 12 + (size_t)n * 32 + 64)) return -1;
	if (mga_h2d(d_a.p, a, (size_t)tot * 16) < 0 || mga_h2d(d_aoff.p, a_off, (n + 1) * 8) < 0) return -1;
	if (mga_dev_sort128x(SC, n, d_a.as<mg128_t>(), d_aoff.as<int64_t>(), d_tmp.as<mg128_t>(), d_stk.as<int32_t>()) < 0) return -1;
	if (mga_ssync(SC) < 0 || mga_d2h(a, d_a.p, (size_t)tot * 16) < 0) return -1;
	return 0;
}

extern "C" int mga_lchain_batch(int n, const mg128_t *a, const int64_t *a_off, const mga_lchain_par_t *par,
								uint64_t **u, int64_t **u_off, mg128_t **b, int64_t **b_off)
{
	*u = 0, *u_off = 0, *b = 0, *b_off = 0;
	if (mga_dev_init() < 0) return -1;
	mga_sctx_t *SC = mga_sctx_default();
	if (SC == 0) return -1;
	if (n <= 0) { *u_off = (int64_t*)calloc(1, 8); *b_off = (int64_t*)calloc(1, 8); return 0; }
	const int64_t tot = a_off[n];
	dptr d_a, d_aoff, d_u, d_b, d_nu, d_nb, d_ws;
	const size_t wsb = mga_dev_lchain_ws_bytes(tot);
	if (!d_a.alloc((size_t)tot * 16 + 16) || !d_aoff.alloc((n + 1) * 8) || !d_u.alloc((size_t)tot * 8 + 8) || !d_b.alloc((size_t)tot * 16 + 16) ||
		!d_nu.alloc(n * 4) || !d_nb.alloc(n * 4) || !d_ws.alloc(wsb)) return -1;
	if (mga_h2d(d_a.p, a, (size_t)tot * 16) < 0 || mga_h2d(d_aoff.p, a_off, (n + 1) * 8) < 0) return -1;
	if (mga_dev_lchain(SC, n, d_a.as<mg128_t>(), d_aoff.as<int64_t>(), par, 0, 0, d_u.as<uint64_t>(), d_b.as<mg128_t>(), d_nu.as<int32_t>(), d_nb.as<int32_t>(), 0,
					   d_ws.p, wsb, tot) < 0) return -1;
	std::vector<int32_t> nu(n), nb(n);
	std::vector<uint64_t> hu((size_t)tot + 1);
	std::vector<mg128_t> hb((size_t)tot + 1);
	if (mga_ssync(SC) < 0 || mga_d2h(nu.data(), d_nu.p, n * 4) < 0 || mga_d2h(nb.data(), d_nb.p, n * 4) < 0 ||
		mga_d2h(hu.data(), d_u.p, (size_t)tot * 8) < 0 || mga_d2h(hb.data(), d_b.p, (size_t)tot * 16) < 0) return -1;
	int64_t *uo = (int64_t*)malloc((n + 1) * 8), *bo = (int64_t*)malloc((n + 1) * 8);
	int64_t tu = 0, tb = 0;
	for (int i = 0; i < n; ++i) { uo[i] = tu, bo[i] = tb; tu += nu[i], tb += nb[i]; }
	uo[n] = tu, bo[n] = tb;
	uint64_t *ou = (uint64_t*)malloc((size_t)tu * 8 + 8);
	mg128_t *ob = (mg128_t*)malloc((size_t)tb * 16 + 16);
	for (int i = 0; i < n; ++i) {
		memcpy(ou + uo[i], hu.data() + a_off[i], (size_t)nu[i] * 8);
		memcpy(ob + bo[i], hb.data() + a_off[i], (size_t)nb[i] * 16);
	}
	*u = ou, *u_off = uo, *b = ob, *b_off = bo;
	return 0;
}
