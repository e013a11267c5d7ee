// stage_api.hip -- public stage-level entry points of include/minigraph_amd.h (host pointers in/out).
// These are what the parity tests and foreign-language bindings call; the mapping pipeline itself
// keeps intermediate results in HBM and calls the mga_dev_* launchers directly.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mga_dev.h"
#include "dev_common.h"

extern "C" void mga_free(void *p) { free(p); }

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
	if (n <= 0) { *mz_off = (int64_t*)calloc(1, 8); return 0; }
	const int64_t tot = off[n];
	dptr d_seq, d_off, d_rid, d_cnt, d_mzoff, d_mz;
	if (!d_seq.alloc(tot + 64) || !d_off.alloc((n + 1) * 8) || !d_cnt.alloc(n * 4) || !d_mzoff.alloc((n + 1) * 8)) return -1;
	if (rid && !d_rid.alloc(n * 4)) return -1;
	if (mga_h2d(d_seq.p, seq, tot) < 0 || mga_h2d(d_off.p, off, (n + 1) * 8) < 0) return -1;
	if (rid && mga_h2d(d_rid.p, rid, n * 4) < 0) return -1;
	if (mga_dev_sketch(n, d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, d_cnt.as<int32_t>(), 0, 0) < 0) return -1;
	if (mga_dev_scan_i32_to_i64(d_cnt.as<int32_t>(), n, d_mzoff.as<int64_t>()) < 0) return -1;
	int64_t *h_off = (int64_t*)malloc((n + 1) * 8);
	if (mga_d2h(h_off, d_mzoff.p, (n + 1) * 8) < 0) { free(h_off); return -1; }
	const int64_t n_mz = h_off[n];
	if (!d_mz.alloc((size_t)n_mz * 16 + 16)) { free(h_off); return -1; }
	if (mga_dev_sketch(n, d_seq.as<char>(), d_off.as<int64_t>(), d_rid.as<uint32_t>(), w, k, 0, d_mzoff.as<int64_t>(), d_mz.as<mg128_t>()) < 0) { free(h_off); return -1; }
	mg128_t *h_mz = (mg128_t*)malloc((size_t)n_mz * 16 + 16);
	if (mga_d2h(h_mz, d_mz.p, (size_t)n_mz * 16) < 0 || mga_dsync() < 0) { free(h_off); free(h_mz); return -1; }
	*mz = h_mz, *mz_off = h_off;
	return 0;
}

extern "C" int mga_wfa_batch(int n, const char *tseq, const int64_t *t_off, const char *qseq, const int64_t *q_off,
							 int32_t **score, uint32_t **cigar, int64_t **cig_off)
{
	*score = 0, *cigar = 0, *cig_off = 0;
	if (mga_dev_init() < 0) return -1;
	if (n <= 0) { *cig_off = (int64_t*)calloc(1, 8); return 0; }
	const int64_t tt = t_off[n], tq = q_off[n];
	std::vector<mga_wfa_prob_t> prob(n);
	for (int i = 0; i < n; ++i) {
		prob[i].t_off = t_off[i], prob[i].q_off = q_off[i];
		prob[i].tl = (int32_t)(t_off[i + 1] - t_off[i]), prob[i].ql = (int32_t)(q_off[i + 1] - q_off[i]);
		if (prob[i].tl <= 0 || prob[i].ql <= 0) { mga_set_error("wfa: problem %d has an empty sequence (the caller handles those, galign.c:98-100)", i); return -1; }
	}
	dptr d_t, d_q, d_prob, d_res, d_pool, d_used, d_list;
	int64_t pool_cap = (tt + tq) / 4 + n * 4 + 1024;
	if (!d_t.alloc(tt + 64) || !d_q.alloc(tq + 64) || !d_prob.alloc((size_t)n * sizeof(mga_wfa_prob_t)) ||
		!d_res.alloc((size_t)n * sizeof(mga_wfa_res_t)) || !d_used.alloc(8)) return -1;
	if (mga_h2d(d_t.p, tseq, tt) < 0 || mga_h2d(d_q.p, qseq, tq) < 0 || mga_h2d(d_prob.p, prob.data(), (size_t)n * sizeof(mga_wfa_prob_t)) < 0) return -1;
	if (mga_dmemset((char*)d_t.p + tt, 0, 64) < 0 || mga_dmemset((char*)d_q.p + tq, 0, 64) < 0) return -1;

	std::vector<mga_wfa_res_t> res(n);
	std::vector<int32_t> todo(n);
	std::vector<std::vector<uint32_t> > chunks; // CIGAR pools of successive rounds
	std::vector<int64_t> chunk_of(n, -1), off_in(n, 0);
	for (int i = 0; i < n; ++i) todo[i] = i;
	int tier = 0;
	while (!todo.empty()) {
		const int m = (int)todo.size();
		if (!d_pool.p || true) { mga_dfree(d_pool.p); d_pool.p = 0; if (!d_pool.alloc((size_t)pool_cap * 4)) return -1; }
		if (mga_dmemset(d_used.p, 0, 8) < 0) return -1;
		mga_dfree(d_list.p); d_list.p = 0;
		if (!d_list.alloc((size_t)m * 4) || mga_h2d(d_list.p, todo.data(), (size_t)m * 4) < 0) return -1;
		if (mga_dev_wfa(m, d_list.as<int32_t>(), d_prob.as<mga_wfa_prob_t>(), d_t.as<char>(), d_q.as<char>(), d_res.as<mga_wfa_res_t>(),
						d_pool.as<uint32_t>(), pool_cap, (unsigned long long*)d_used.p, tier) < 0) return -1;
		if (mga_dsync() < 0) return -1;
		if (mga_d2h(res.data(), d_res.p, (size_t)n * sizeof(mga_wfa_res_t)) < 0) return -1;
		unsigned long long used = 0;
		if (mga_d2h(&used, d_used.p, 8) < 0) return -1;
		int64_t keep = (int64_t)used < pool_cap ? (int64_t)used : pool_cap;
		chunks.push_back(std::vector<uint32_t>((size_t)keep));
		if (keep && mga_d2h(chunks.back().data(), d_pool.p, (size_t)keep * 4) < 0) return -1;
		std::vector<int32_t> next;
		bool pool_full = false, need_tier = false;
		for (int j = 0; j < m; ++j) {
			const int i = todo[j];
			if (res[i].status == MGA_WFA_OK) chunk_of[i] = (int64_t)chunks.size() - 1, off_in[i] = res[i].cig_off;
			else if (res[i].status == MGA_WFA_MAX_ITER) chunk_of[i] = -2;
			else { next.push_back(i); if (res[i].status == MGA_WFA_POOL_FULL) pool_full = true; else need_tier = true; }
		}
		if (pool_full) pool_cap *= 4;
		if (need_tier && !pool_full) {
			if (tier == 2) { mga_set_error("wfa: problem exceeds the largest capacity tier"); return -1; }
			// problems that only ran out of pool keep their tier; the rest moves up
			std::vector<int32_t> up;
			for (size_t j = 0; j < next.size(); ++j) if (res[next[j]].status == MGA_WFA_RETRY_TIER) up.push_back(next[j]);
			next.swap(up);
			++tier;
		} else if (need_tier && pool_full) {
			// rerun everything left in the same tier with the larger pool; tier escalation happens next round
		}
		todo.swap(next);
	}
	int32_t *h_score = (int32_t*)malloc((size_t)n * 4);
	int64_t *h_off = (int64_t*)malloc((size_t)(n + 1) * 8);
	int64_t tot = 0;
	for (int i = 0; i < n; ++i) { h_off[i] = tot; h_score[i] = res[i].score; if (chunk_of[i] >= 0) tot += res[i].n_cigar; }
	h_off[n] = tot;
	uint32_t *h_cig = (uint32_t*)malloc((size_t)tot * 4 + 4);
	for (int i = 0; i < n; ++i)
		if (chunk_of[i] >= 0 && res[i].n_cigar)
			memcpy(h_cig + h_off[i], chunks[(size_t)chunk_of[i]].data() + off_in[i], (size_t)res[i].n_cigar * 4);
	*score = h_score, *cigar = h_cig, *cig_off = h_off;
	return 0;
}
