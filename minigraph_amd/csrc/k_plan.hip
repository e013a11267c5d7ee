// k_plan.hip -- the gap list of base alignment on the device, for chunks whose graph chains were made there (k_gchain.hip).
//
// Reference: mg_gchain_cigar (galign.c:39-145) walks the kept anchors of a graph chain; between two consecutive ones it either
// emits a ready operator (pure match / insertion / deletion, galign.c:98-100) or aligns the query stretch against the target
// spliced from the oriented vertex sequences (galign.c:66-93).  align.c:mga_plan_cigar is that walk on a host thread, producing
// the inputs of the WFA ladder and of the text kernel; this file is the same walk for chains that never left the device, so the
// host neither touches the anchors nor uploads targets, problems and plan items ([measured] ~1 CPU-s and ~1 GB of PCIe per
// 125k reads of 10 kb):
//
//   k_plan_walk<false>  lane per read: counts printed chains, plan items, WFA problems, walk vertices and target bytes;
//   k_plan_scan5        one workgroup: the five exclusive scans (+ totals for the host, which sizes the buffers);
//   k_plan_walk<true>   the same walk, writing mga_cigitem_t / mga_wfa_prob_t / mga_txt_chain_t / vertices at the scanned offsets
//                       and, per problem, where its target lies on the walk;
//   k_plan_target       wavefront per problem: splices the target bytes from the forward / reverse-complement segment images.
//
// Which strand a line is printed on (format.c:123) depends on segment names' ranks in the host's gfa_t: the host computes that one
// flag per printed chain from the chain records it reads back anyway, and uploads it (d_rev) before the fill pass.
#include "mga_dev.h"
#include "dev_common.h"

struct plan_in_t {
	const mga_gc_hdr_t *hdr;
	const mga_gc_rec_t *gc_pool;
	const mg_llchain_t *lc_pool;
	const mg128_t *a_pool;
	const int32_t *seg_len;
	const int64_t *q_off;
	int32_t n, print_2nd;
};

struct plan_out_t {
	const int64_t *off;    // [5][n + 1]: chains, items, problems, vertices, target bytes
	const int32_t *rev;    // per printed chain
	mga_cigitem_t *item;
	mga_wfa_prob_t *prob;
	mga_plan_src_t *src;
	mga_txt_chain_t *chain;
	uint32_t *vert;
};

template<bool FILL>
__global__ void __launch_bounds__(64) k_plan_walk(const plan_in_t in, int32_t *cnt /* [5][n] */, unsigned long long *tot, const plan_out_t out)
{
	const int32_t i = blockIdx.x * 64 + threadIdx.x;
	if (i >= in.n) return;
	const mga_gc_hdr_t h = in.hdr[i];
	const int32_t n = in.n;
	int32_t n_chain = 0, n_item = 0, n_prob = 0, n_vert = 0;
	int64_t n_tb = 0, n_qb = 0;
	int64_t o_chain = 0, o_item = 0, o_prob = 0, o_vert = 0, o_tb = 0, q_base = 0;
	if (FILL) {
		o_chain = out.off[i], o_item = out.off[(int64_t)(n + 1) + i], o_prob = out.off[2 * (int64_t)(n + 1) + i];
		o_vert = out.off[3 * (int64_t)(n + 1) + i], o_tb = out.off[4 * (int64_t)(n + 1) + i];
		q_base = in.q_off[i];
	}
	if (h.status == 0) {
		const mga_gc_rec_t *gc = in.gc_pool + h.gc_off;
		const mg_llchain_t *lc = in.lc_pool + h.lc_off;
		const mg128_t *a = in.a_pool + h.a_off;
		for (int32_t k = 0; k < h.n_gc; ++k) {
			const mga_gc_rec_t r = gc[k];
			if ((r.id != r.parent && !in.print_2nd) || r.cnt == 0) continue; // not printed (format.c:135-136): no alignment needed
			int32_t l0 = r.off;
			const int32_t off_a0 = lc[l0].off, l_end = r.off + r.cnt;
			const int64_t item_beg = o_item + n_item;
			mg128_t q = a[off_a0];
			const int32_t span0 = (int32_t)(q.y >> 32 & 0xff);
			if (FILL) { mga_cigitem_t it; it.op = 7, it.val = span0; out.item[o_item + n_item] = it; }
			++n_item;
			for (int32_t j = 1; j < r.n_anchor; ++j) {
				const mg128_t p = a[off_a0 + j];
				if ((p.y & MG_SEED_IGNORE) && j != r.n_anchor - 1) continue;
				int32_t l = l0;
				for (; l < l_end; ++l) { // the vertex holding anchor j
					const int32_t lo = lc[l].off;
					if (off_a0 + j >= lo && off_a0 + j < lo + lc[l].cnt) break;
				}
				int32_t l_seq;
				if (l == l0) l_seq = (int32_t)p.x - (int32_t)q.x;
				else {
					l_seq = in.seg_len[lc[l0].v >> 1] - (int32_t)q.x - 1;
					for (int32_t t = l0 + 1; t < l; ++t) l_seq += in.seg_len[lc[t].v >> 1];
					l_seq += (int32_t)p.x + 1;
				}
				const int32_t qlen = (int32_t)p.y - (int32_t)q.y;
				int32_t op, val;
				if (l_seq == 0) op = 1, val = qlen;
				else if (qlen == 0) op = 2, val = l_seq;
				else if (l_seq == qlen && qlen <= (int32_t)(q.y >> 32 & 0xff)) op = 7, val = qlen;
				else { // a gap for the WFA ladder: target spliced across vertices l0..l, query = read[q.y+1 .. p.y]
					op = -1, val = (int32_t)(o_prob + n_prob);
					if (FILL) {
						mga_wfa_prob_t pb;
						mga_plan_src_t s;
						pb.t_off = o_tb + n_tb, pb.tl = l_seq, pb.q_off = q_base + (int32_t)q.y + 1, pb.ql = qlen;
						s.lc0 = h.lc_off + l0, s.n_lc = l - l0, s.x0 = (int32_t)q.x, s.x1 = (int32_t)p.x, s.pad = 0;
						out.prob[o_prob + n_prob] = pb, out.src[o_prob + n_prob] = s;
					}
					++n_prob, n_tb += l_seq, n_qb += qlen;
				}
				if (FILL) { mga_cigitem_t it; it.op = op, it.val = val; out.item[o_item + n_item] = it; }
				++n_item;
				q = p, l0 = l;
			}
			if (FILL) {
				mga_txt_chain_t c;
				const mg128_t last = a[off_a0 + r.n_anchor - 1];
				c.item_beg = item_beg, c.item_end = o_item + n_item, c.prob_base = 0, c.q_base = q_base;
				c.vert_beg = o_vert + n_vert, c.vert_cnt = r.cnt;
				c.qs = r.qs, c.qe = r.qe, c.ps = r.ps, c.pe = r.pe;
				c.ss = (int32_t)a[off_a0].x + 1 - span0; // galign.c:128-129
				c.ee = (int32_t)last.x + 1;
				c.rev_sign = out.rev[o_chain + n_chain];
				out.chain[o_chain + n_chain] = c;
				for (int32_t t = 0; t < r.cnt; ++t) out.vert[o_vert + n_vert + t] = lc[r.off + t].v;
			}
			++n_chain, n_vert += r.cnt;
		}
	}
	if (!FILL) {
		cnt[i] = n_chain, cnt[(int64_t)n + i] = n_item, cnt[2 * (int64_t)n + i] = n_prob, cnt[3 * (int64_t)n + i] = n_vert;
		cnt[4 * (int64_t)n + i] = n_tb > 0x7fffffffLL ? 0x7fffffff : (int32_t)n_tb;
		if (n_qb) atomicAdd(&tot[5], (unsigned long long)n_qb);
		if (n_tb > 0x7fffffffLL) atomicAdd(&tot[6], 1ULL); // a read with >= 2 Gbp of gap targets: the host fails the chunk loudly
	}
}

// five exclusive scans of int32 counts into int64 offsets by one workgroup (n = reads of a chunk); tot[a] = the a-th total
__global__ void __launch_bounds__(1024) k_plan_scan5(const int32_t *cnt, int32_t n, int64_t *off, unsigned long long *tot)
{
	__shared__ int64_t s_part[1024];
	const int32_t t = threadIdx.x, per = (n + 1023) / 1024;
	const int32_t b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
	for (int a = 0; a < 5; ++a) {
		const int32_t *c = cnt + (int64_t)a * n;
		int64_t *o = off + (int64_t)a * (n + 1);
		int64_t sum = 0;
		for (int32_t i = b; i < e; ++i) sum += c[i];
		s_part[t] = sum;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) { // inclusive scan of the partial sums
			const int64_t y = t >= d ? s_part[t - d] : 0;
			__syncthreads();
			s_part[t] += y;
			__syncthreads();
		}
		int64_t run = s_part[t] - sum;
		for (int32_t i = b; i < e; ++i) { o[i] = run; run += c[i]; }
		if (t == 1023) { o[n] = s_part[1023]; tot[a] = (unsigned long long)s_part[1023]; }
		__syncthreads();
	}
}

// one wavefront per problem: its target = the walk from base x0+1 of vertex lc0 to base x1 of vertex lc0+n_lc (galign.c:66-93)
__global__ void __launch_bounds__(256) k_plan_target(int64_t n_prob, const mga_wfa_prob_t *prob, const mga_plan_src_t *src, const mg_llchain_t *lc_pool,
													 const int32_t *seg_len, const int64_t *gseq_off, const char *gseq, const char *gseq_rc, char *tseq)
{
	const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_prob) return;
	const mga_plan_src_t s = src[w];
	char *dst = tseq + prob[w].t_off;
	int32_t pos = 0;
	for (int32_t t = 0; t <= s.n_lc; ++t) {
		const uint32_t v = lc_pool[s.lc0 + t].v;
		const char *base = ((v & 1) ? gseq_rc : gseq) + gseq_off[v >> 1];
		const int32_t b = t == 0 ? s.x0 + 1 : 0, e = t == s.n_lc ? s.x1 + 1 : seg_len[v >> 1];
		for (int32_t o = b + lane; o < e; o += 64) dst[pos + (o - b)] = base[o];
		pos += e - b;
	}
}

// the same for gaps listed by the HOST (round 5: host threads chain and list, the device splices): the walk is the flattened vertex array of the text kernel
__global__ void __launch_bounds__(256) k_plan_target_v(int64_t n_prob, const mga_wfa_prob_t *prob, const mga_plan_src_t *src, const uint32_t *vert,
													   const int32_t *seg_len, const int64_t *gseq_off, const char *gseq, const char *gseq_rc, char *tseq)
{
	const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_prob) return;
	const mga_plan_src_t s = src[w];
	char *dst = tseq + prob[w].t_off;
	int32_t pos = 0;
	for (int32_t t = 0; t <= s.n_lc; ++t) {
		const uint32_t v = vert[s.lc0 + t];
		const char *base = ((v & 1) ? gseq_rc : gseq) + gseq_off[v >> 1];
		const int32_t b = t == 0 ? s.x0 + 1 : 0, e = t == s.n_lc ? s.x1 + 1 : seg_len[v >> 1];
		for (int32_t o = b + lane; o < e; o += 64) dst[pos + (o - b)] = base[o];
		pos += e - b;
	}
}

extern "C" int mga_dev_plan_target_verts(mga_sctx_t *sc, const mga_didx_t *ix, int64_t n_prob, const mga_wfa_prob_t *d_prob, const mga_plan_src_t *d_src, const uint32_t *d_vert, char *d_tseq)
{
	if (n_prob <= 0) return 0;
	hipStream_t st = (hipStream_t)sc->stream;
	mga_prof_begin(sc->stream, MGA_K_PLAN);
	hipLaunchKernelGGL(k_plan_target_v, dim3((unsigned)((n_prob + 3) / 4)), dim3(256), 0, st, n_prob, d_prob, d_src, d_vert, (const int32_t*)ix->d_seg_len, (const int64_t*)ix->d_gseq_off,
					   (const char*)ix->d_gseq, (const char*)ix->d_gseq_rc, d_tseq);
	mga_prof_end(sc->stream, MGA_K_PLAN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

static void plan_in_fill(plan_in_t *in, const mga_didx_t *ix, int n, int print_2nd, const mga_gc_hdr_t *d_hdr, const void *d_gc_pool, const mg_llchain_t *d_lc_pool,
						 const mg128_t *d_a_pool, const int64_t *d_q_off)
{
	in->hdr = d_hdr, in->gc_pool = (const mga_gc_rec_t*)d_gc_pool, in->lc_pool = d_lc_pool, in->a_pool = d_a_pool, in->seg_len = ix->d_seg_len, in->q_off = d_q_off;
	in->n = n, in->print_2nd = print_2nd;
}

// pass 1: d_cnt = int32[5][n] scratch, d_off = int64[5][n+1] (chains, items, problems, vertices, target bytes),
// d_tot = 8 x uint64: the five totals, [5] = query bases of all problems, [6] = reads whose target bytes overflowed
extern "C" int mga_dev_plan_count(mga_sctx_t *sc, const mga_didx_t *ix, int n, int print_2nd, const mga_gc_hdr_t *d_hdr, const void *d_gc_pool, const mg_llchain_t *d_lc_pool,
								  const mg128_t *d_a_pool, int32_t *d_cnt, int64_t *d_off, unsigned long long *d_tot)
{
	hipStream_t st = (hipStream_t)sc->stream;
	plan_in_t in;
	plan_out_t out = {};
	plan_in_fill(&in, ix, n, print_2nd, d_hdr, d_gc_pool, d_lc_pool, d_a_pool, 0);
	MGA_HIP_CHECK(hipMemsetAsync(d_tot, 0, 64, st));
	if (n <= 0) { MGA_HIP_CHECK(hipMemsetAsync(d_off, 0, 5 * 8, st)); return 0; }
	mga_prof_begin(sc->stream, MGA_K_PLAN);
	hipLaunchKernelGGL(k_plan_walk<false>, dim3((n + 63) / 64), dim3(64), 0, st, in, d_cnt, d_tot, out);
	hipLaunchKernelGGL(k_plan_scan5, dim3(1), dim3(1024), 0, st, (const int32_t*)d_cnt, n, d_off, d_tot);
	mga_prof_end(sc->stream, MGA_K_PLAN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

// pass 2: everything the WFA ladder and the text kernel read, at the scanned offsets; d_tseq gets the n_prob targets
extern "C" int mga_dev_plan_fill(mga_sctx_t *sc, const mga_didx_t *ix, int n, int print_2nd, const mga_gc_hdr_t *d_hdr, const void *d_gc_pool, const mg_llchain_t *d_lc_pool,
								 const mg128_t *d_a_pool, const int64_t *d_q_off, const int64_t *d_off, const int32_t *d_rev, int64_t n_prob,
								 mga_cigitem_t *d_item, mga_wfa_prob_t *d_prob, mga_plan_src_t *d_src, mga_txt_chain_t *d_chain, uint32_t *d_vert, char *d_tseq)
{
	hipStream_t st = (hipStream_t)sc->stream;
	plan_in_t in;
	plan_out_t out;
	if (n <= 0) return 0;
	plan_in_fill(&in, ix, n, print_2nd, d_hdr, d_gc_pool, d_lc_pool, d_a_pool, d_q_off);
	out.off = d_off, out.rev = d_rev, out.item = d_item, out.prob = d_prob, out.src = d_src, out.chain = d_chain, out.vert = d_vert;
	mga_prof_begin(sc->stream, MGA_K_PLAN);
	hipLaunchKernelGGL(k_plan_walk<true>, dim3((n + 63) / 64), dim3(64), 0, st, in, (int32_t*)0, (unsigned long long*)0, out);
	if (n_prob > 0)
		hipLaunchKernelGGL(k_plan_target, dim3((unsigned)((n_prob + 3) / 4)), dim3(256), 0, st, n_prob, (const mga_wfa_prob_t*)d_prob, (const mga_plan_src_t*)d_src, d_lc_pool,
						   (const int32_t*)ix->d_seg_len, (const int64_t*)ix->d_gseq_off, (const char*)ix->d_gseq, (const char*)ix->d_gseq_rc, d_tseq);
	mga_prof_end(sc->stream, MGA_K_PLAN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
