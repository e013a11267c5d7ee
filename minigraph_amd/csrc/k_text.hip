// k_text.hip -- alignment text on the device: CIGAR stitching (mg_gchain_cigar, galign.c:39-145), the ds:Z difference
// string (mg_gchain_gen_ds, galign.c:182-293) and the cg:Z / ds:Z fields of the GAF line (format.c:205-246), for
// the path that only wants GAF bytes (mga_map_reads).  These three loops cost the host 45 us per 10 kb read
// ([measured]: ~1600 branchy operators per read), half of its budget.  One wavefront per printed chain:
//
//   S1  the plan items (ready operators / references to WFA problems, align.c:mga_plan_cigar) are expanded into
//       the concatenated operator list (64 output slots per step, owner item found by binary search over the
//       lanes' offsets);
//   S2  runs: an element starts a new operator unless it is allowed to merge (append_cigar1: every ready operator
//       and the FIRST operator of a WFA CIGAR) and equals the previous operator; run lengths by atomic adds;
//   S3  per run (lane = run): target/query coordinates by wave scans, statistics (mlen/blen/aplen), the length of
//       its cg:Z piece and of its ds:Z entries (mismatch runs compare nt4 codes base by base; indels look for
//       micro-homology, write_indel galign.c:153-180);
//   S4  space for the two strings is taken from the text pool with one atomic, then every run writes its pieces
//       at its scanned offset -- from the END of the strings, entry by entry reversed/complemented, when the line
//       is printed on the reverse strand (rev_sign, format.c:123,183,217-241).
// k_text_count (lane per chain) sizes the element lists and tabulates where each vertex of the walk starts.
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_lcscan.h"
#include <string.h>
#include <stdlib.h>

struct txt_tables_t { unsigned char comp[256], nt4[256]; };
__constant__ txt_tables_t c_txt;

struct txt_walk_t {
	const uint32_t *vert; const int32_t *vwb; int32_t cnt, ss; // vwb[k]: walk position where vertex k starts
	const char *gseq; const int64_t *gseq_off; const int32_t *seg_len;
};

// base x of the aligned stretch of the walk (what mg_gchain_gen_ds copies into seq[], galign.c:195-200)
__device__ __forceinline__ char walk_get(const txt_walk_t &W, int32_t x)
{
	int32_t lo = 0, hi = W.cnt - 1;
	while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (W.vwb[mid] <= x) lo = mid; else hi = mid - 1; }
	const uint32_t v = W.vert[lo];
	const int32_t p = (lo > 0 ? 0 : W.ss) + (x - W.vwb[lo]);
	const int64_t o = W.gseq_off[v >> 1];
	if (!(v & 1)) return W.gseq[o + p];
	return (char)c_txt.comp[(unsigned char)W.gseq[o + (W.seg_len[v >> 1] - 1 - p)]]; // gfa_edseq_init: reverse complement (gfa-ed.c:24-42)
}

__device__ __forceinline__ int txt_ndigits(uint32_t x) { int n = 1; while (x >= 10) x /= 10, ++n; return n; }
__device__ __forceinline__ void txt_put_uint(char *w, uint32_t x, int nd) { for (int i = nd - 1; i >= 0; --i) { w[i] = (char)('0' + x % 10); x /= 10; } }

// ---- sizes: elements per chain, walk offsets of its vertices (lane per chain) ----
__global__ void __launch_bounds__(64) k_text_count(int n_chain, const mga_txt_chain_t *__restrict__ chain, const mga_cigitem_t *__restrict__ item, const uint32_t *__restrict__ vert,
												   const int32_t *__restrict__ seg_len, const int32_t *__restrict__ ncig, int32_t *__restrict__ n_el, int32_t *__restrict__ vwb)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_chain) return;
	const mga_txt_chain_t C = chain[c];
	int32_t n = 0, w = 0;
	for (int64_t t = C.item_beg; t < C.item_end; ++t) { const mga_cigitem_t it = item[t]; n += it.op >= 0 ? 1 : ncig[C.prob_base + it.val]; }
	n_el[c] = n;
	for (int32_t k = 0; k < C.vert_cnt; ++k) {
		const int32_t len = seg_len[vert[C.vert_beg + k] >> 1];
		vwb[C.vert_beg + k] = w;
		w += (k < C.vert_cnt - 1 ? len : C.ee) - (k > 0 ? 0 : C.ss);
	}
}

// the same for chromosome-scale chains (-x asm: a handful per launch, 10^6 plan items and 10^5 walk vertices each -- [measured, round 4] a lane walked them alone for 1.1 + 1.95 s of a
// 6.6 s job): a workgroup per chain, items summed by 256 threads, the walk offsets as an exclusive running sum a tile of 256 vertices at a time
__global__ void __launch_bounds__(256) k_text_count_wg(int n_chain, const mga_txt_chain_t *__restrict__ chain, const mga_cigitem_t *__restrict__ item, const uint32_t *__restrict__ vert,
													  const int32_t *__restrict__ seg_len, const int32_t *__restrict__ ncig, int32_t *__restrict__ n_el, int32_t *__restrict__ vwb)
{
	__shared__ int32_t red[256];
	const int c = blockIdx.x, tid = threadIdx.x;
	if (c >= n_chain) return; // (uniform over the workgroup)
	const mga_txt_chain_t C = chain[c];
	int32_t n = 0;
	for (int64_t t = C.item_beg + tid; t < C.item_end; t += 256) { const mga_cigitem_t it = item[t]; n += it.op >= 0 ? 1 : ncig[C.prob_base + it.val]; }
	red[tid] = n;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
	if (tid == 0) n_el[c] = red[0];
	int32_t carry = 0;
	for (int32_t k0 = 0; k0 < C.vert_cnt; k0 += 256) {
		const int32_t k = k0 + tid;
		int32_t len = 0;
		if (k < C.vert_cnt) len = (k < C.vert_cnt - 1 ? seg_len[vert[C.vert_beg + k] >> 1] : C.ee) - (k > 0 ? 0 : C.ss);
		__syncthreads(); // red[] of the previous tile (or of the sum above) has been read by everyone
		red[tid] = len;
		__syncthreads();
		for (int s = 1; s < 256; s <<= 1) { // inclusive scan
			const int32_t v = tid >= s ? red[tid - s] : 0;
			__syncthreads();
			red[tid] += v;
			__syncthreads();
		}
		if (k < C.vert_cnt) vwb[C.vert_beg + k] = carry + red[tid] - len;
		carry += red[255];
	}
}

// ds:Z entries of one run: returns their total length; WRITE: stores them at w (REV: from w + n backwards, each entry transformed)
template<bool WRITE, bool REV> __device__ int32_t txt_ds_run(const txt_walk_t &W, const char *q, int32_t op, int32_t len, int32_t x, int32_t y, int32_t qs, int32_t qe, int32_t apl,
															char *w, int32_t n_total)
{
	int32_t n = 0; // bytes produced so far
#define TXT_PLACE(len_) (REV ? w + (n_total - n - (len_)) : w + n)
	if (op == 7) { // byte-identical bases: one ":len" entry (the reference's base loop sees len equal codes, galign.c:228-243)
		if (len > 0) { const int nd = txt_ndigits((uint32_t)len); if (WRITE) { char *p = TXT_PLACE(1 + nd); p[0] = ':'; txt_put_uint(p + 1, (uint32_t)len, nd); } n += 1 + nd; }
	} else if (op == 0 || op == 8) {
		int32_t l = 0;
		for (int32_t z = 0; z <= len; ++z) {
			unsigned char cx = 0, cy = 0;
			const bool last = z == len;
			if (!last) cx = c_txt.nt4[(unsigned char)walk_get(W, x + z)], cy = c_txt.nt4[(unsigned char)q[y + z]];
			if (last || cx != cy) {
				if (l > 0) { const int nd = txt_ndigits((uint32_t)l); if (WRITE) { char *p = TXT_PLACE(1 + nd); p[0] = ':'; txt_put_uint(p + 1, (uint32_t)l, nd); } n += 1 + nd; }
				if (!last) {
					if (WRITE) {
						char *p = TXT_PLACE(3);
						char c1 = "acgtn"[cx], c2 = "acgtn"[cy];
						if (REV) c1 = (char)c_txt.comp[(unsigned char)c1], c2 = (char)c_txt.comp[(unsigned char)c2]; // format.c:225-226: same order, complemented
						p[0] = '*', p[1] = c1, p[2] = c2;
					}
					n += 3;
				}
				l = 0;
			} else ++l;
		}
	} else if (op == 1 || op == 2) { // micro-homology on either side (galign.c:229-249), then write_indel (galign.c:153-180)
		const bool ins = op == 1;
		int32_t z, ll, lr;
		if (ins) {
			for (z = 1; z <= len; ++z) if (y - z < qs || q[y + len - z] != q[y - z]) break;
			lr = z - 1;
			for (z = 0; z < len; ++z) if (y + len + z >= qe || q[y + len + z] != q[y + z]) break;
			ll = z;
		} else {
			for (z = 1; z <= len; ++z) if (x - z < 0 || walk_get(W, x + len - z) != walk_get(W, x - z)) break;
			lr = z - 1;
			for (z = 0; z < len; ++z) if (x + len + z >= apl || walk_get(W, x + z) != walk_get(W, x + len + z)) break;
			ll = z;
		}
		int32_t m = 1 + len;
		if (ll + lr >= len) m += 2; else m += (ll > 0 ? 2 : 0) + (lr > 0 ? 2 : 0);
		if (WRITE) {
			char *p0 = TXT_PLACE(m), *p = p0;
			*p++ = ins ? '+' : '-';
#define TXT_B(i_) ("acgtn"[c_txt.nt4[(unsigned char)(ins ? q[y + (i_)] : walk_get(W, x + (i_)))]])
			if (ll + lr >= len) {
				*p++ = '[';
				for (int32_t i = 0; i < len; ++i) *p++ = TXT_B(i);
				*p++ = ']';
			} else {
				int32_t k = 0;
				if (ll > 0) { *p++ = '['; for (int32_t i = 0; i < ll; ++i) *p++ = TXT_B(k + i); *p++ = ']'; k += ll; }
				for (int32_t i = 0; i < len - lr - ll; ++i) *p++ = TXT_B(k + i);
				k += len - lr - ll;
				if (lr > 0) { *p++ = '['; for (int32_t i = 0; i < lr; ++i) *p++ = TXT_B(k + i); *p++ = ']'; }
			}
#undef TXT_B
			if (REV) { // everything after the sign: reversed, complemented, brackets swapped (format.c:229-237)
#define TXT_T(ch_) ((ch_) == '[' ? ']' : (ch_) == ']' ? '[' : (char)c_txt.comp[(unsigned char)(ch_)])
				char *a = p0 + 1, *b = p0 + m - 1;
				while (a < b) { const char ca = TXT_T(*a), cb = TXT_T(*b); *a++ = cb; *b-- = ca; }
				if (a == b) *a = TXT_T(*a);
#undef TXT_T
			}
		}
		n += m;
	}
#undef TXT_PLACE
	return n;
}

// ---- workgroup-wide scan / sum (NT = 64: one wavefront, no LDS; NT = 1024: sixteen wavefronts, partial sums through LDS) ----
template<int NT> __device__ __forceinline__ int32_t blk_excl_scan(int32_t v, int tid, int32_t *total, int32_t *s_w)
{
	const int lane = tid & 63;
	int32_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
	if (NT == 64) { *total = __shfl(x, 63); return x - v; }
	const int w = tid >> 6;
	__syncthreads(); // (s_w of the previous call has been read by everybody)
	if (lane == 63) s_w[w] = x;
	__syncthreads();
	int32_t base = 0, tot = 0;
#pragma unroll
	for (int k = 0; k < NT / 64; ++k) { const int32_t t = s_w[k]; if (k < w) base += t; tot += t; }
	*total = tot;
	return base + x - v;
}
template<int NT> __device__ __forceinline__ int32_t blk_sum(int32_t v, int tid, int32_t *s_w)
{
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
	if (NT == 64) return v;
	__syncthreads();
	if ((tid & 63) == 0) s_w[tid >> 6] = v;
	__syncthreads();
	int32_t tot = 0;
#pragma unroll
	for (int k = 0; k < NT / 64; ++k) tot += s_w[k];
	return tot;
}

// One WORKGROUP of NT threads per printed chain.  NT = 64 for the chains of ordinary reads (~1600 operators); NT = 1024 when the launch holds chromosome-scale chains
// (-x asm contigs, ultra-long -x lr reads: 10^5 - 10^6 operators each, a handful of chains per launch -- [measured, round 3] one wavefront per 50 Mbp chain was 1.4 s of a
// 3.6 s job): the tile loops below stride by NT, the scans run over the workgroup.
template<int NT>
__global__ void __launch_bounds__(NT) k_text(int n_chain, const mga_txt_chain_t *__restrict__ chain, const mga_cigitem_t *__restrict__ item, const uint32_t *__restrict__ vert,
											 const int32_t *__restrict__ vwb, const char *__restrict__ gseq, const int64_t *__restrict__ gseq_off, const int32_t *__restrict__ seg_len,
											 const char *__restrict__ reads, const int32_t *__restrict__ ncig, const int64_t *__restrict__ cigoff, const uint32_t *__restrict__ ord,
											 const int64_t *__restrict__ el_off, uint32_t *__restrict__ el, uint32_t *__restrict__ run, int32_t *__restrict__ run_txt,
											 mga_txt_res_t *__restrict__ res, char *__restrict__ pool, long long pool_cap, unsigned long long *pool_used)
{
	__shared__ int32_t s_w[NT / 64 + 1];
	__shared__ int32_t s_o[NT], s_op[NT], s_val[NT];
	__shared__ int64_t s_src[NT];
	__shared__ unsigned long long s_res;
	const int c = blockIdx.x, tid = threadIdx.x;
	if (c >= n_chain) return;
	const mga_txt_chain_t C = chain[c];
	const int64_t eo = el_off[c];
	const int32_t n_el = (int32_t)(el_off[c + 1] - eo);
	uint32_t *E = el + eo, *R = run + eo; // elements / runs: len << 5 | mergeable << 4 | op
	int32_t *RT = run_txt + 2 * eo;      // per run: cg piece length, ds entries length
	const char *q = reads + C.q_base;
	txt_walk_t W;
	W.vert = vert + C.vert_beg, W.vwb = vwb + C.vert_beg, W.cnt = C.vert_cnt, W.ss = C.ss, W.gseq = gseq, W.gseq_off = gseq_off, W.seg_len = seg_len;

	// ---- S1: concatenated operator list: a tile of NT items at a time, every output slot finds its item by a binary search over the tile's offsets (LDS)
	{
		int32_t base = 0;
		for (int64_t tb = C.item_beg; tb < C.item_end; tb += NT) {
			const int64_t t = tb + tid;
			mga_cigitem_t it; it.op = 0, it.val = 0;
			int32_t cnt = 0;
			int64_t src = 0;
			if (t < C.item_end) {
				it = item[t];
				if (it.op >= 0) cnt = 1; else { const int64_t pj = C.prob_base + it.val; cnt = ncig[pj]; src = cigoff[pj]; }
			}
			int32_t tot;
			const int32_t o_l = blk_excl_scan<NT>(cnt, tid, &tot, s_w);
			__syncthreads(); // (the previous tile's readers are through)
			s_o[tid] = o_l, s_op[tid] = it.op, s_val[tid] = it.val, s_src[tid] = src;
			__syncthreads();
			for (int32_t o = tid; o < tot; o += NT) {
				int lo = 0;
#pragma unroll
				for (int step = NT / 2; step > 0; step >>= 1) if (s_o[lo + step] <= o) lo += step; // last item whose first slot is <= o (empty items share offsets)
				const int32_t k = o - s_o[lo], op_l = s_op[lo];
				uint32_t e;
				if (op_l >= 0) e = (uint32_t)s_val[lo] << 5 | 1u << 4 | (uint32_t)op_l;
				else { const uint32_t cg = ord[s_src[lo] + k]; e = (cg >> 4) << 5 | (k == 0 ? 1u << 4 : 0u) | (cg & 0xf); }
				E[base + o] = e;
			}
			base += tot;
		}
	}
	for (int32_t i = tid; i < n_el; i += NT) R[i] = 0;
	__threadfence_block();
	__syncthreads();
	// ---- S2: runs (append_cigar1 / append_cigar, galign.c:11-37)
	int32_t n_run = 0;
	for (int32_t b0 = 0; b0 < n_el; b0 += NT) {
		const int32_t i = b0 + tid;
		const uint32_t e = i < n_el ? E[i] : 0;
		const int32_t op = (int32_t)(e & 0xf);
		const int32_t prev = i > 0 && i < n_el ? (int32_t)(E[i - 1] & 0xf) : -1;
		const bool head = i < n_el && !((e >> 4 & 1) && op == prev);
		int32_t tot;
		const int32_t before = blk_excl_scan<NT>(head ? 1 : 0, tid, &tot, s_w);
		const int32_t ridx = n_run + before + (head ? 1 : 0) - 1;
		if (i < n_el) atomicAdd(&R[ridx], (e >> 5) << 5 | (head ? (uint32_t)op : 0u)); // lengths add up; the head contributes the operator bits
		n_run += tot;
	}
	__threadfence_block();
	__syncthreads();
	// ---- S3: coordinates, statistics, text lengths per run
	const int32_t apl = C.pe - C.ps;
	int32_t mlen = 0, blen = 0, aplen = 0, qlen = 0, cg_n = 0, ds_n = 0;
	for (int pass = 0; pass < 2; ++pass) {
		int32_t x0 = 0, y0 = C.qs, cg0 = 0, ds0 = 0;
		char *cg_base = 0, *ds_base = 0;
		if (pass == 1) {
			const unsigned long long need = (unsigned long long)cg_n + (unsigned long long)ds_n;
			if (tid == 0) s_res = atomicAdd(pool_used, need);
			__syncthreads();
			const unsigned long long o = s_res;
			mga_txt_res_t r;
			r.txt_off = (int64_t)o, r.cg_len = cg_n, r.ds_len = ds_n, r.n_cigar = n_run, r.mlen = mlen, r.blen = blen, r.aplen = aplen, r.pad = 0;
			r.status = (qlen == C.qe - C.qs && aplen == apl) ? 0 : 1; // galign.c:140
			if (r.status == 0 && (long long)(o + need) > pool_cap) r.status = 2;
			if (tid == 0) res[c] = r;
			if (r.status != 0) return;
			cg_base = pool + o, ds_base = pool + o + cg_n;
		}
		for (int32_t b0 = 0; b0 < n_run; b0 += NT) {
			const int32_t r = b0 + tid;
			const uint32_t e = r < n_run ? R[r] : 0;
			const int32_t op = (int32_t)(e & 0xf), len = (int32_t)(e >> 5);
			const int32_t dx = r < n_run && op != 1 ? len : 0, dy = r < n_run && op != 2 ? len : 0;
			int32_t tx, ty;
			const int32_t x = x0 + blk_excl_scan<NT>(dx, tid, &tx, s_w), y = y0 + blk_excl_scan<NT>(dy, tid, &ty, s_w);
			if (pass == 0) {
				int32_t lc = 0, ld = 0;
				if (r < n_run) {
					lc = txt_ndigits((uint32_t)len) + 1;
					ld = txt_ds_run<false, false>(W, q, op, len, x, y, C.qs, C.qe, apl, 0, 0);
					RT[2 * r] = lc, RT[2 * r + 1] = ld;
				}
				mlen += blk_sum<NT>(r < n_run && op == 7 ? len : 0, tid, s_w), blen += blk_sum<NT>(r < n_run ? len : 0, tid, s_w);
				aplen += tx, qlen += ty, cg_n += blk_sum<NT>(lc, tid, s_w), ds_n += blk_sum<NT>(ld, tid, s_w);
			} else {
				const int32_t lc = r < n_run ? RT[2 * r] : 0, ld = r < n_run ? RT[2 * r + 1] : 0;
				int32_t tc, td;
				const int32_t oc = cg0 + blk_excl_scan<NT>(lc, tid, &tc, s_w), od = ds0 + blk_excl_scan<NT>(ld, tid, &td, s_w);
				if (r < n_run) {
					char *wc = C.rev_sign ? cg_base + (cg_n - oc - lc) : cg_base + oc; // cg:Z piece: "<len><op>" (format.c:205-215)
					txt_put_uint(wc, (uint32_t)len, lc - 1);
					wc[lc - 1] = "MIDNSHP=XB"[op];
					if (C.rev_sign) txt_ds_run<true, true>(W, q, op, len, x, y, C.qs, C.qe, apl, ds_base + (ds_n - od - ld), ld);
					else txt_ds_run<true, false>(W, q, op, len, x, y, C.qs, C.qe, apl, ds_base + od, ld);
				}
				cg0 += tc, ds0 += td;
			}
			x0 += tx, y0 += ty;
		}
	}
}


// ---- round 6: the wavefront-per-chain form again, built for what the chains of ordinary reads look like ---------------------------------------------------------------------
// k_text<64> spent its 32 ms per 125 000 reads (35 GB/s for 1.1 GB of text, 79 % of the waves' cycles parked) on CHAINS OF DEPENDENT LOADS and on scattered byte stores:
// every target base went through a binary search over the walk's vertex table in global memory, then the vertex, its offset, its length, the base, and two tables in constant
// memory (six round trips); the run lengths were summed by global atomics into a zeroed array; the second pass read every sequence again; and every byte of text was its own
// global store (11.7 x the text's bytes in HBM traffic).  k_text_w keeps the stages (S1 expand, S2 runs, S3 count, S4 write) and changes where their data lives:
//   * the walk as <= 32 (start, pointer) pairs in LDS -- the pointer of a reverse vertex goes into the reverse-complement image of the segments (d_gseq_rc), so a base of either
//     orientation is ONE global byte load -- and a lane remembers the vertex it looked at last; the nt4 / complement tables are copied to LDS once per wavefront;
//   * run lengths by LDS atomics in a 64-run tile, stored once, coalesced (the run still open at a tile's end is carried);
//   * the first pass leaves what the second needs in the run's scratch word: the codes of a short mismatch run (<= 5 bases), the micro-homology lengths of an indel;
//   * text is assembled in LDS, a tile of 64 runs at a time, in print order (tiles run towards the front of a reverse-strand string), and leaves as dwords.
// Chains whose walk has more than TXW_WCAP vertices or whose launch is "wide" (chromosome-scale) take k_text<NT> above.  Same bytes (tests/test_gpu_e2e.py, every e2e compare).
#define TXW_WCAP 32
#define TXW_CGB 768    // a tile's cg:Z pieces: 64 x (<= 10 digits + operator)
#define TXW_DSB 3072   // a tile's ds:Z entries; a tile that needs more writes them straight to the pool

typedef const __attribute__((address_space(1))) char *txw_gp;          // a byte in global memory (a pointer read from a table would otherwise be a flat access)
typedef __attribute__((address_space(3))) char *txw_lp;                // a byte in LDS
typedef const __attribute__((address_space(3))) unsigned char *txw_ltab;
struct txw_t {
	const __attribute__((address_space(3))) int32_t *vwb; const __attribute__((address_space(3))) txw_gp *ptr; int32_t cnt; txw_ltab nt4, comp;   // LDS
	const int32_t *g_vwb; const uint32_t *g_vert; const char *g_fw, *g_rc; const int64_t *g_off; int32_t ss;                                      // the same walk in global memory (cnt > TXW_WCAP)
};
struct txw_memo_t { int32_t lo, hi; txw_gp p; };

__device__ __forceinline__ unsigned char txw_get(const txw_t &T, txw_memo_t &m, int32_t x)
{
	if (x < m.lo || x >= m.hi) { // another vertex: the last one whose start is <= x
		if (T.cnt <= TXW_WCAP) {
			int32_t k = 0;
#pragma unroll
			for (int32_t step = TXW_WCAP / 2; step > 0; step >>= 1) if (k + step < T.cnt && T.vwb[k + step] <= x) k += step;
			m.lo = T.vwb[k], m.hi = k + 1 < T.cnt ? T.vwb[k + 1] : 0x7fffffff, m.p = T.ptr[k];
		} else { // a walk of many vertices (a read across dozens of bubbles): the table stays in global memory, the lane's memo does the rest
			int32_t lo = 0, hi = T.cnt - 1;
			while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (T.g_vwb[mid] <= x) lo = mid; else hi = mid - 1; }
			const uint32_t v = T.g_vert[lo];
			m.lo = T.g_vwb[lo], m.hi = lo + 1 < T.cnt ? T.g_vwb[lo + 1] : 0x7fffffff;
			m.p = (txw_gp)(((v & 1) ? T.g_rc : T.g_fw) + T.g_off[v >> 1] + (lo > 0 ? 0 : T.ss) - m.lo);
		}
	}
	return (unsigned char)m.p[x];
}

// where a run's bytes go: the tile's staging block in LDS, or -- a tile of more than TXW_DSB bytes -- the pool itself
struct txw_out_lds { txw_lp b; __device__ __forceinline__ void put(int32_t i, char c) const { b[i] = c; } __device__ __forceinline__ char get(int32_t i) const { return b[i]; } };
struct txw_out_glb { char *b; __device__ __forceinline__ void put(int32_t i, char c) const { b[i] = c; } __device__ __forceinline__ char get(int32_t i) const { return b[i]; } };
template<class OUT> __device__ __forceinline__ void txw_put_uint(const OUT &o, int32_t at, uint32_t x, int nd) { for (int i = nd - 1; i >= 0; --i) { o.put(at + i, (char)('0' + x % 10)); x /= 10; } }
__device__ __forceinline__ char txw_letter(uint32_t code) { return (char)(0x6e74676361ULL >> (8 * code)); } // "acgtn"[code]

// the entries of a match / mismatch run from the nt4 codes of its bases (galign.c:228-243): code(z) for z in [0, len); returns the bytes, WRITE: places them
template<bool WRITE, class OUT, class CODE> __device__ __forceinline__ int32_t txw_mm_entries(bool rev, int32_t len, const CODE &code, const OUT &o, int32_t at, int32_t n_total, txw_ltab comp)
{
	int32_t n = 0, l = 0;
#define TXW_AT(len_) (rev ? at + (n_total - n - (len_)) : at + n)
	for (int32_t z = 0; z <= len; ++z) {
		const bool last = z == len;
		uint32_t cx = 0, cy = 0;
		if (!last) { const uint32_t c = code(z); cx = c & 7u, cy = c >> 3; }
		if (last || cx != cy) {
			if (l > 0) { const int nd = txt_ndigits((uint32_t)l); if (WRITE) { const int32_t p = TXW_AT(1 + nd); o.put(p, ':'); txw_put_uint(o, p + 1, (uint32_t)l, nd); } n += 1 + nd; }
			if (!last) {
				if (WRITE) {
					const int32_t p = TXW_AT(3);
					char c1 = txw_letter(cx), c2 = txw_letter(cy);
					if (rev) c1 = (char)comp[(unsigned char)c1], c2 = (char)comp[(unsigned char)c2]; // format.c:225-226: same order, complemented
					o.put(p, '*'), o.put(p + 1, c1), o.put(p + 2, c2);
				}
				n += 3;
			}
			l = 0;
		} else ++l;
	}
	return n;
}

// ds:Z entries of one run.  First pass (WRITE = false): returns their length and leaves in *aux what the writing pass can reuse (-1: nothing).  WRITE: places them at o[at ...] (rev: from
// at + n_total backwards, each entry transformed), using *aux when it is there.
template<bool WRITE, class OUT> __device__ __forceinline__ int32_t txw_ds_run(const txw_t &T, txw_memo_t &M, bool rev, const char *__restrict__ q, int32_t op, int32_t len, int32_t x, int32_t y,
																			 int32_t qs, int32_t qe, int32_t apl, const OUT &o, int32_t at, int32_t n_total, int32_t *aux)
{
	int32_t n = 0;
	if (op == 7) {
		if (len > 0) { const int nd = txt_ndigits((uint32_t)len); if (WRITE) { const int32_t p = rev ? at + (n_total - 1 - nd) : at; o.put(p, ':'); txw_put_uint(o, p + 1, (uint32_t)len, nd); } n = 1 + nd; }
		if (!WRITE) *aux = -1;
	} else if (op == 0 || op == 8) {
		uint32_t pk = 0;
		const bool packed = len <= 5; // the codes travel in the scratch word: 6 bits per base
		if (packed) {
			if (!WRITE) {
				for (int32_t z = 0; z < len; ++z) pk |= ((uint32_t)T.nt4[txw_get(T, M, x + z)] | (uint32_t)T.nt4[(unsigned char)q[y + z]] << 3) << (6 * z);
				*aux = (int32_t)pk;
			} else pk = (uint32_t)*aux;
		} else if (!WRITE) *aux = -1;
		n = txw_mm_entries<WRITE>(rev, len, [&](int32_t z) { return packed ? pk >> (6 * z) & 63u : (uint32_t)T.nt4[txw_get(T, M, x + z)] | (uint32_t)T.nt4[(unsigned char)q[y + z]] << 3; }, o, at, n_total, T.comp);
	} else if (op == 1 || op == 2) { // micro-homology on either side (galign.c:229-249), then write_indel (galign.c:153-180)
		const bool ins = op == 1;
		int32_t z, ll, lr;
		if (WRITE && *aux >= 0) ll = *aux & 0x7fff, lr = *aux >> 15;
		else {
			if (ins) {
				for (z = 1; z <= len; ++z) if (y - z < qs || q[y + len - z] != q[y - z]) break;
				lr = z - 1;
				for (z = 0; z < len; ++z) if (y + len + z >= qe || q[y + len + z] != q[y + z]) break;
				ll = z;
			} else {
				for (z = 1; z <= len; ++z) if (x - z < 0 || txw_get(T, M, x + len - z) != txw_get(T, M, x - z)) break;
				lr = z - 1;
				for (z = 0; z < len; ++z) if (x + len + z >= apl || txw_get(T, M, x + z) != txw_get(T, M, x + len + z)) break;
				ll = z;
			}
			if (!WRITE) *aux = (ll < 0x8000 && lr < 0x8000) ? (ll | lr << 15) : -1;
		}
		int32_t m = 1 + len;
		if (ll + lr >= len) m += 2; else m += (ll > 0 ? 2 : 0) + (lr > 0 ? 2 : 0);
		if (WRITE) {
			// the entry in forward order: sign, then the bases with the homologous stretches at either end in brackets (all of them in one pair when the two meet)
			const int32_t p0 = rev ? at + (n_total - m) : at;
			const bool whole = ll + lr >= len;
			const int32_t b_l = whole ? len : ll, b_r = whole ? 0 : lr; // bracketed prefix / suffix
			int32_t p = p0;
			o.put(p++, ins ? '+' : '-');
			for (int32_t i = 0; i < len; ++i) {
				if (i == 0 && b_l > 0) o.put(p++, '[');
				if (b_r > 0 && i == len - b_r) o.put(p++, '[');
				o.put(p++, txw_letter(T.nt4[ins ? (unsigned char)q[y + i] : txw_get(T, M, x + i)]));
				if (b_l > 0 && i == b_l - 1) o.put(p++, ']');
				if (b_r > 0 && i == len - 1) o.put(p++, ']');
			}
			if (rev) { // everything after the sign: reversed, complemented, brackets swapped (format.c:229-237)
#define TXW_T(ch_) ((ch_) == '[' ? ']' : (ch_) == ']' ? '[' : (char)T.comp[(unsigned char)(ch_)])
				int32_t a = p0 + 1, b = p0 + m - 1;
				while (a < b) { const char ca = TXW_T(o.get(a)), cb = TXW_T(o.get(b)); o.put(a++, cb); o.put(b--, ca); }
				if (a == b) o.put(a, TXW_T(o.get(a)));
#undef TXW_T
			}
		}
		n = m;
	} else if (!WRITE) *aux = -1;
#undef TXW_AT
	return n;
}

__device__ __forceinline__ int32_t txw_excl_scan(int32_t v, int32_t *total)
{
	const int32_t incl = lc_scan_add(v, 0);
	*total = __builtin_amdgcn_readlane(incl, 63);
	return incl - v;
}
__device__ __forceinline__ int32_t txw_sum(int32_t v) { return __builtin_amdgcn_readlane(lc_scan_add(v, 0), 63); }

// a tile's staged bytes s[0 .. n) -> dst[0 .. n): dwords by the lanes, the tail bytewise
__device__ __forceinline__ void txw_flush(char *__restrict__ dst, const char *s, int32_t n, int lane)
{
	const int32_t n4 = n & ~3;
	for (int32_t b = lane * 4; b < n4; b += 256) { const uint32_t v = *(const uint32_t*)(s + b); __builtin_memcpy(dst + b, &v, 4); } // (the pool offset is a byte offset: an unaligned dword store)
	if (lane < n - n4) dst[n4 + lane] = s[n4 + lane];
}

__global__ void __launch_bounds__(64) k_text_w(int n_chain, const mga_txt_chain_t *__restrict__ chain, const mga_cigitem_t *__restrict__ item, const uint32_t *__restrict__ vert,
											   const int32_t *__restrict__ vwb, const char *__restrict__ gseq, const char *__restrict__ gseq_rc, const int64_t *__restrict__ gseq_off,
											   const char *__restrict__ reads, const int32_t *__restrict__ ncig, const int64_t *__restrict__ cigoff,
											   const uint32_t *__restrict__ ord, const int64_t *__restrict__ el_off, uint32_t *__restrict__ el, uint32_t *__restrict__ run,
											   int32_t *__restrict__ run_txt, mga_txt_res_t *__restrict__ res, char *__restrict__ pool, long long pool_cap, unsigned long long *pool_used)
{
	__shared__ int32_t s_o[64], s_op[64], s_val[64];
	__shared__ int64_t s_src[64];
	__shared__ int32_t s_vwb[TXW_WCAP];
	__shared__ txw_gp s_ptr[TXW_WCAP];
	__shared__ uint32_t s_run[65];
	__shared__ __attribute__((aligned(16))) unsigned char s_tab[512];
	__shared__ __attribute__((aligned(16))) char s_cg[TXW_CGB], s_ds[TXW_DSB];
	const int c = blockIdx.x, lane = threadIdx.x;
	if (c >= n_chain) return;
	const mga_txt_chain_t C = chain[c];
	const int64_t eo = el_off[c];
	const int32_t n_el = (int32_t)(el_off[c + 1] - eo);
	uint32_t *E = el + eo, *R = run + eo;
	int32_t *RT = run_txt + 2 * eo;      // per run: ds entries length, scratch word of the first pass
	const char *q = reads + C.q_base;
	// tables and the walk into LDS
	((uint32_t*)s_tab)[lane] = ((const uint32_t*)c_txt.nt4)[lane], ((uint32_t*)s_tab)[64 + lane] = ((const uint32_t*)c_txt.comp)[lane];
	if (lane < C.vert_cnt && C.vert_cnt <= TXW_WCAP) {
		const uint32_t v = vert[C.vert_beg + lane];
		const int32_t wb = vwb[C.vert_beg + lane];
		s_vwb[lane] = wb;
		s_ptr[lane] = (txw_gp)(((v & 1) ? gseq_rc : gseq) + gseq_off[v >> 1] + (lane > 0 ? 0 : C.ss) - wb); // base x of the walk is s_ptr[k][x] (gfa_edseq_init, gfa-ed.c:24-42: an oriented vertex is one string)
	}
	txw_t T;
	T.vwb = (const __attribute__((address_space(3))) int32_t*)s_vwb, T.ptr = (const __attribute__((address_space(3))) txw_gp*)s_ptr, T.cnt = C.vert_cnt;
	T.nt4 = (txw_ltab)s_tab, T.comp = (txw_ltab)s_tab + 256;
	T.g_vwb = vwb + C.vert_beg, T.g_vert = vert + C.vert_beg, T.g_fw = gseq, T.g_rc = gseq_rc, T.g_off = gseq_off, T.ss = C.ss;
	txw_memo_t M; M.lo = 0, M.hi = -1, M.p = 0;
	__syncthreads();

	// ---- S1: concatenated operator list (as in k_text)
	{
		int32_t base = 0;
		for (int64_t tb = C.item_beg; tb < C.item_end; tb += 64) {
			const int64_t t = tb + lane;
			mga_cigitem_t it; it.op = 0, it.val = 0;
			int32_t cnt = 0;
			int64_t src = 0;
			if (t < C.item_end) {
				it = item[t];
				if (it.op >= 0) cnt = 1; else { const int64_t pj = C.prob_base + it.val; cnt = ncig[pj]; src = cigoff[pj]; }
			}
			int32_t tot;
			const int32_t o_l = txw_excl_scan(cnt, &tot);
			__syncthreads();
			s_o[lane] = o_l, s_op[lane] = it.op, s_val[lane] = it.val, s_src[lane] = src;
			__syncthreads();
			for (int32_t o = lane; o < tot; o += 64) {
				int lo = 0;
#pragma unroll
				for (int step = 32; step > 0; step >>= 1) if (s_o[lo + step] <= o) lo += step;
				const int32_t k = o - s_o[lo], op_l = s_op[lo];
				uint32_t e;
				if (op_l >= 0) e = (uint32_t)s_val[lo] << 5 | 1u << 4 | (uint32_t)op_l;
				else { const uint32_t cg = ord[s_src[lo] + k]; e = (cg >> 4) << 5 | (k == 0 ? 1u << 4 : 0u) | (cg & 0xf); }
				E[base + o] = e;
			}
			base += tot;
		}
	}
	__threadfence_block();
	__syncthreads();
	// ---- S2: runs (append_cigar1 / append_cigar, galign.c:11-37): lengths add up in LDS, a tile of 64 elements at a time; slot 0 is the run the previous tile left open
	int32_t n_run = 0;
	{
		uint32_t carry = 0;
		for (int32_t b0 = 0; b0 < n_el; b0 += 64) {
			const int32_t i = b0 + lane;
			const uint32_t e = i < n_el ? E[i] : 0;
			const int32_t op = (int32_t)(e & 0xf);
			const int32_t prev = i > 0 && i < n_el ? (int32_t)(E[i - 1] & 0xf) : -1;
			const bool head = i < n_el && !((e >> 4 & 1) && op == prev);
			int32_t tot;
			const int32_t before = txw_excl_scan(head ? 1 : 0, &tot);
			s_run[lane + 1] = 0;
			if (lane == 0) s_run[0] = carry;
			__syncthreads();
			if (i < n_el) atomicAdd(&s_run[before + (head ? 1 : 0)], (e >> 5) << 5 | (head ? (uint32_t)op : 0u)); // lengths add up; the head contributes the operator bits
			__syncthreads();
			// runs [n_run - 1, n_run + tot) now have their values so far (slot 0 only exists when a run was open)
			if (lane < tot) R[n_run + lane] = s_run[lane + 1];
			if (lane == 0 && n_run > 0) R[n_run - 1] = s_run[0];
			carry = s_run[tot];
			n_run += tot;
			__syncthreads();
		}
	}
	__threadfence_block();
	__syncthreads();
	// ---- S3: coordinates, statistics, text lengths per run; S4: the text, a tile at a time through LDS
	const int32_t apl = C.pe - C.ps;
	int32_t mlen = 0, blen = 0, aplen = 0, qlen = 0, cg_n = 0, ds_n = 0;
	{
		int32_t x0 = 0, y0 = C.qs;
		for (int32_t b0 = 0; b0 < n_run; b0 += 64) {
			const int32_t r = b0 + lane;
			const uint32_t e = r < n_run ? R[r] : 0;
			const int32_t op = (int32_t)(e & 0xf), len = (int32_t)(e >> 5);
			const int32_t dx = r < n_run && op != 1 ? len : 0, dy = r < n_run && op != 2 ? len : 0;
			int32_t tx, ty;
			const int32_t x = x0 + txw_excl_scan(dx, &tx), y = y0 + txw_excl_scan(dy, &ty);
			int32_t lc = 0, ld = 0, aux = -1;
			if (r < n_run) {
				lc = txt_ndigits((uint32_t)len) + 1;
				ld = txw_ds_run<false>(T, M, false, q, op, len, x, y, C.qs, C.qe, apl, txw_out_glb{0}, 0, 0, &aux);
				RT[2 * r] = ld, RT[2 * r + 1] = aux;
			}
			mlen += txw_sum(r < n_run && op == 7 ? len : 0), blen += txw_sum(r < n_run ? len : 0);
			aplen += tx, qlen += ty, cg_n += txw_sum(lc), ds_n += txw_sum(ld);
			x0 += tx, y0 += ty;
		}
	}
	unsigned long long o_pool = 0;
	{
		const unsigned long long need = (unsigned long long)cg_n + (unsigned long long)ds_n;
		if (lane == 0) o_pool = atomicAdd(pool_used, need);
		o_pool = (unsigned long long)__shfl((long long)o_pool, 0);
		mga_txt_res_t rr;
		rr.txt_off = (int64_t)o_pool, rr.cg_len = cg_n, rr.ds_len = ds_n, rr.n_cigar = n_run, rr.mlen = mlen, rr.blen = blen, rr.aplen = aplen, rr.pad = 0;
		rr.status = (qlen == C.qe - C.qs && aplen == apl) ? 0 : 1; // galign.c:140
		if (rr.status == 0 && (long long)(o_pool + need) > pool_cap) rr.status = 2;
		if (lane == 0) res[c] = rr;
		if (rr.status != 0) return;
	}
	__threadfence_block();
	__syncthreads();
	{
		char *const cg_base = pool + o_pool, *const ds_base = pool + o_pool + cg_n;
		int32_t x0 = 0, y0 = C.qs, cg0 = 0, ds0 = 0;
		const bool rev = C.rev_sign != 0;
		for (int32_t b0 = 0; b0 < n_run; b0 += 64) {
			const int32_t r = b0 + lane;
			const uint32_t e = r < n_run ? R[r] : 0;
			const int32_t op = (int32_t)(e & 0xf), len = (int32_t)(e >> 5);
			const int32_t dx = r < n_run && op != 1 ? len : 0, dy = r < n_run && op != 2 ? len : 0;
			int32_t tx, ty, tc, td;
			const int32_t x = x0 + txw_excl_scan(dx, &tx), y = y0 + txw_excl_scan(dy, &ty);
			const int32_t lc = r < n_run ? txt_ndigits((uint32_t)len) + 1 : 0, ld = r < n_run ? RT[2 * r] : 0;
			int32_t aux = r < n_run ? RT[2 * r + 1] : -1;
			const int32_t oc = txw_excl_scan(lc, &tc), od = txw_excl_scan(ld, &td); // offsets inside the tile
			// where the tile goes in the strings: forward lines grow from the front, reverse-strand lines from the back; inside the tile the same rule
			char *const cg_dst = rev ? cg_base + (cg_n - cg0 - tc) : cg_base + cg0, *const ds_dst = rev ? ds_base + (ds_n - ds0 - td) : ds_base + ds0;
			const bool ds_lds = td <= TXW_DSB; // (uniform)
			if (r < n_run) {
				const txw_out_lds oc_l{(txw_lp)s_cg};
				const int32_t pc = rev ? tc - oc - lc : oc; // cg:Z piece: "<len><op>" (format.c:205-215)
				txw_put_uint(oc_l, pc, (uint32_t)len, lc - 1);
				oc_l.put(pc + lc - 1, "MIDNSHP=XB"[op]);
				const int32_t pd = rev ? td - od - ld : od;
				if (ds_lds) txw_ds_run<true>(T, M, rev, q, op, len, x, y, C.qs, C.qe, apl, txw_out_lds{(txw_lp)s_ds}, pd, ld, &aux);
				else txw_ds_run<true>(T, M, rev, q, op, len, x, y, C.qs, C.qe, apl, txw_out_glb{ds_dst}, pd, ld, &aux);
			}
			__syncthreads();
			txw_flush(cg_dst, s_cg, tc, lane);
			if (ds_lds) txw_flush(ds_dst, s_ds, td, lane);
			__syncthreads();
			cg0 += tc, ds0 += td, x0 += tx, y0 += ty;
		}
	}
}

extern "C" int mga_dev_text_tables(const unsigned char *comp, const unsigned char *nt4)
{
	txt_tables_t h;
	memcpy(h.comp, comp, 256); memcpy(h.nt4, nt4, 256);
	MGA_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_txt), &h, sizeof h));
	return 0;
}

// scratch (grow-only, in the stream context): per chain element counts/offsets, per vertex walk offsets, per element 16 bytes
extern "C" int mga_dev_text(mga_sctx_t *sc, int n_chain, const mga_txt_chain_t *d_chain, const mga_cigitem_t *d_item, int64_t n_vert, const uint32_t *d_vert,
							const mga_didx_t *ix, const char *d_reads, int64_t n_el_max, const int32_t *d_ncig, const int64_t *d_cigoff, const uint32_t *d_ord,
							mga_txt_res_t *d_res, char *d_pool, int64_t pool_cap, unsigned long long *d_pool_used)
{
	if (n_chain <= 0) return 0;
	if (ix->d_gseq == 0) { mga_set_error("text kernel: the index holds no device copy of the graph sequences"); return -1; }
	hipStream_t st = (hipStream_t)sc->stream;
	if (mga_dbuf_reserve(&sc->txt_cnt, (size_t)(n_chain + 1) * 4) < 0 || mga_dbuf_reserve(&sc->txt_off, (size_t)(n_chain + 2) * 8) < 0 ||
		mga_dbuf_reserve(&sc->txt_vwb, (size_t)(n_vert + 1) * 4) < 0 || mga_dbuf_reserve(&sc->txt_el, (size_t)(n_el_max + 64) * 16) < 0) return -1;
	uint32_t *el = (uint32_t*)sc->txt_el.p, *run = el + (n_el_max + 64);
	int32_t *run_txt = (int32_t*)(run + (n_el_max + 64));
	// chromosome-scale chains (a handful per launch, 10^5+ operators each) get a workgroup each in both passes -- sixteen wavefronts in the writing pass --, ordinary reads' chains a
	// lane in the counting pass and one wavefront in the writing pass
	const bool wide = n_el_max / n_chain >= 32768;
	mga_prof_begin(sc->stream, MGA_K_TEXT);
	if (wide) hipLaunchKernelGGL(k_text_count_wg, dim3(n_chain), dim3(256), 0, st, n_chain, d_chain, d_item, d_vert, (const int32_t*)ix->d_seg_len, d_ncig, (int32_t*)sc->txt_cnt.p, (int32_t*)sc->txt_vwb.p);
	else hipLaunchKernelGGL(k_text_count, dim3((n_chain + 63) / 64), dim3(64), 0, st, n_chain, d_chain, d_item, d_vert, (const int32_t*)ix->d_seg_len, d_ncig,
					   (int32_t*)sc->txt_cnt.p, (int32_t*)sc->txt_vwb.p);
	mga_prof_end(sc->stream, MGA_K_TEXT);
	MGA_HIP_CHECK(hipGetLastError());
	if (mga_dev_scan_i32_to_i64(sc, (const int32_t*)sc->txt_cnt.p, n_chain, (int64_t*)sc->txt_off.p) < 0) return -1;
	mga_prof_begin(sc->stream, MGA_K_TEXT);
#define TXT_LAUNCH(NT_) hipLaunchKernelGGL((k_text<NT_>), dim3(n_chain), dim3(NT_), 0, st, n_chain, d_chain, d_item, d_vert, (const int32_t*)sc->txt_vwb.p, (const char*)ix->d_gseq, \
					   (const int64_t*)ix->d_gseq_off, (const int32_t*)ix->d_seg_len, d_reads, d_ncig, d_cigoff, d_ord, (const int64_t*)sc->txt_off.p, \
					   el, run, run_txt, d_res, d_pool, (long long)pool_cap, d_pool_used)
	// MGA_TEXT_W=0: the general kernel for every launch (A/B, tests)
	static int use_w = -1;
	if (use_w < 0) { const char *e = getenv("MGA_TEXT_W"); use_w = e && *e ? atoi(e) : 1; }
	if (wide) TXT_LAUNCH(1024);
	else if (use_w && ix->d_gseq_rc != 0)
		hipLaunchKernelGGL(k_text_w, dim3(n_chain), dim3(64), 0, st, n_chain, d_chain, d_item, d_vert, (const int32_t*)sc->txt_vwb.p, (const char*)ix->d_gseq, (const char*)ix->d_gseq_rc,
						   (const int64_t*)ix->d_gseq_off, d_reads, d_ncig, d_cigoff, d_ord, (const int64_t*)sc->txt_off.p, el, run, run_txt, d_res, d_pool, (long long)pool_cap, d_pool_used);
	else TXT_LAUNCH(64);
#undef TXT_LAUNCH
	mga_prof_end(sc->stream, MGA_K_TEXT);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
