// k_gaf.hip -- whole GAF lines on the device (round 6): mg_write_gaf (reference format.c:121-250) for the path that only wants GAF bytes.
//
// k_text.hip leaves cg:Z / ds:Z of every printed chain in the text pool; until round 5 the host then formatted the twelve columns and the numeric tags of every line and
// copied the two strings behind them -- twice over the chains (a measuring pass, a writing pass) and, at a rank's share of the cores, a third of its CPU time.  Here one
// wavefront per LINE does it: the read's name, the coordinates, the path column -- the walk folded into intervals of stable sequences exactly like format.c:140-199, the
// one-interval "compact" form with its strand in column 5 included --, mlen / blen from the text kernel's own counts, the tags, dv:f: with printf's "%.4f" rounding, and the
// two strings copied behind them.  The host supplies what only it knows -- per line: the read, the chain's coordinates, mapq / score / div (their log / logf stay on the
// host: gcmisc.c) -- and receives the chunk's lines in read order, back to back: a counting pass sizes every line, a device scan places them, the writing pass runs the SAME
// formatter with stores (the two cannot disagree: one template).
//
// The formatter is scalar code run by the whole wavefront (every value uniform): lane 0 stores bytes into an LDS stage, all lanes flush the stage to global memory when it
// fills and copy names / alignment strings cooperatively.  A line's header is ~150 bytes next to ~9 000 of cg + ds for a 10 kb read: the kernel is the copy.
#include "mga_dev.h"
#include "dev_common.h"
#include <string.h>
#include <stdlib.h>

struct gaf_seg_t { int64_t name_off; int32_t name_len, snid, soff, len; };   // per segment (gfa_seg_t: name, snid, soff, len)
struct gaf_sseq_t { int64_t name_off; int32_t name_len, min, max, rank; };   // per stable sequence (gfa_sseq_t)

#define GAF_STAGE 1024 // bytes of the LDS stage; a token (a number, a tag) is at most 16 bytes, longer strings are copied in pieces

template<bool WRITE> struct gaf_out_t {
	char *stage;      // LDS
	char *dst;        // where the next flush goes
	int32_t n;        // bytes in the stage
	int64_t total;    // bytes of the line so far (flushed + staged)
	int lane;
	__device__ __forceinline__ void flush()
	{
		if (WRITE) {
			mga_wave_sync();
			for (int32_t b = lane; b < n; b += 64) dst[b] = stage[b];
			mga_wave_sync();
			dst += n;
		}
		n = 0;
	}
	__device__ __forceinline__ void room(int32_t need) { if (n + need > GAF_STAGE) flush(); }
	__device__ __forceinline__ void c(char ch) { room(1); if (WRITE && lane == 0) stage[n] = ch; ++n, ++total; }
	__device__ __forceinline__ void lit(const char *s, int32_t len) { room(len); if (WRITE && lane == 0) for (int32_t i = 0; i < len; ++i) stage[n + i] = s[i]; n += len, total += len; }
	__device__ __forceinline__ void u(uint32_t x)
	{
		int32_t nd = 1;
		for (uint32_t y = x; y >= 10; y /= 10) ++nd;
		room(nd);
		if (WRITE && lane == 0) { uint32_t y = x; for (int32_t i = nd - 1; i >= 0; --i) { stage[n + i] = (char)('0' + y % 10); y /= 10; } }
		n += nd, total += nd;
	}
	__device__ __forceinline__ void d(int32_t x) { if (x < 0) { c('-'); u((uint32_t)-(int64_t)x); } else u((uint32_t)x); }
	__device__ __forceinline__ void tab_d(int32_t x) { c('\t'); d(x); }
	// len bytes of global memory behind the staged ones (names, alignment strings): all lanes copy
	__device__ __forceinline__ void copy(const char *__restrict__ src, int32_t len)
	{
		if (len <= 0) return;
		if (!WRITE) { total += len; return; }
		if (len <= 64) { // short: through the stage, one byte per lane
			room(len);
			if (lane < len) stage[n + lane] = src[lane];
			n += len, total += len;
			return;
		}
		flush();
		// dwords where source and destination are both reachable as (possibly unaligned) dwords; the tail bytewise
		const int32_t n4 = len & ~3;
		for (int32_t b = lane * 4; b < n4; b += 256) { uint32_t v; __builtin_memcpy(&v, src + b, 4); __builtin_memcpy(dst + b, &v, 4); }
		if (lane < len - n4) dst[n4 + lane] = src[n4 + lane];
		dst += len, total += len;
	}
};

// "%.4f" of a float in [0, 1], "0" for exactly zero (format.c:200-203).  A float times 10^4 is exact in a double (24 + 14 bits), so rounding THAT to an integer half-to-even
// is printf's rounding of the exact decimal expansion (tests/test_gpu_stages.py: test_gaf_div_text, against snprintf)
template<bool WRITE> __device__ __forceinline__ void gaf_put_div(gaf_out_t<WRITE> &O, float div)
{
	if (div == 0.0f) { O.c('0'); return; }
	const uint32_t r = (uint32_t)rint((double)div * 10000.0);
	O.u(r / 10000u); O.c('.');
	const uint32_t f = r % 10000u;
	O.c((char)('0' + f / 1000u)); O.c((char)('0' + f / 100u % 10u)); O.c((char)('0' + f / 10u % 10u)); O.c((char)('0' + f % 10u));
}

struct gaf_tabs_t { const gaf_seg_t *seg; const gaf_sseq_t *sseq; const char *names; };

// the path column's pieces (format.c:140-199): a vertex printed by name, or a maximal run of vertices that continue each other on one stable sequence, printed as an interval of it.
// EMIT = false: only counts the pieces and remembers the first one (is the line of the compact form?)
template<bool WRITE, bool EMIT>
__device__ __forceinline__ int32_t gaf_fold(gaf_out_t<WRITE> &O, const gaf_tabs_t &G, const uint32_t *__restrict__ vert, int32_t cnt, bool by_name, int32_t *first_snid, int lane)
{
	int32_t n_piece = 0;
	bool open = false;
	int32_t c_snid = -1, c_rev = 0, c_st = 0, c_en = 0;
	*first_snid = -1;
	auto close = [&]() {
		if (!open) return;
		if (EMIT) {
			const gaf_sseq_t sq = G.sseq[c_snid];
			O.c(c_rev ? '<' : '>');
			O.copy(G.names + sq.name_off, sq.name_len);
			O.c(':'); O.d(c_st); O.c('-'); O.d(c_en);
		}
		open = false;
	};
	for (int32_t j0 = 0; j0 < cnt; j0 += 64) {
		const int32_t j = j0 + lane;
		uint32_t vv = 0;
		gaf_seg_t sg; sg.name_off = 0, sg.name_len = 0, sg.snid = -1, sg.soff = 0, sg.len = 0;
		if (j < cnt) { vv = vert[j]; sg = G.seg[vv >> 1]; }
		const int32_t nb = cnt - j0 < 64 ? cnt - j0 : 64;
		for (int32_t l = 0; l < nb; ++l) {
			const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int32_t)vv, l);
			const int32_t snid = __builtin_amdgcn_readlane(sg.snid, l), soff = __builtin_amdgcn_readlane(sg.soff, l), len = __builtin_amdgcn_readlane(sg.len, l);
			const int32_t rev = (int32_t)(v & 1);
			if (by_name || snid < 0) {
				close();
				if (n_piece == 0) *first_snid = -1;
				++n_piece;
				if (EMIT) {
					const int64_t no = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)sg.name_off, l) | (int64_t)__builtin_amdgcn_readlane((int32_t)(sg.name_off >> 32), l) << 32;
					O.c(rev ? '<' : '>');
					O.copy(G.names + no, __builtin_amdgcn_readlane(sg.name_len, l));
				}
				continue;
			}
			// does this vertex continue the open interval?  forward: it starts where the interval ends; reverse: it ends where the interval starts
			if (open && c_snid == snid && c_rev == rev && (rev ? soff + len == c_st : soff == c_en)) {
				if (rev) c_st = soff; else c_en = soff + len;
				continue;
			}
			close();
			if (n_piece == 0) *first_snid = snid;
			++n_piece;
			open = true, c_snid = snid, c_rev = rev, c_st = soff, c_en = soff + len;
		}
	}
	close();
	return n_piece;
}

template<bool WRITE>
__global__ void __launch_bounds__(64) k_gaf(int n_lines, const mga_gaf_line_t *__restrict__ line, const char *__restrict__ qnames, const int64_t *__restrict__ qname_off,
											gaf_tabs_t G, uint64_t flag, const mga_txt_chain_t *__restrict__ chain, const uint32_t *__restrict__ vert,
											const mga_txt_res_t *__restrict__ tres, const char *__restrict__ tpool,
											int32_t *__restrict__ line_len, const int64_t *__restrict__ line_off, char *__restrict__ out)
{
	__shared__ char stage[GAF_STAGE];
	const int li = blockIdx.x, lane = threadIdx.x;
	if (li >= n_lines) return;
	const mga_gaf_line_t L = line[li];
	gaf_out_t<WRITE> O;
	O.stage = stage, O.dst = WRITE ? out + line_off[li] : 0, O.n = 0, O.total = 0, O.lane = lane;
	{
		const int64_t q0 = qname_off[L.read];
		O.copy(qnames + q0, (int32_t)(qname_off[L.read + 1] - q0));
	}
	O.tab_d(L.qlen);
	if (L.chain < 0) { // an unmapped read's line (MG_M_SHOW_UNMAP, format.c:130-133)
		O.lit("\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0\n", 21);
	} else {
		const mga_txt_chain_t C = chain[L.chain];
		const mga_txt_res_t T = tres[L.chain];
		const uint32_t *V = vert + C.vert_beg;
		const bool by_name = (flag & MG_M_VERTEX_COOR) != 0;
		int32_t first_snid;
		const int32_t n_piece = gaf_fold<WRITE, false>(O, G, V, C.vert_cnt, by_name, &first_snid, lane);
		bool compact = false;
		if (!(flag & (MG_M_VERTEX_COOR | MG_M_NO_COMP_PATH)) && n_piece == 1 && first_snid >= 0) { const gaf_sseq_t sq = G.sseq[first_snid]; compact = sq.rank == 0 && sq.min == 0; }
		const int32_t first_rev = (int32_t)(V[0] & 1);
		O.tab_d(L.qs); O.tab_d(L.qe);
		O.c('\t'); O.c(compact && first_rev ? '-' : '+'); O.c('\t');
		if (compact) { // the stable sequence, its length, the interval of the alignment on it
			const gaf_sseq_t sq = G.sseq[first_snid];
			const gaf_seg_t t = G.seg[V[first_rev ? C.vert_cnt - 1 : 0] >> 1]; // the segment the path's coordinates count from
			const int32_t beg = first_rev ? L.plen - L.pe : L.ps, end = first_rev ? L.plen - L.ps : L.pe;
			O.copy(G.names + sq.name_off, sq.name_len);
			O.tab_d(sq.max); O.tab_d(t.soff + beg); O.tab_d(t.soff + end);
		} else {
			int32_t dummy;
			(void)gaf_fold<WRITE, true>(O, G, V, C.vert_cnt, by_name, &dummy, lane);
			O.tab_d(L.plen); O.tab_d(L.ps); O.tab_d(L.pe);
		}
		O.tab_d(T.mlen); O.tab_d(T.blen); O.tab_d(L.mapq);
		O.lit("\ttp:A:", 6); O.c(L.primary ? 'P' : 'S');
		O.lit("\tNM:i:", 6); O.d(T.blen - T.mlen);
		O.lit("\tcm:i:", 6); O.d(L.n_anchor);
		O.lit("\ts1:i:", 6); O.d(L.score);
		O.lit("\ts2:i:", 6); O.d(L.subsc);
		if (L.div >= 0.0f && L.div <= 1.0f) { // "%.4f", "0" for exactly zero (format.c:200-203): a float times 10^4 is exact in a double, so rounding it to an integer
			O.lit("\tdv:f:", 6);                 // half-to-even IS printf's rounding of the exact decimal expansion
			gaf_put_div(O, L.div);
		}
		O.lit("\tcg:Z:", 6); O.copy(tpool + T.txt_off, T.cg_len);
		O.lit("\tds:Z:", 6); O.copy(tpool + T.txt_off + T.cg_len, T.ds_len);
		O.c('\n');
	}
	O.flush();
	if (!WRITE && lane == 0) line_len[li] = (int32_t)(O.total > 0x7fffffff ? 0x7fffffff : O.total);
}

// stage test: dv:f: of n values, 8 bytes each (NUL-padded)
__global__ void __launch_bounds__(64) k_gaf_div(int n, const float *__restrict__ div, char *__restrict__ out)
{
	__shared__ char stage[GAF_STAGE];
	const int i = blockIdx.x, lane = threadIdx.x;
	if (i >= n) return;
	if (lane < 8) out[8 * (int64_t)i + lane] = 0;
	mga_wave_sync();
	gaf_out_t<true> O;
	O.stage = stage, O.dst = out + 8 * (int64_t)i, O.n = 0, O.total = 0, O.lane = lane;
	gaf_put_div(O, div[i]);
	O.flush();
}
extern "C" int mga_dev_gaf_div(mga_sctx_t *sc, int n, const float *d_div, char *d_out)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_gaf_div, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_div, d_out);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

// names of the segments and the stable sequences -> HBM (once per index; mga_dev_graph_upload)
extern "C" int mga_dev_gaf_names_upload(const gfa_t *g, mga_didx_t *ix)
{
	size_t tot = 0;
	uint32_t i;
	for (i = 0; i < g->n_seg; ++i) tot += strlen(g->seg[i].name);
	for (i = 0; i < g->n_sseq; ++i) tot += strlen(g->sseq[i].name);
	gaf_seg_t *hs = (gaf_seg_t*)malloc(((size_t)g->n_seg + 1) * sizeof(gaf_seg_t));
	gaf_sseq_t *hq = (gaf_sseq_t*)malloc(((size_t)g->n_sseq + 1) * sizeof(gaf_sseq_t));
	char *hn = (char*)malloc(tot + 64);
	if (hs == 0 || hq == 0 || hn == 0) { free(hs); free(hq); free(hn); mga_set_error("gaf names: out of memory"); return -1; }
	size_t at = 0;
	for (i = 0; i < g->n_seg; ++i) {
		const gfa_seg_t *s = &g->seg[i];
		const size_t l = strlen(s->name);
		hs[i].name_off = (int64_t)at, hs[i].name_len = (int32_t)l, hs[i].snid = s->snid, hs[i].soff = s->soff, hs[i].len = (int32_t)s->len;
		memcpy(hn + at, s->name, l); at += l;
	}
	for (i = 0; i < g->n_sseq; ++i) {
		const gfa_sseq_t *s = &g->sseq[i];
		const size_t l = strlen(s->name);
		hq[i].name_off = (int64_t)at, hq[i].name_len = (int32_t)l, hq[i].min = s->min, hq[i].max = s->max, hq[i].rank = s->rank;
		memcpy(hn + at, s->name, l); at += l;
	}
	memset(hn + at, 0, 64);
	ix->d_gaf_seg = mga_dmalloc(((size_t)g->n_seg + 1) * sizeof(gaf_seg_t));
	ix->d_gaf_sseq = mga_dmalloc(((size_t)g->n_sseq + 1) * sizeof(gaf_sseq_t));
	ix->d_gaf_names = (char*)mga_dmalloc(tot + 64);
	int rc = 0;
	if (ix->d_gaf_seg == 0 || ix->d_gaf_sseq == 0 || ix->d_gaf_names == 0 ||
		mga_h2d(ix->d_gaf_seg, hs, (size_t)g->n_seg * sizeof(gaf_seg_t)) < 0 || mga_h2d(ix->d_gaf_sseq, hq, (size_t)g->n_sseq * sizeof(gaf_sseq_t)) < 0 ||
		mga_h2d(ix->d_gaf_names, hn, tot + 64) < 0) {
		mga_dfree(ix->d_gaf_seg); mga_dfree(ix->d_gaf_sseq); mga_dfree(ix->d_gaf_names);
		ix->d_gaf_seg = ix->d_gaf_sseq = 0, ix->d_gaf_names = 0;
		rc = -1;
	}
	free(hs); free(hq); free(hn);
	return rc;
}

// pass 1 (d_out == NULL): d_len[i] = bytes of line i.  pass 2: the lines at d_out + d_off[i].
extern "C" int mga_dev_gaf(mga_sctx_t *sc, const mga_didx_t *ix, int n_lines, const mga_gaf_line_t *d_line, const char *d_qnames, const int64_t *d_qname_off, uint64_t flag,
						   const mga_txt_chain_t *d_chain, const uint32_t *d_vert, const mga_txt_res_t *d_tres, const char *d_tpool,
						   int32_t *d_len, const int64_t *d_off, char *d_out)
{
	if (n_lines <= 0) return 0;
	if (ix->d_gaf_seg == 0) { mga_set_error("gaf kernel: the index holds no device copy of the graph's names"); return -1; }
	gaf_tabs_t G;
	G.seg = (const gaf_seg_t*)ix->d_gaf_seg, G.sseq = (const gaf_sseq_t*)ix->d_gaf_sseq, G.names = ix->d_gaf_names;
	hipStream_t st = (hipStream_t)sc->stream;
	mga_prof_begin(sc->stream, MGA_K_GAF);
	if (d_out == 0) hipLaunchKernelGGL(k_gaf<false>, dim3(n_lines), dim3(64), 0, st, n_lines, d_line, d_qnames, d_qname_off, G, flag, d_chain, d_vert, d_tres, d_tpool, d_len, d_off, d_out);
	else hipLaunchKernelGGL(k_gaf<true>, dim3(n_lines), dim3(64), 0, st, n_lines, d_line, d_qnames, d_qname_off, G, flag, d_chain, d_vert, d_tres, d_tpool, d_len, d_off, d_out);
	mga_prof_end(sc->stream, MGA_K_GAF);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
