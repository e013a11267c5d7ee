// k_wfa.hip -- exact 2-piece affine-gap wavefront alignment (WFA) with traceback, one wavefront
// (64 lanes) per problem, lanes = diagonals.
//
// Replaces mwf_wfa_auto()'s exact mode as mg_gchain_cigar() calls it once per anchor gap
// (reference galign.c:102-107 -> miniwfa.c:824-828 -> :603-615 -> mwf_wfa_core :380-435):
// penalties x / o1,e1 / o2,e2, global alignment, max_iter = 1e8 wavefront cells, CIGAR out.
// Semantics that must match bit for bit (they decide ties in the CIGAR):
//   - recurrence and the traceback byte of every cell                         miniwfa.c:281-308
//   - reads outside a slice's [lo,hi] and from negative scores are NEG_INF    miniwfa.c:79-119
//   - band growth wf->lo/hi by at most one per score, only when the edge cell is reachable
//     (>= -1)                                                                 miniwfa.c:323-324,412-413
//   - dead-diagonal trimming every 256 scores over the 17-slice ring          miniwfa.c:139-169,420
//   - extension along exact matches on RAW bytes (N matches N)                miniwfa.c:212-226,399-411
//   - last_state at termination and the traceback walk                        miniwfa.c:405-408,329-377
//
// Data layout per wavefront (workspace in HBM, persistent waves pull problems from an atomic queue):
//   ring  int32 [17 slices][5 arrays H,E1,F1,E2,F2][WMAX]   slice of score s lives in slot s % 17
//   tb    uint8 [TBCAP]      one byte per wavefront cell, rows appended score by score
//   row   int64 [SMAX+1], rlo int32 [SMAX+1]                 start / lowest diagonal of each tb row
//   cig   uint32 [CIGCAP]    CIGAR built back-to-front during traceback
// Three capacity tiers (WMAX/TBCAP/SMAX/CIGCAP); a problem that does not fit is handed to the
// next tier by the host driver (status MGA_WFA_RETRY_TIER).  Nothing falls back to the CPU.
#include "mga_dev.h"
#include "dev_common.h"

#define WF_NEG_INF (-0x40000000)
#define WF_NSLOT 17

struct wfa_cfg_t {
	int32_t x, o1, e1, o2, e2;
	int32_t wmax, smax, cigcap;
	int64_t tbcap, max_iter;
	int64_t ws_stride; // bytes of workspace per wave
};

struct wfa_ws_t { // carved out of the per-wave workspace
	int32_t *ring; uint8_t *tb; int64_t *row; int32_t *rlo; uint32_t *cig;
};

__device__ __forceinline__ wfa_ws_t wfa_carve(char *base, const wfa_cfg_t &c)
{
	wfa_ws_t w;
	size_t o = 0;
	w.ring = (int32_t*)(base + o); o += (size_t)WF_NSLOT * 5 * c.wmax * 4;
	w.row  = (int64_t*)(base + o); o += (size_t)(c.smax + 1) * 8;
	w.rlo  = (int32_t*)(base + o); o += (size_t)(c.smax + 1) * 4; o = (o + 15) & ~(size_t)15;
	w.cig  = (uint32_t*)(base + o); o += (size_t)c.cigcap * 4;
	w.tb   = (uint8_t*)(base + o);
	return w;
}

static size_t wfa_ws_bytes(const wfa_cfg_t &c)
{
	size_t o = 0;
	o += (size_t)WF_NSLOT * 5 * c.wmax * 4;
	o += (size_t)(c.smax + 1) * 8;
	o += (size_t)(c.smax + 1) * 4; o = (o + 15) & ~(size_t)15;
	o += (size_t)c.cigcap * 4;
	o += (size_t)c.tbcap;
	return (o + 255) & ~(size_t)255;
}

// number of leading equal bytes of a[0..maxlen) and b[0..maxlen); buffers are padded by >= 8 bytes
__device__ __forceinline__ int32_t wfa_lcp(const char *a, const char *b, int32_t maxlen)
{
	int32_t n = 0;
	while (n < maxlen) {
		uint64_t x, y;
		__builtin_memcpy(&x, a + n, 8);
		__builtin_memcpy(&y, b + n, 8);
		const uint64_t c = x ^ y;
		if (c) { n += __builtin_ctzll(c) >> 3; break; }
		n += 8;
	}
	return n < maxlen ? n : maxlen;
}

__global__ void __launch_bounds__(64) k_wfa(const int *__restrict__ n_items_p, int cap, int first, const int32_t *__restrict__ list_,
											const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq, const char *__restrict__ qseq,
											mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used,
											char *__restrict__ ws_base, int *__restrict__ counter, mga_wfa_retry_t rt, wfa_cfg_t cfg)
{
	__shared__ int32_t lo_s[WF_NSLOT], hi_s[WF_NSLOT];
	__shared__ int32_t item_s;
	const int lane = threadIdx.x;
	const int n_items = max(0, min(*n_items_p, cap) - first);
	const int32_t *__restrict__ list = list_ + first;
	const wfa_ws_t W = wfa_carve(ws_base + (size_t)blockIdx.x * cfg.ws_stride, cfg);
	const int32_t oe1 = cfg.o1 + cfg.e1, oe2 = cfg.o2 + cfg.e2;
	const int32_t wmax = cfg.wmax;

#define RING(slot, arr) (W.ring + ((size_t)(slot) * 5 + (arr)) * wmax)
	// value of array `arr` of the slice with score `sc` at diagonal d (NEG_INF outside / before score 0)
	auto rd = [&](int32_t sc, int arr, int32_t d) -> int32_t {
		if (sc < 0) return WF_NEG_INF;
		const int slot = sc % WF_NSLOT;
		const int32_t lo = lo_s[slot], hi = hi_s[slot];
		if (d < lo || d > hi) return WF_NEG_INF;
		return RING(slot, arr)[d - lo];
	};

	for (;;) {
		if (lane == 0) item_s = atomicAdd(counter, 1);
		__syncthreads();
		const int item = item_s;
		__syncthreads();
		if (item >= n_items) break;
		const int pi = list ? list[item] : item;
		const mga_wfa_prob_t pb = prob[pi];
		const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
		const int32_t tl = pb.tl, ql = pb.ql;

		if (lane < WF_NSLOT) { lo_s[lane] = 1; hi_s[lane] = 0; }
		__syncthreads();
		if (lane == 0) {
			lo_s[0] = 0; hi_s[0] = 0;
			RING(0, 0)[0] = -1;
			RING(0, 1)[0] = RING(0, 2)[0] = RING(0, 3)[0] = RING(0, 4)[0] = WF_NEG_INF;
			W.row[0] = 0; W.rlo[0] = 0; W.tb[0] = 0;
		}
		__syncthreads();

		int32_t s = 0, wlo = 0, whi = 0, status = MGA_WFA_OK, last_state = 0;
		int64_t n_iter = 0, tb_used = 1;

		for (;;) {
			// ---- extend every diagonal of the current slice along exact matches (miniwfa.c:399-411)
			const int slot = s % WF_NSLOT;
			const int32_t lo = lo_s[slot], hi = hi_s[slot];
			int32_t *Hc = RING(slot, 0);
			bool found = false;
			for (int32_t d0 = lo; d0 <= hi; d0 += 64) {
				const int32_t d = d0 + lane;
				bool term = false;
				int32_t ls = 0;
				if (d <= hi) {
					const int32_t k0 = Hc[d - lo];
					if (!(k0 < -1 || d + k0 < -1 || k0 >= tl || d + k0 >= ql)) {
						int32_t room = tl - (k0 + 1);
						if (ql - (d + k0 + 1) < room) room = ql - (d + k0 + 1);
						const int32_t k = k0 + wfa_lcp(ts + k0 + 1, qs + d + k0 + 1, room);
						term = (k == tl - 1 && d + k == ql - 1);
						if (term) { if (k == k0) ls = W.tb[W.row[s] + (d - lo)] & 7; }
						else Hc[d - lo] = k;
					}
				}
				const uint64_t m = __ballot(term);
				if (m) { last_state = __shfl(ls, __builtin_ctzll(m)); found = true; break; }
			}
			if (found) break;
			// ---- next slice (miniwfa.c:412-415, wf_next_basic :311-325, wf_next_tb :281-308)
			const int32_t nlo = wlo > -tl ? wlo - 1 : -tl;
			const int32_t nhi = whi < ql ? whi + 1 : ql;
			const int32_t width = nhi - nlo + 1;
			if (width > wmax || s + 1 > cfg.smax || tb_used + width > cfg.tbcap) { status = MGA_WFA_RETRY_TIER; break; }
			__syncthreads(); // all extension stores are done before anyone reads them as a predecessor
			++s;
			const int nslot = s % WF_NSLOT;
			if (lane == 0) { lo_s[nslot] = nlo; hi_s[nslot] = nhi; W.row[s] = tb_used; W.rlo[s] = nlo; }
			__syncthreads();
			int32_t *Hn = RING(nslot, 0), *E1n = RING(nslot, 1), *F1n = RING(nslot, 2), *E2n = RING(nslot, 3), *F2n = RING(nslot, 4);
			uint8_t *tbrow = W.tb + tb_used;
			for (int32_t d0 = nlo; d0 <= nhi; d0 += 64) {
				const int32_t d = d0 + lane;
				if (d <= nhi) {
					const int32_t ho1l = rd(s - oe1, 0, d - 1), e1l = rd(s - cfg.e1, 1, d - 1);
					const int32_t ho2l = rd(s - oe2, 0, d - 1), e2l = rd(s - cfg.e2, 3, d - 1);
					const int32_t ho1r = rd(s - oe1, 0, d + 1), f1r = rd(s - cfg.e1, 2, d + 1);
					const int32_t ho2r = rd(s - oe2, 0, d + 1), f2r = rd(s - cfg.e2, 4, d + 1);
					const int32_t hx = rd(s - cfg.x, 0, d);
					uint32_t bits = 0;
					if (!(ho1l >= e1l)) bits |= 0x08;
					const int32_t E1 = ho1l >= e1l ? ho1l : e1l;
					if (!(ho2l >= e2l)) bits |= 0x20;
					const int32_t E2 = ho2l >= e2l ? ho2l : e2l;
					const uint32_t ze = E1 >= E2 ? 1 : 3;
					const int32_t e = E1 >= E2 ? E1 : E2;
					if (!(ho1r >= f1r)) bits |= 0x10;
					const int32_t F1 = (ho1r >= f1r ? ho1r : f1r) + 1;
					if (!(ho2r >= f2r)) bits |= 0x40;
					const int32_t F2 = (ho2r >= f2r ? ho2r : f2r) + 1;
					const uint32_t zf = F1 >= F2 ? 2 : 4;
					const int32_t f = F1 >= F2 ? F1 : F2;
					uint32_t z = e >= f ? ze : zf;
					const int32_t h = e >= f ? e : f;
					if (hx + 1 >= h) z = 0;
					const int32_t H = hx + 1 >= h ? hx + 1 : h;
					const int32_t o = d - nlo;
					Hn[o] = H; E1n[o] = E1; F1n[o] = F1; E2n[o] = E2; F2n[o] = F2;
					tbrow[o] = (uint8_t)(bits | z);
				}
			}
			tb_used += width;
			__syncthreads();
			{ // band bookkeeping (miniwfa.c:323-324): every lane reads the two edge cells (same address: broadcast)
				const int32_t o = nhi - nlo;
				if (Hn[0] >= -1 || E1n[0] >= -1 || F1n[0] >= -1 || E2n[0] >= -1 || F2n[0] >= -1) wlo = nlo;
				if (Hn[o] >= -1 || E1n[o] >= -1 || F1n[o] >= -1 || E2n[o] >= -1 || F2n[o] >= -1) whi = nhi;
			}
			if ((s & 0xff) == 0) { // wf_stripe_shrink (miniwfa.c:139-169)
				auto alive = [&](int32_t d) -> bool {
					for (int j = 0; j < WF_NSLOT; ++j) {
						const int32_t sc = s - j;
						if (sc < 0) break;
						const int sl = sc % WF_NSLOT;
						const int32_t l2 = lo_s[sl], h2 = hi_s[sl];
						if (d < l2 || d > h2) continue;
						for (int a = 0; a < 5; ++a) {
							const int32_t k = RING(sl, a)[d - l2];
							if (k >= -1 && k < tl && d + k >= -1 && d + k < ql) return true;
						}
					}
					return false;
				};
				int32_t d0;
				for (d0 = wlo; d0 <= whi; d0 += 64) {
					const int32_t d = d0 + lane;
					const uint64_t m = __ballot(d <= whi && alive(d));
					if (m) { d0 += __builtin_ctzll(m); break; }
				}
				wlo = d0 > whi ? whi + 1 : d0; // reference asserts a live diagonal exists
				for (d0 = whi; d0 >= wlo; d0 -= 64) {
					const int32_t d = d0 - lane;
					const uint64_t m = __ballot(d >= wlo && alive(d));
					if (m) { d0 -= __builtin_ctzll(m); break; }
				}
				whi = d0 < wlo ? wlo - 1 : d0;
			}
			n_iter += width;
			if (cfg.max_iter > 0 && n_iter > cfg.max_iter) { status = MGA_WFA_MAX_ITER; break; }
		}

		// ---- traceback (miniwfa.c:329-377), wave-cooperative: match runs by ballot, state walk uniform
		int32_t n_cig = 0;
		int64_t cig_off = 0;
		if (status == MGA_WFA_OK) {
			int32_t i = ql - 1, k = tl - 1, sc = s, last = last_state;
			int32_t cur_op = -1, cur_len = 0;
			bool overflow = false;
#define PUSH(op, len) do { \
				if (cur_op == (op)) cur_len += (len); \
				else { \
					if (cur_op >= 0) { if (n_cig < cfg.cigcap) { if (lane == 0) W.cig[n_cig] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; } else overflow = true; ++n_cig; } \
					cur_op = (op), cur_len = (len); \
				} \
			} while (0)
			while (i >= 0 && k >= 0) {
				if (last == 0) {
					int32_t tot = 0;
					for (;;) {
						const bool eq = (i - lane >= 0 && k - lane >= 0) && qs[i - lane] == ts[k - lane];
						const uint64_t m = __ballot(eq);
						const int run = m == ~0ULL ? 64 : __builtin_ctzll(~m);
						tot += run, i -= run, k -= run;
						if (run < 64) break;
					}
					if (tot > 0) PUSH(7, tot);
					if (i < 0 || k < 0) break;
				}
				const uint32_t x = W.tb[W.row[sc] + ((i - k) - W.rlo[sc])];
				const int32_t state = last == 0 ? (int32_t)(x & 7) : last;
				const int32_t ext = state > 0 ? (int32_t)(x >> (state + 2) & 1) : 0;
				if (state == 0) { PUSH(8, 1); --i, --k, sc -= cfg.x; }
				else if (state == 1) { PUSH(1, 1); --i, sc -= ext ? cfg.e1 : oe1; }
				else if (state == 3) { PUSH(1, 1); --i, sc -= ext ? cfg.e2 : oe2; }
				else if (state == 2) { PUSH(2, 1); --k, sc -= ext ? cfg.e1 : oe1; }
				else { PUSH(2, 1); --k, sc -= ext ? cfg.e2 : oe2; }
				last = state > 0 && ext ? state : 0;
			}
			if (i >= 0) PUSH(1, i + 1);
			else if (k >= 0) PUSH(2, k + 1);
			PUSH(15, 0); // flush the pending operator
#undef PUSH
			if (overflow) status = MGA_WFA_RETRY_TIER;
			else {
				unsigned long long o = 0;
				if (lane == 0) o = atomicAdd(pool_used, (unsigned long long)n_cig);
				o = __shfl(o, 0);
				if ((long long)(o + n_cig) > pool_cap) status = MGA_WFA_POOL_FULL;
				else {
					__syncthreads();
					for (int32_t j = lane; j < n_cig; j += 64) pool[o + j] = W.cig[n_cig - 1 - j]; // built back-to-front
					cig_off = (int64_t)o;
				}
			}
		}
		if (lane == 0) {
			mga_wfa_res_t r;
			r.score = status == MGA_WFA_OK ? s : -1;
			r.n_cigar = status == MGA_WFA_OK ? n_cig : 0;
			r.cig_off = cig_off, r.status = status, r.pad = 0, r.n_iter = n_iter;
			res[pi] = r;
			if (status == MGA_WFA_RETRY_TIER) mga_wfa_give_up(rt, pi); // next tier's work list
			else if (status == MGA_WFA_MAX_ITER) rt.fb_list[atomicAdd(rt.fb_cnt, 1)] = pi; // for the chained fallback
			else if (status != MGA_WFA_OK) atomicAdd(rt.err, 1);
		}
		__syncthreads();
	}
#undef RING
}

// ---- host driver -------------------------------------------------------------------------------

static const wfa_cfg_t g_tier[3] = {
	// x o1 e1 o2 e2   wmax   smax   cigcap   tbcap        max_iter   stride
	{ 4, 4, 2, 15, 1,  4096,   8192,   65536,  1 << 24,    100000000, 0 },
	{ 4, 4, 2, 15, 1, 32768,  32768, 1 << 20,  104000000,  100000000, 0 }, // tbcap just above max_iter: the cap triggers first
	// the sub-problems of the chained fallback have no cell cap (miniwfa.c:831): last tier for them -- round 5: a million diagonals and 32 GiB of traceback for ONE problem at a
	// time (allocated when a stretch first gets here: 33 GB of the 288), where rounds 1-4 stopped at 131 072 diagonals / 4 GiB.  The reference bounds the memory of such a stretch
	// instead (mwf_wfa_seg, miniwfa.c:440-601: checkpoints every 5000 scores, then a second pass whose band restarts at each checkpoint) and arrives at the same CIGAR (DESIGN 4,
	// chained fallback); a stretch beyond THIS tier -- two unrelated sequences of ~180 kb each without a shared 13-mer chain between them -- still fails loudly.
	{ 4, 4, 2, 15, 1, 1 << 20, 1 << 20, 1 << 22, 32LL << 30,  -1,        0 },
};
static const int g_tier_waves[3] = { 256, 16, 1 };


extern "C" int mga_dev_wfa(mga_sctx_t *sc, const int *d_n, int n, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
						   mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt)
{
	if (n <= 0) return 0;
	if (tier < 0 || tier > 1) { mga_set_error("wfa: bad tier %d", tier); return -1; }
	if (sc->wfa_uncapped) tier = tier == 1 ? 2 : tier; // ladder of the fallback's sub-problems: same first HBM tier, then the unbounded one
	wfa_cfg_t cfg = g_tier[tier];
	if (sc->wfa_uncapped) cfg.max_iter = -1;
	cfg.ws_stride = (int64_t)wfa_ws_bytes(cfg);
	int waves = g_tier_waves[tier];
	if (waves > n - first) waves = n - first;
	if (mga_dbuf_reserve(&sc->wfa_ws[7 + tier], (size_t)cfg.ws_stride * waves) < 0) return -1;
	hipStream_t st = (hipStream_t)(stream ? stream : sc->stream);
	int *d_counter = (int*)((char*)sc->wfa_cnt.p + 64 * slot);
	const int kid = MGA_K_WFA0 + 7 + (tier > 1 ? 1 : tier); // the unbounded tier is accounted with the widest regular one
	mga_prof_begin(st, kid);
	hipLaunchKernelGGL(k_wfa, dim3(waves), dim3(64), 0, st, d_n, n, first, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap,
					   d_pool_used, (char*)sc->wfa_ws[7 + tier].p, d_counter, rt, cfg);
	mga_prof_end(st, kid);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
