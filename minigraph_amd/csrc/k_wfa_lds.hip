// k_wfa_lds.hip -- LDS-resident exact 2-piece affine WFA: the fast tiers of the gap-filling kernel.
//
// Same algorithm, same bit-exact semantics as k_wfa.hip (reference miniwfa.c:281-435, see the header
// of that file); what changes is where the state lives:
//
//   * wavefront ring in LDS, trimmed to what the recurrence reads: H needs the last 17 scores
//     (x=4, o1+e1=6, o2+e2=16), E1/F1 the last 3 (e1=2), E2/F2 the last 2 (e2=1): 27 rows of W int32
//     instead of the reference's 85 (W=128: 13.5 KB, W=512: 54 KB, W=1024: 108 KB);
//   * both sequences staged into LDS once per problem (coalesced 16-byte global loads), extension
//     compares 4 bytes per step from aligned dword reads + funnel shift;
//   * the periodic dead-diagonal trimming (miniwfa.c:139-169,420) needs "does diagonal d hold an
//     in-matrix value in any of the last 17 slices".  In-matrix-ness of a cell never changes after it
//     is created (extension keeps H inside the matrix), so one int16 per diagonal -- the last score at
//     which the diagonal received an in-matrix value -- replaces the 85-row scan;
//   * traceback bytes stream to a per-workgroup HBM scratch (write-once, L2-resident when the walk
//     reads them back), CIGAR is built by wave 0.
//
// One workgroup per problem, lanes = diagonals: NT = 64 threads for bands <= 128, 256 threads for
// bands <= 512 / 1024.  Persistent workgroups pull problems from an atomic queue.  A problem whose
// band, sequence length or traceback outgrows the tier returns MGA_WFA_RETRY_TIER and is re-run by the
// next tier (finally by the HBM-resident kernel of k_wfa.hip); nothing runs on the CPU.
#include "mga_dev.h"
#include "dev_common.h"

#define WF_NEG_INF (-0x40000000)

struct wfl_cfg_t {
	int32_t x, o1, e1, o2, e2;
	int32_t smax, cigcap;
	int64_t tbcap, max_iter;
	int64_t ws_stride;
};

__device__ __forceinline__ uint32_t wfl_load4(const uint32_t *w, int32_t p) // 4 bytes at byte offset p of an LDS dword array
{
	const int32_t i = p >> 2;
	return __funnelshift_r(w[i], w[i + 1], (p & 3) << 3);
}

// leading equal bytes of t[tp..] and q[qp..], at most maxlen
__device__ __forceinline__ int32_t wfl_lcp(const uint32_t *t, int32_t tp, const uint32_t *q, int32_t qp, int32_t maxlen)
{
	int32_t n = 0;
	while (n < maxlen) {
		const uint32_t c = wfl_load4(t, tp + n) ^ wfl_load4(q, qp + n);
		if (c) { n += __builtin_ctz(c) >> 3; break; }
		n += 4;
	}
	return n < maxlen ? n : maxlen;
}

template<int NT, int W, int SEQCAP, int SMAX, int TBLDS>
__global__ void __launch_bounds__(NT) k_wfa_lds(int n_items, const int32_t *__restrict__ list,
												const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq, const char *__restrict__ qseq,
												mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used,
												char *__restrict__ ws_base, int *__restrict__ counter, wfl_cfg_t cfg)
{
	constexpr int SEQW = (SEQCAP + 16) / 4; // dwords per staged sequence (>= 8 pad bytes)
	__shared__ int32_t Hr[17 * W], E1r[3 * W], F1r[3 * W], E2r[2 * W], F2r[2 * W];
	__shared__ uint32_t Ts[SEQW], Qs[SEQW];
	__shared__ int16_t good_last[2 * SEQCAP + 2];
	__shared__ int32_t lo_s[17], hi_s[17];
	__shared__ int32_t sh_item, sh_flag[4]; // [0] reach_lo [1] reach_hi [2] term last_state+1 [3] tb byte of the end diagonal
	__shared__ int32_t row[SMAX + 1];       // start of each score's traceback row
	__shared__ int16_t rlo[SMAX + 1];       // its lowest diagonal
	__shared__ uint8_t tb_lds[TBLDS > 0 ? TBLDS : 4];
	const int tid = threadIdx.x, lane = tid & 63;
	const int32_t oe1 = cfg.o1 + cfg.e1, oe2 = cfg.o2 + cfg.e2;
	// per-workgroup HBM scratch: row table, row lo, cigar, traceback bytes
	char *wsb = ws_base + (size_t)blockIdx.x * cfg.ws_stride;
	uint32_t *cig = (uint32_t*)wsb;
	uint8_t *tb = TBLDS > 0 ? tb_lds : (uint8_t*)(cig + cfg.cigcap);
	const int64_t tbcap = TBLDS > 0 ? (int64_t)TBLDS : cfg.tbcap;
	// LDS-only barrier: LDS traffic is drained (lgkmcnt) but traceback stores to HBM stay in flight
#define WFL_BAR() do { if (NT > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)

	auto rdH = [&](int32_t sc, int32_t d) -> int32_t {
		if (sc < 0) return WF_NEG_INF;
		const int sl = sc % 17;
		const int32_t lo = lo_s[sl];
		return (d < lo || d > hi_s[sl]) ? WF_NEG_INF : Hr[sl * W + d - lo];
	};
	auto rdX = [&](const int32_t *ring, int depth, int32_t sc, int32_t d) -> int32_t {
		if (sc < 0) return WF_NEG_INF;
		const int sl = sc % 17;
		const int32_t lo = lo_s[sl];
		return (d < lo || d > hi_s[sl]) ? WF_NEG_INF : ring[(sc % depth) * W + d - lo];
	};

	// CIGAR pool: each workgroup sub-allocates from blocks it takes from the global bump pointer (one atomic per
	// POOL_BLK operators instead of one per problem: a single hot word saturates at ~88 atomics/us on this chip)
	constexpr int POOL_BLK = 512;
	long long blk_beg = 0, blk_end = 0; // wave-0 registers
	constexpr int QCHUNK = 4; // problems taken per queue atomic
	int q_next = 0, q_end = 0;
	for (;;) {
		__syncthreads();
		if (q_next == q_end) {
			if (tid == 0) sh_item = atomicAdd(counter, QCHUNK);
			__syncthreads();
			q_next = sh_item, q_end = q_next + QCHUNK;
		}
		const int item = q_next++;
		if (item >= n_items) break;
		const int pi = list ? list[item] : item;
		const mga_wfa_prob_t pb = prob[pi];
		const int32_t tl = pb.tl, ql = pb.ql;
		int32_t status = MGA_WFA_OK, s = 0, wlo = 0, whi = 0, last_state = 0;
		int64_t n_iter = 0, tb_used = 1;

		if (tl > SEQCAP || ql > SEQCAP) status = MGA_WFA_RETRY_TIER;
		else {
			// stage both sequences (byte copies through aligned dword stores; sources are arbitrary-aligned)
			const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
			uint8_t *Tb = (uint8_t*)Ts, *Qb = (uint8_t*)Qs;
			for (int32_t i = tid; i < tl + 8; i += NT) Tb[i] = i < tl ? (uint8_t)ts[i] : (uint8_t)0;
			for (int32_t i = tid; i < ql + 8; i += NT) Qb[i] = i < ql ? (uint8_t)qs[i] : (uint8_t)1; // distinct pads, as wf_pad_str does
			for (int32_t i = tid; i < tl + ql + 1; i += NT) good_last[i] = -1;
			if (tid < 17) { lo_s[tid] = 1; hi_s[tid] = 0; }
			__syncthreads();
			if (tid == 0) {
				lo_s[0] = 0; hi_s[0] = 0; Hr[0] = -1;
				E1r[0] = F1r[0] = E2r[0] = F2r[0] = WF_NEG_INF;
				row[0] = 0; rlo[0] = 0; tb[0] = 0; sh_flag[3] = 0;
				good_last[0 + tl] = 0; // H[0][0] = -1 is inside the matrix
			}
			__syncthreads();

			for (;;) {
				// ---- extension (miniwfa.c:399-411)
				const int slot = s % 17;
				const int32_t lo = lo_s[slot], hi = hi_s[slot];
				int32_t *Hc = Hr + slot * W;
				if (tid == 0) sh_flag[2] = 0;
				WFL_BAR();
				for (int32_t d = lo + tid; d <= hi; d += NT) {
					const int32_t k0 = Hc[d - lo];
					if (!(k0 < -1 || d + k0 < -1 || k0 >= tl || d + k0 >= ql)) {
						int32_t room = tl - (k0 + 1);
						if (ql - (d + k0 + 1) < room) room = ql - (d + k0 + 1);
						const int32_t k = k0 + wfl_lcp(Ts, k0 + 1, Qs, d + k0 + 1, room);
						if (k == tl - 1 && d + k == ql - 1) sh_flag[2] = 1 + (k == k0 ? (sh_flag[3] & 7) : 0); // only d = ql - tl can end
						else Hc[d - lo] = k;
					}
				}
				WFL_BAR();
				if (sh_flag[2]) { last_state = sh_flag[2] - 1; break; }
				// ---- next slice (miniwfa.c:281-325,412-415)
				const int32_t nlo = wlo > -tl ? wlo - 1 : -tl;
				const int32_t nhi = whi < ql ? whi + 1 : ql;
				const int32_t width = nhi - nlo + 1;
				if (width > W || s + 1 > SMAX || tb_used + width > tbcap) { status = MGA_WFA_RETRY_TIER; break; }
				++s;
				const int nslot = s % 17;
				if (tid == 0) { lo_s[nslot] = nlo; hi_s[nslot] = nhi; row[s] = (int32_t)tb_used; rlo[s] = (int16_t)nlo; sh_flag[0] = 0; sh_flag[1] = 0; }
				WFL_BAR();
				int32_t *Hn = Hr + nslot * W, *E1n = E1r + (s % 3) * W, *F1n = F1r + (s % 3) * W, *E2n = E2r + (s % 2) * W, *F2n = F2r + (s % 2) * W;
				uint8_t *tbrow = tb + tb_used;
				for (int32_t d = nlo + tid; d <= nhi; d += NT) {
					const int32_t ho1l = rdH(s - oe1, d - 1), e1l = rdX(E1r, 3, s - cfg.e1, d - 1);
					const int32_t ho2l = rdH(s - oe2, d - 1), e2l = rdX(E2r, 2, s - cfg.e2, d - 1);
					const int32_t ho1r = rdH(s - oe1, d + 1), f1r = rdX(F1r, 3, s - cfg.e1, d + 1);
					const int32_t ho2r = rdH(s - oe2, d + 1), f2r = rdX(F2r, 2, s - cfg.e2, d + 1);
					const int32_t hx = rdH(s - cfg.x, d);
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
					if (d == ql - tl) sh_flag[3] = (int32_t)(bits | z);
					const bool reach = H >= -1 || E1 >= -1 || F1 >= -1 || E2 >= -1 || F2 >= -1;
					if (d == nlo && reach) sh_flag[0] = 1;
					if (d == nhi && reach) sh_flag[1] = 1;
#define WFL_IN(k_) ((k_) >= -1 && (k_) < tl && d + (k_) >= -1 && d + (k_) < ql)
					if (WFL_IN(H) || WFL_IN(E1) || WFL_IN(F1) || WFL_IN(E2) || WFL_IN(F2)) good_last[d + tl] = (int16_t)s;
#undef WFL_IN
				}
				tb_used += width;
				WFL_BAR();
				if (sh_flag[0]) wlo = nlo;
				if (sh_flag[1]) whi = nhi;
				if ((s & 0xff) == 0) { // trimming: a diagonal is alive iff it got an in-matrix value within the last 17 scores
					WFL_BAR(); // everyone has consumed the reach flags
					if (tid == 0) { sh_flag[0] = 0x7fffffff; sh_flag[1] = -0x7fffffff; }
					WFL_BAR();
					int32_t mn = 0x7fffffff, mx = -0x7fffffff;
					for (int32_t d = wlo + tid; d <= whi; d += NT)
						if ((int32_t)good_last[d + tl] > s - 17) { if (d < mn) mn = d; if (d > mx) mx = d; }
					if (mn != 0x7fffffff) { atomicMin(&sh_flag[0], mn); atomicMax(&sh_flag[1], mx); }
					WFL_BAR();
					if (sh_flag[0] != 0x7fffffff) { wlo = sh_flag[0]; whi = sh_flag[1]; }
					else { const int32_t e0 = whi + 1; wlo = e0; whi = e0 - 1; } // reference would assert
					WFL_BAR();
				}
				n_iter += width;
				if (cfg.max_iter > 0 && n_iter > cfg.max_iter) { status = MGA_WFA_MAX_ITER; break; }
			}
		}
		__syncthreads();

		// ---- traceback by wave 0 (miniwfa.c:329-377)
		if (tid < 64) {
			int32_t n_cig = 0;
			int64_t cig_off = 0;
			if (status == MGA_WFA_OK) {
				const uint8_t *Tb = (const uint8_t*)Ts, *Qb = (const uint8_t*)Qs;
				int32_t i = ql - 1, k = tl - 1, sc = s, last = last_state;
				int32_t cur_op = -1, cur_len = 0;
				bool overflow = false;
#define PUSH(op, len) do { \
					if (cur_op == (op)) cur_len += (len); \
					else { \
						if (cur_op >= 0) { if (n_cig < cfg.cigcap) { if (lane == 0) cig[n_cig] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; } else overflow = true; ++n_cig; } \
						cur_op = (op), cur_len = (len); \
					} \
				} while (0)
				while (i >= 0 && k >= 0) {
					if (last == 0) {
						int32_t tot = 0;
						for (;;) {
							const bool eq = (i - lane >= 0 && k - lane >= 0) && Qb[i - lane] == Tb[k - lane];
							const uint64_t m = __ballot(eq);
							const int run = m == ~0ULL ? 64 : __builtin_ctzll(~m);
							tot += run, i -= run, k -= run;
							if (run < 64) break;
						}
						if (tot > 0) PUSH(7, tot);
						if (i < 0 || k < 0) break;
					}
					const uint32_t x = tb[row[sc] + ((i - k) - (int32_t)rlo[sc])];
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
				PUSH(15, 0);
#undef PUSH
				if (overflow) status = MGA_WFA_RETRY_TIER;
				else {
					if (blk_end - blk_beg < n_cig) { // take a fresh block (the tail of the old one is abandoned)
						const long long want = n_cig > POOL_BLK ? n_cig : POOL_BLK;
						unsigned long long o2 = 0;
						if (lane == 0) o2 = atomicAdd(pool_used, (unsigned long long)want);
						o2 = __shfl(o2, 0);
						blk_beg = (long long)o2, blk_end = blk_beg + want;
					}
					const unsigned long long o = (unsigned long long)blk_beg;
					if ((long long)(o + n_cig) > pool_cap) status = MGA_WFA_POOL_FULL;
					else {
						blk_beg += n_cig;
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						__builtin_amdgcn_wave_barrier();
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						for (int32_t j = lane; j < n_cig; j += 64) pool[o + j] = cig[n_cig - 1 - j];
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
			}
		}
	}
}

// ---- host driver ---------------------------------------------------------------------------------

struct wfl_tier_t { int nt, w, seqcap, n_wg; int32_t smax, cigcap; int64_t tbcap; };
static const wfl_tier_t g_ltier[3] = {
	//  NT    W   SEQCAP  WGs    smax  cigcap   tbcap (HBM traceback scratch; tier 0 keeps its traceback in LDS)
	{   64,  128,   256, 1536,    256,   1024,   0 },
	{  256,  512,  1024,  768,   1024,   4096,   512 << 10 },
	{  256, 1024,  1024,  256,   2048,   8192,    2 << 20 },
};

static size_t wfl_ws_bytes(const wfl_tier_t &t)
{
	size_t o = (size_t)t.cigcap * 4 + (size_t)t.tbcap;
	return (o + 255) & ~(size_t)255;
}

extern "C" int mga_dev_wfa_lds(mga_sctx_t *sc, int n, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
							   mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier)
{
	if (n <= 0) return 0;
	if (tier < 0 || tier > 2) { mga_set_error("wfa_lds: bad tier %d", tier); return -1; }
	const wfl_tier_t &T = g_ltier[tier];
	wfl_cfg_t cfg = { 4, 4, 2, 15, 1, T.smax, T.cigcap, T.tbcap, 100000000, 0 }; // ring depths 17/3/2 are tied to these penalties (miniwfa.c:11-18)
	cfg.ws_stride = (int64_t)wfl_ws_bytes(T);
	int wgs = T.n_wg < n ? T.n_wg : n;
	if (mga_dbuf_reserve(&sc->wfa_ws[4], (size_t)cfg.ws_stride * T.n_wg) < 0) return -1;
	if (mga_dbuf_reserve(&sc->wfa_cnt, 256) < 0) return -1;
	MGA_HIP_CHECK(hipMemsetAsync(sc->wfa_cnt.p, 0, 4, (hipStream_t)sc->stream));
	mga_prof_begin(sc, MGA_K_WFA0 + 4);
	if (tier != 2) { mga_set_error("wfa_lds: only the band-1024 tier is instantiated"); return -1; }
		hipLaunchKernelGGL((k_wfa_lds<256, 1024, 1024, 2048, 0>), dim3(wgs), dim3(256), 0, (hipStream_t)sc->stream, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[4].p, (int*)sc->wfa_cnt.p, cfg);
	mga_prof_end(sc, MGA_K_WFA0 + 4);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
