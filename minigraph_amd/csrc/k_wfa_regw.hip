// k_wfa_regw.hip -- register-resident exact WFA for WIDE bands: NW wavefronts of one workgroup share a
// problem, each keeping a contiguous block of 64*J diagonals in VGPRs (see k_wfa_reg.hip for the
// single-wave version and the shared semantics; reference miniwfa.c:281-435).
//
// A band of 256..512 diagonals in ONE wave needs 4..8 slots per lane: ~260 VGPRs, one wave per SIMD and a
// serial chain of 8 slot updates per score.  Spreading the diagonals over the 4 SIMDs of a CU keeps every
// wave at J <= 2 (4 waves/SIMD resident) and runs the slot updates of one score in parallel.  Per score the
// waves exchange, through LDS, the 8 boundary values their neighbours read next (H[s-6], H[s-16], E1/F1[s-2],
// E2/F2[s-1] of the first and last diagonal of each block), the two "edge cell reachable" flags and the
// termination flag -- ONE s_barrier per score, double-buffered by score parity.  Control (band bookkeeping,
// capacity checks, trimming) is computed redundantly and identically by every wave from those flags.
#include "mga_dev.h"
#include "dev_common.h"

#define WF_NEG_INF (-0x40000000)

struct wfw_cfg_t {
	int32_t x, o1, e1, o2, e2;
	int32_t cigcap;
	int64_t tbcap, max_iter, ws_stride;
};

__device__ __forceinline__ uint32_t wfw_load4(const uint32_t *w, int32_t p)
{
	const int32_t i = p >> 2;
	return __funnelshift_r(w[i], w[i + 1], (p & 3) << 3);
}

__device__ __forceinline__ int32_t wfw_lcp(const uint32_t *t, int32_t tp, const uint32_t *q, int32_t qp, int32_t maxlen)
{
	int32_t n = 0;
	while (n < maxlen) {
		const uint32_t c = wfw_load4(t, tp + n) ^ wfw_load4(q, qp + n);
		if (c) { n += __builtin_ctz(c) >> 3; break; }
		n += 4;
	}
	return n < maxlen ? n : maxlen;
}

template<int NW, int J, int SEQCAP, int SMAX>
__global__ void __launch_bounds__(64 * NW) k_wfa_regw(int n_items, const int32_t *__restrict__ list,
													 const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq, const char *__restrict__ qseq,
													 mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used,
													 char *__restrict__ ws_base, int *__restrict__ counter, wfw_cfg_t cfg)
{
	constexpr int NV = 64 * J * NW;  // diagonals covered by the workgroup
	constexpr int NT = 64 * NW;
	constexpr int SEQW = (SEQCAP + 16) / 4;
	__shared__ uint32_t Ts[SEQW], Qs[SEQW];
	__shared__ int32_t row[SMAX + 1];
	__shared__ int16_t rlo[SMAX + 1];
	__shared__ int32_t edgeL[2][NW][4], edgeR[2][NW][4]; // [parity][wave]: values of the block's FIRST (L) / LAST (R) diagonal
	__shared__ int32_t f_reach_lo[2], f_reach_hi[2], f_term, f_last, f_item, f_mn, f_mx;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int32_t oe1 = cfg.o1 + cfg.e1, oe2 = cfg.o2 + cfg.e2;
	char *wsb = ws_base + (size_t)blockIdx.x * cfg.ws_stride;
	uint32_t *cig = (uint32_t*)wsb;
	uint8_t *tb = (uint8_t*)(cig + cfg.cigcap);
	const int64_t tbcap = cfg.tbcap;
#define WFW_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

	constexpr int POOL_BLK = 512, QCHUNK = 4;
	long long blk_beg = 0, blk_end = 0; // wave 0
	int q_next = 0, q_end = 0;

	for (;;) {
		__syncthreads(); // previous problem completely finished (traceback reads LDS, HBM scratch)
		if (q_next == q_end) {
			if (tid == 0) f_item = atomicAdd(counter, QCHUNK);
			__syncthreads();
			q_next = f_item, q_end = q_next + QCHUNK;
		}
		const int item = q_next++;
		if (item >= n_items) break;
		const int pi = list ? list[item] : item;
		const mga_wfa_prob_t pb = prob[pi];
		const int32_t tl = pb.tl, ql = pb.ql;
		int32_t status = MGA_WFA_OK, s = 0, wlo = 0, whi = 0, last_state = 0, clo = 0, chi = 0;
		int64_t n_iter = 0, tb_used = 1;

		if (tl > SEQCAP || ql > SEQCAP) status = MGA_WFA_RETRY_TIER;
		else {
			int32_t D0 = -(NV / 2);
			if (-tl > D0) D0 = -tl;
			else if (ql < D0 + NV - 1) { D0 = ql - NV + 1; if (D0 < -tl) D0 = -tl; }
			const int32_t W0 = D0 + 64 * J * wv; // first diagonal of this wave's block
			{
				const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
				uint8_t *Tb = (uint8_t*)Ts, *Qb = (uint8_t*)Qs;
				for (int32_t i = tid; i < tl + 8; i += NT) Tb[i] = i < tl ? (uint8_t)ts[i] : (uint8_t)0;
				for (int32_t i = tid; i < ql + 8; i += NT) Qb[i] = i < ql ? (uint8_t)qs[i] : (uint8_t)1;
				if (tid == 0) { row[0] = 0; rlo[0] = 0; tb[0] = 0; f_term = 0; f_last = 0; f_reach_lo[0] = f_reach_lo[1] = f_reach_hi[0] = f_reach_hi[1] = -1; }
				if (tid < 2 * NW * 4) { ((int32_t*)edgeL)[tid] = WF_NEG_INF; ((int32_t*)edgeR)[tid] = WF_NEG_INF; }
				__syncthreads();
			}
			int32_t H[J][17], E1[J][3], F1[J][3], E2[J][2], F2[J][2], GL[J], TBC[J];
#pragma unroll
			for (int j = 0; j < J; ++j) {
#pragma unroll
				for (int a = 0; a < 17; ++a) H[j][a] = WF_NEG_INF;
#pragma unroll
				for (int a = 0; a < 3; ++a) E1[j][a] = F1[j][a] = WF_NEG_INF;
#pragma unroll
				for (int a = 0; a < 2; ++a) E2[j][a] = F2[j][a] = WF_NEG_INF;
				GL[j] = -1, TBC[j] = 0;
				if (W0 + lane + 64 * j == 0) H[j][0] = -1, GL[j] = 0;
			}

			for (;;) {
				const int par = s & 1; // edges/flags published at the end of the step that produced score s
				// ---- extension of slice s (miniwfa.c:399-411)
#pragma unroll
				for (int j = 0; j < J; ++j) {
					if (W0 + 64 * j > chi || W0 + 64 * j + 63 < clo) continue;
					const int32_t d = W0 + lane + 64 * j, k0 = H[j][0];
					if (!(k0 < -1 || d + k0 < -1 || k0 >= tl || d + k0 >= ql)) {
						int32_t room = tl - (k0 + 1);
						if (ql - (d + k0 + 1) < room) room = ql - (d + k0 + 1);
						const int32_t k = k0 + wfw_lcp(Ts, k0 + 1, Qs, d + k0 + 1, room);
						if (k == tl - 1 && d + k == ql - 1) { f_last = k == k0 ? (TBC[j] & 7) : 0; f_term = s + 1; } // unique diagonal ql - tl
						else H[j][0] = k;
					}
				}
				// ---- next slice s+1, computed speculatively by everyone (the terminating wave cannot tell the others before the barrier)
				const int32_t nlo = wlo > -tl ? wlo - 1 : -tl;
				const int32_t nhi = whi < ql ? whi + 1 : ql;
				const int32_t width = nhi - nlo + 1;
				const bool track_alive = ((s + 1) & 0xff) >= 239 || ((s + 1) & 0xff) == 0; // the trimming at score 256k looks back 17 scores only
				const bool fits = !(nlo < D0 || nhi > D0 + NV - 1 || s + 1 > SMAX || tb_used + width > tbcap);
				int32_t nH[J], nE1[J], nF1[J], nE2[J], nF2[J];
				bool reach_lo = false, reach_hi = false;
				if (fits) {
					if (tid == 0) { row[s + 1] = (int32_t)tb_used; rlo[s + 1] = (int16_t)nlo; }
					// boundary values from the neighbouring waves (published after the previous step)
					int32_t eL[4], eR[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						eL[q] = wv > 0 ? edgeR[par][wv > 0 ? wv - 1 : 0][q] : WF_NEG_INF;          // my left neighbour = last diagonal of wave wv-1
						eR[q] = wv < NW - 1 ? edgeL[par][wv < NW - 1 ? wv + 1 : wv][q] : WF_NEG_INF; // my right neighbour = first diagonal of wave wv+1
					}
#pragma unroll
					for (int j = 0; j < J; ++j) {
						if (W0 + 64 * j > nhi || W0 + 64 * j + 63 < nlo) { nH[j] = nE1[j] = nF1[j] = nE2[j] = nF2[j] = WF_NEG_INF; continue; }
						const int32_t d = W0 + lane + 64 * j;
#define WFW_LEFT(R, q)  ({ int32_t u_ = __shfl_up((R)[j], 1); const int32_t w_ = j > 0 ? __shfl((R)[j > 0 ? j - 1 : 0], 63) : eL[q]; lane == 0 ? w_ : u_; })
#define WFW_RIGHT(R, q) ({ int32_t u_ = __shfl_down((R)[j], 1); const int32_t w_ = j < J - 1 ? __shfl((R)[j < J - 1 ? j + 1 : j], 0) : eR[q]; lane == 63 ? w_ : u_; })
						int32_t Ho1[J], Ho2[J], E1p[J], F1p[J], E2p[J], F2p[J];
#pragma unroll
						for (int jj = 0; jj < J; ++jj) { Ho1[jj] = H[jj][5]; Ho2[jj] = H[jj][15]; E1p[jj] = E1[jj][1]; F1p[jj] = F1[jj][1]; E2p[jj] = E2[jj][0]; F2p[jj] = F2[jj][0]; }
						const int32_t ho1l = WFW_LEFT(Ho1, 0), e1l = WFW_LEFT(E1p, 1), ho2l = WFW_LEFT(Ho2, 2), e2l = WFW_LEFT(E2p, 3);
						const int32_t ho1r = WFW_RIGHT(Ho1, 0), f1r = WFW_RIGHT(F1p, 1), ho2r = WFW_RIGHT(Ho2, 2), f2r = WFW_RIGHT(F2p, 3);
						const int32_t hx = H[j][3];
						uint32_t bits = 0;
						if (!(ho1l >= e1l)) bits |= 0x08;
						int32_t vE1 = ho1l >= e1l ? ho1l : e1l;
						if (!(ho2l >= e2l)) bits |= 0x20;
						int32_t vE2 = ho2l >= e2l ? ho2l : e2l;
						const uint32_t ze = vE1 >= vE2 ? 1 : 3;
						const int32_t e = vE1 >= vE2 ? vE1 : vE2;
						if (!(ho1r >= f1r)) bits |= 0x10;
						int32_t vF1 = (ho1r >= f1r ? ho1r : f1r) + 1;
						if (!(ho2r >= f2r)) bits |= 0x40;
						int32_t vF2 = (ho2r >= f2r ? ho2r : f2r) + 1;
						const uint32_t zf = vF1 >= vF2 ? 2 : 4;
						const int32_t f = vF1 >= vF2 ? vF1 : vF2;
						uint32_t z = e >= f ? ze : zf;
						const int32_t h = e >= f ? e : f;
						if (hx + 1 >= h) z = 0;
						int32_t vH = hx + 1 >= h ? hx + 1 : h;
						if (d >= nlo && d <= nhi) {
							tb[tb_used + (d - nlo)] = (uint8_t)(bits | z);
							TBC[j] = (int32_t)(bits | z);
							const bool reach = vH >= -1 || vE1 >= -1 || vF1 >= -1 || vE2 >= -1 || vF2 >= -1;
							if (d == nlo) reach_lo = reach;
							if (d == nhi) reach_hi = reach;
#define WFW_IN(k_) ((k_) >= -1 && (k_) < tl && d + (k_) >= -1 && d + (k_) < ql)
							if (track_alive && (WFW_IN(vH) || WFW_IN(vE1) || WFW_IN(vF1) || WFW_IN(vE2) || WFW_IN(vF2))) GL[j] = s + 1;
#undef WFW_IN
						} else vH = vE1 = vF1 = vE2 = vF2 = WF_NEG_INF;
						nH[j] = vH, nE1[j] = vE1, nF1[j] = vF1, nE2[j] = vE2, nF2[j] = vF2;
					}
#undef WFW_LEFT
#undef WFW_RIGHT
#pragma unroll
					for (int j = 0; j < J; ++j) { // age shift
#pragma unroll
						for (int a = 16; a > 0; --a) H[j][a] = H[j][a - 1];
						H[j][0] = nH[j];
						E1[j][2] = E1[j][1]; E1[j][1] = E1[j][0]; E1[j][0] = nE1[j];
						F1[j][2] = F1[j][1]; F1[j][1] = F1[j][0]; F1[j][0] = nF1[j];
						E2[j][1] = E2[j][0]; E2[j][0] = nE2[j];
						F2[j][1] = F2[j][0]; F2[j][0] = nF2[j];
					}
					// publish what the neighbours read in the NEXT step (post-shift ages) into the other parity's buffers
					if (lane == 0)  { edgeL[par ^ 1][wv][0] = H[0][5]; edgeL[par ^ 1][wv][1] = F1[0][1]; edgeL[par ^ 1][wv][2] = H[0][15]; edgeL[par ^ 1][wv][3] = F2[0][0]; }
					if (lane == 63) { edgeR[par ^ 1][wv][0] = H[J - 1][5]; edgeR[par ^ 1][wv][1] = E1[J - 1][1]; edgeR[par ^ 1][wv][2] = H[J - 1][15]; edgeR[par ^ 1][wv][3] = E2[J - 1][0]; }
					if (reach_lo) f_reach_lo[par ^ 1] = s + 1;
					if (reach_hi) f_reach_hi[par ^ 1] = s + 1;
				}
				WFW_BAR();
				if (f_term) { last_state = f_last; break; } // slice s reached the end: the speculative slice is dropped
				if (!fits) { status = MGA_WFA_RETRY_TIER; break; }
				++s;
				tb_used += width;
				clo = nlo, chi = nhi;
				if (f_reach_lo[s & 1] == s) wlo = nlo;
				if (f_reach_hi[s & 1] == s) whi = nhi;
				if ((s & 0xff) == 0) { // trimming (miniwfa.c:139-169): min/max alive diagonal over the workgroup
					if (tid == 0) { f_mn = 0x7fffffff; f_mx = -0x7fffffff; }
					WFW_BAR();
					int32_t mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t d = W0 + lane + 64 * j;
						if (d >= wlo && d <= whi && GL[j] > s - 17) { if (d < mn) mn = d; if (d > mx) mx = d; }
					}
					if (mn != 0x7fffffff) { atomicMin(&f_mn, mn); atomicMax(&f_mx, mx); }
					WFW_BAR();
					if (f_mn != 0x7fffffff) { wlo = f_mn; whi = f_mx; }
					else { const int32_t e0 = whi + 1; wlo = e0; whi = e0 - 1; }
					WFW_BAR();
				}
				n_iter += width;
				if (cfg.max_iter > 0 && n_iter > cfg.max_iter) { status = MGA_WFA_MAX_ITER; break; }
			}
		}
		__syncthreads(); // HBM traceback rows complete and visible to wave 0

		// ---- traceback by wave 0 (miniwfa.c:329-377)
		if (wv == 0) {
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
					if (blk_end - blk_beg < n_cig) {
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

// ---- host driver: tier 0 = 4 waves x 1 slot (256 diagonals), 1 = 4 waves x 2 slots (512), 2 = 8 waves x 2 slots (1024) ----
struct wfw_tier_t { int n_wg; int32_t cigcap; int64_t tbcap; };
static const wfw_tier_t g_wtier[3] = { { 2048, 2048, 192 << 10 }, { 1024, 4096, 768 << 10 }, { 512, 8192, 3 << 20 } };

extern "C" int mga_dev_wfa_regw(mga_sctx_t *sc, int n, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
								mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, int ws_slot)
{
	if (n <= 0) return 0;
	if (tier < 0 || tier > 2) { mga_set_error("wfa_regw: bad tier %d", tier); return -1; }
	const wfw_tier_t &T = g_wtier[tier];
	wfw_cfg_t cfg = { 4, 4, 2, 15, 1, T.cigcap, T.tbcap, 100000000, 0 };
	cfg.ws_stride = (int64_t)(((size_t)T.cigcap * 4 + (size_t)T.tbcap + 255) & ~(size_t)255);
	int wgs = T.n_wg < (n + 3) / 4 ? T.n_wg : (n + 3) / 4;
	if (wgs < 1) wgs = 1;
	if (mga_dbuf_reserve(&sc->wfa_ws[ws_slot], (size_t)cfg.ws_stride * T.n_wg) < 0) return -1;
	if (mga_dbuf_reserve(&sc->wfa_cnt, 1024) < 0) return -1;
	hipStream_t st = (hipStream_t)mga_wfa_stream(sc, ws_slot);
	int *d_counter = (int*)((char*)sc->wfa_cnt.p + 64 * (ws_slot));
	MGA_HIP_CHECK(hipMemsetAsync(d_counter, 0, 4, st));
	mga_prof_begin(st, MGA_K_WFA0 + ws_slot);
	if (tier == 0)
		hipLaunchKernelGGL((k_wfa_regw<4, 1, 512, 512>), dim3(wgs), dim3(256), 0, st, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[ws_slot].p, d_counter, cfg);
	else if (tier == 1)
		hipLaunchKernelGGL((k_wfa_regw<4, 2, 1024, 1024>), dim3(wgs), dim3(256), 0, st, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[ws_slot].p, d_counter, cfg);
	else
		hipLaunchKernelGGL((k_wfa_regw<8, 2, 2048, 2048>), dim3(wgs), dim3(512), 0, st, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[ws_slot].p, d_counter, cfg);
	mga_prof_end(st, MGA_K_WFA0 + ws_slot);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
