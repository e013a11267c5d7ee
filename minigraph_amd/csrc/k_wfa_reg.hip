// k_wfa_reg.hip -- register-resident exact 2-piece affine WFA: the fastest tiers of the gap filler.
//
// Same algorithm and bit-exact semantics as k_wfa.hip (reference miniwfa.c:281-435).  One wavefront
// per problem and NO workgroup barrier anywhere: the whole wavefront state lives in VGPRs.
//
//   * lane l, slot j holds diagonal d = D0 + l + 64*j (J slots -> 64*J diagonals, J = 1,2,4,8);
//   * per diagonal 27 registers: H of the last 17 scores (the recurrence reads s-4, s-6, s-16), E1/F1 of
//     the last 3 (s-2), E2/F2 of the last 2 (s-1), indexed by AGE so that every access has a constant
//     index; one step shifts the ages (v_mov) instead of indexing a ring;
//   * diagonals d-1 / d+1 are the neighbouring lanes: wave shuffles, with the slot boundary patched
//     from lane 63 / lane 0 of the adjacent slot;
//   * cells outside the current slice are kept at NEG_INF, which is exactly what the reference's padded
//     slices return, so no per-slice bounds are needed;
//   * the periodic trimming uses one more register per diagonal: the last score at which the diagonal
//     received an in-matrix value (see k_wfa_lds.hip);
//   * sequences are staged in LDS; traceback bytes go to LDS (J <= 2) or to an HBM scratch.
//
// A problem whose band leaves the 64*J-diagonal window (or outgrows the traceback / score tables)
// returns MGA_WFA_RETRY_TIER and is re-run by the next tier.
#include "mga_dev.h"
#include "dev_common.h"

#define WF_NEG_INF (-0x40000000)

struct wfr_cfg_t {
	int32_t x, o1, e1, o2, e2;
	int32_t cigcap;
	int64_t tbcap, max_iter, ws_stride;
};

__device__ __forceinline__ uint32_t wfr_load4(const uint32_t *w, int32_t p)
{
	const int32_t i = p >> 2;
	return __funnelshift_r(w[i], w[i + 1], (p & 3) << 3);
}

__device__ __forceinline__ int32_t wfr_lcp(const uint32_t *t, int32_t tp, const uint32_t *q, int32_t qp, int32_t maxlen)
{
	int32_t n = 0;
	while (n < maxlen) {
		const uint32_t c = wfr_load4(t, tp + n) ^ wfr_load4(q, qp + n);
		if (c) { n += __builtin_ctz(c) >> 3; break; }
		n += 4;
	}
	return n < maxlen ? n : maxlen;
}

template<int J, int SEQCAP, int SMAX, int TBLDS>
__global__ void __launch_bounds__(64) k_wfa_reg(int n_items, const int32_t *__restrict__ list,
												const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq, const char *__restrict__ qseq,
												mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used,
												char *__restrict__ ws_base, int *__restrict__ counter, wfr_cfg_t cfg)
{
	constexpr int NV = 64 * J;
	constexpr int SEQW = (SEQCAP + 16) / 4;
	__shared__ uint32_t Ts[SEQW], Qs[SEQW];
	__shared__ int32_t row[SMAX + 1];
	__shared__ int16_t rlo[SMAX + 1];
	__shared__ uint8_t tb_lds[TBLDS > 0 ? TBLDS : 4];
	const int lane = threadIdx.x;
	const int32_t oe1 = cfg.o1 + cfg.e1, oe2 = cfg.o2 + cfg.e2;
	char *wsb = ws_base + (size_t)blockIdx.x * cfg.ws_stride;
	uint32_t *cig = (uint32_t*)wsb;
	uint8_t *tb = TBLDS > 0 ? tb_lds : (uint8_t*)(cig + cfg.cigcap);
	const int64_t tbcap = TBLDS > 0 ? (int64_t)TBLDS : cfg.tbcap;
#define WFR_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

	constexpr int POOL_BLK = 512, QCHUNK = 8;
	long long blk_beg = 0, blk_end = 0;
	int q_next = 0, q_end = 0;

	for (;;) {
		if (q_next == q_end) {
			int v = 0;
			if (lane == 0) v = atomicAdd(counter, QCHUNK);
			q_next = __shfl(v, 0), q_end = q_next + QCHUNK;
		}
		const int item = q_next++;
		if (item >= n_items) break;
		const int pi = list ? list[item] : item;
		const mga_wfa_prob_t pb = prob[pi];
		const int32_t tl = pb.tl, ql = pb.ql;
		int32_t status = MGA_WFA_OK, s = 0, wlo = 0, whi = 0, last_state = 0, clo = 0, chi = 0; // [clo,chi]: range of the current slice
		int64_t n_iter = 0, tb_used = 1;

		if (tl > SEQCAP || ql > SEQCAP) status = MGA_WFA_RETRY_TIER;
		else {
			// window of 64*J diagonals, centred on 0 unless the matrix is narrower on one side
			int32_t D0 = -(NV / 2);
			if (-tl > D0) D0 = -tl;
			else if (ql < D0 + NV - 1) { D0 = ql - NV + 1; if (D0 < -tl) D0 = -tl; }
			{ // stage the sequences
				const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
				uint8_t *Tb = (uint8_t*)Ts, *Qb = (uint8_t*)Qs;
				WFR_LDS_FENCE(); // the previous problem's traceback has finished reading LDS (same wave, in order)
				for (int32_t i = lane; i < tl + 8; i += 64) Tb[i] = i < tl ? (uint8_t)ts[i] : (uint8_t)0;
				for (int32_t i = lane; i < ql + 8; i += 64) Qb[i] = i < ql ? (uint8_t)qs[i] : (uint8_t)1;
				if (lane == 0) { row[0] = 0; rlo[0] = 0; tb[0] = 0; }
				WFR_LDS_FENCE();
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
				if (D0 + lane + 64 * j == 0) H[j][0] = -1, GL[j] = 0; // score 0: H[d=0] = -1
			}

			for (;;) {
				// ---- extension of the current slice (miniwfa.c:399-411)
				uint64_t m_term = 0;
				int32_t ls = 0;
#pragma unroll
				for (int j = 0; j < J; ++j) {
					if (D0 + 64 * j > chi || D0 + 64 * j + 63 < clo) continue; // slot entirely outside the current slice (uniform)
					const int32_t d = D0 + lane + 64 * j, k0 = H[j][0];
					bool term = false;
					if (!(k0 < -1 || d + k0 < -1 || k0 >= tl || d + k0 >= ql)) {
						int32_t room = tl - (k0 + 1);
						if (ql - (d + k0 + 1) < room) room = ql - (d + k0 + 1);
						const int32_t k = k0 + wfr_lcp(Ts, k0 + 1, Qs, d + k0 + 1, room);
						term = (k == tl - 1 && d + k == ql - 1);
						if (term) ls = k == k0 ? (TBC[j] & 7) : 0;
						else H[j][0] = k;
					}
					const uint64_t m = __ballot(term);
					if (m) m_term = m;
				}
				if (m_term) { last_state = __shfl(ls, __builtin_ctzll(m_term)); break; }
				// ---- next slice (miniwfa.c:281-325,412-415)
				const int32_t nlo = wlo > -tl ? wlo - 1 : -tl;
				const int32_t nhi = whi < ql ? whi + 1 : ql;
				const int32_t width = nhi - nlo + 1;
				if (nlo < D0 || nhi > D0 + NV - 1 || s + 1 > SMAX || tb_used + width > tbcap) { status = MGA_WFA_RETRY_TIER; break; }
				++s;
				const bool track_alive = (s & 0xff) >= 239 || (s & 0xff) == 0; // the trimming at score 256k looks back 17 scores only
				if (lane == 0) { row[s] = (int32_t)tb_used; rlo[s] = (int16_t)nlo; }
				int32_t nH[J], nE1[J], nF1[J], nE2[J], nF2[J];
				bool reach_lo = false, reach_hi = false;
#pragma unroll
				for (int j = 0; j < J; ++j) {
					if (D0 + 64 * j > nhi || D0 + 64 * j + 63 < nlo) { // slot entirely outside the new slice (uniform): all NEG_INF
						nH[j] = nE1[j] = nF1[j] = nE2[j] = nF2[j] = WF_NEG_INF;
						continue;
					}
					const int32_t d = D0 + lane + 64 * j;
					// predecessors: score s-p is age p-1 now (ages are shifted at the end of the step)
#define WFR_LEFT(R)  ({ int32_t u_ = __shfl_up((R)[j], 1); const int32_t w_ = j > 0 ? __shfl((R)[j > 0 ? j - 1 : 0], 63) : WF_NEG_INF; lane == 0 ? w_ : u_; })
#define WFR_RIGHT(R) ({ int32_t u_ = __shfl_down((R)[j], 1); const int32_t w_ = j < J - 1 ? __shfl((R)[j < J - 1 ? j + 1 : j], 0) : WF_NEG_INF; lane == 63 ? w_ : u_; })
					int32_t Ho1[J], Ho2[J], E1p[J], F1p[J], E2p[J], F2p[J];
#pragma unroll
					for (int jj = 0; jj < J; ++jj) { Ho1[jj] = H[jj][5]; Ho2[jj] = H[jj][15]; E1p[jj] = E1[jj][1]; F1p[jj] = F1[jj][1]; E2p[jj] = E2[jj][0]; F2p[jj] = F2[jj][0]; }
					const int32_t ho1l = WFR_LEFT(Ho1), e1l = WFR_LEFT(E1p), ho2l = WFR_LEFT(Ho2), e2l = WFR_LEFT(E2p);
					const int32_t ho1r = WFR_RIGHT(Ho1), f1r = WFR_RIGHT(F1p), ho2r = WFR_RIGHT(Ho2), f2r = WFR_RIGHT(F2p);
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
					const bool in = d >= nlo && d <= nhi;
					if (in) {
						tb[tb_used + (d - nlo)] = (uint8_t)(bits | z);
						TBC[j] = (int32_t)(bits | z);
						const bool reach = vH >= -1 || vE1 >= -1 || vF1 >= -1 || vE2 >= -1 || vF2 >= -1;
						if (d == nlo) reach_lo = reach;
						if (d == nhi) reach_hi = reach;
#define WFR_IN(k_) ((k_) >= -1 && (k_) < tl && d + (k_) >= -1 && d + (k_) < ql)
						if (track_alive && (WFR_IN(vH) || WFR_IN(vE1) || WFR_IN(vF1) || WFR_IN(vE2) || WFR_IN(vF2))) GL[j] = s;
#undef WFR_IN
					} else vH = vE1 = vF1 = vE2 = vF2 = WF_NEG_INF; // outside the slice: what the padded reference slices hold
					nH[j] = vH, nE1[j] = vE1, nF1[j] = vF1, nE2[j] = vE2, nF2[j] = vF2;
				}
#undef WFR_LEFT
#undef WFR_RIGHT
				// age shift
#pragma unroll
				for (int j = 0; j < J; ++j) {
#pragma unroll
					for (int a = 16; a > 0; --a) H[j][a] = H[j][a - 1];
					H[j][0] = nH[j];
					E1[j][2] = E1[j][1]; E1[j][1] = E1[j][0]; E1[j][0] = nE1[j];
					F1[j][2] = F1[j][1]; F1[j][1] = F1[j][0]; F1[j][0] = nF1[j];
					E2[j][1] = E2[j][0]; E2[j][0] = nE2[j];
					F2[j][1] = F2[j][0]; F2[j][0] = nF2[j];
				}
				tb_used += width;
				clo = nlo, chi = nhi;
				if (__ballot(reach_lo)) wlo = nlo;
				if (__ballot(reach_hi)) whi = nhi;
				if ((s & 0xff) == 0) { // trimming (miniwfa.c:139-169)
					int32_t mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t d = D0 + lane + 64 * j;
						const uint64_t m = __ballot(d >= wlo && d <= whi && GL[j] > s - 17);
						if (m) {
							const int32_t first = D0 + 64 * j + __builtin_ctzll(m), last = D0 + 64 * j + 63 - __clzll(m);
							if (first < mn) mn = first;
							if (last > mx) mx = last;
						}
					}
					if (mn != 0x7fffffff) wlo = mn, whi = mx;
					else { const int32_t e0 = whi + 1; wlo = e0; whi = e0 - 1; }
				}
				n_iter += width;
				if (cfg.max_iter > 0 && n_iter > cfg.max_iter) { status = MGA_WFA_MAX_ITER; break; }
			}
		}

		// ---- traceback (miniwfa.c:329-377), wave-cooperative
		int32_t n_cig = 0;
		int64_t cig_off = 0;
		if (status == MGA_WFA_OK) {
			if (TBLDS > 0) WFR_LDS_FENCE();
			else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } // HBM traceback rows are complete
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

// ---- host driver ---------------------------------------------------------------------------------

struct wfr_tier_t { int j, n_wave; int32_t cigcap; int64_t tbcap; };
static const wfr_tier_t g_rtier[4] = {
	// J  waves  cigcap  HBM traceback scratch per wave (0: traceback in LDS)
	{ 1,  8192,   512,   0 },
	{ 2,  6144,  1024,   0 },
	{ 4,  4096,  2048,   192 << 10 },
	{ 8,  2048,  4096,   768 << 10 },
};

extern "C" int mga_dev_wfa_reg(mga_sctx_t *sc, int n, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
							   mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier)
{
	if (n <= 0) return 0;
	if (tier < 0 || tier > 3) { mga_set_error("wfa_reg: bad tier %d", tier); return -1; }
	const wfr_tier_t &T = g_rtier[tier];
	wfr_cfg_t cfg = { 4, 4, 2, 15, 1, T.cigcap, T.tbcap, 100000000, 0 }; // register ages 17/3/2 are tied to these penalties (miniwfa.c:11-18)
	cfg.ws_stride = (int64_t)(((size_t)T.cigcap * 4 + (size_t)T.tbcap + 255) & ~(size_t)255);
	int waves = T.n_wave < (n + 7) / 8 ? T.n_wave : (n + 7) / 8;
	if (waves < 1) waves = 1;
	if (mga_dbuf_reserve(&sc->wfa_ws[tier], (size_t)cfg.ws_stride * T.n_wave) < 0) return -1;
	if (mga_dbuf_reserve(&sc->wfa_cnt, 1024) < 0) return -1;
	hipStream_t st = (hipStream_t)mga_wfa_stream(sc, tier);
	int *d_counter = (int*)((char*)sc->wfa_cnt.p + 64 * (tier));
	MGA_HIP_CHECK(hipMemsetAsync(d_counter, 0, 4, st));
	mga_prof_begin(st, MGA_K_WFA0 + tier);
#define LAUNCH(JJ, SEQ, SM, TBL) hipLaunchKernelGGL((k_wfa_reg<JJ, SEQ, SM, TBL>), dim3(waves), dim3(64), 0, st, n, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[tier].p, d_counter, cfg)
	if (tier == 0) LAUNCH(1, 128, 64, 2048);
	else if (tier == 1) LAUNCH(2, 256, 128, 6144);
	else { mga_set_error("wfa_reg: bands above 128 diagonals run on the multi-wave kernel (k_wfa_regw.hip)"); return -1; }
#undef LAUNCH
	mga_prof_end(st, MGA_K_WFA0 + tier);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
