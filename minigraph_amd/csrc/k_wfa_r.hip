// k_wfa_r.hip -- register-resident exact 2-piece affine WFA: the gap filler's fast tiers (bands of 64..2048
// diagonals).  Bit-exact with the reference's exact mode (miniwfa.c:281-435: wf_next_tb, wf_extend1_padded,
// wf_stripe_shrink, wf_traceback); the HBM-resident kernel in k_wfa.hip takes what does not fit here.
//
// Layout.  A problem is owned by NW wavefronts (NW = 1: no barrier anywhere; NW = 4/8/16: one s_barrier per score).
// Wave w, lane l, slot j holds diagonal d = D0 + 64*J*w + 64*j + l.  Per diagonal 27 VGPRs: H of the last 17
// scores (the recurrence reads s-4, s-6, s-16), E1/F1 of the last 3 (s-2), E2/F2 of the last 2 (s-1), indexed by
// AGE so every access has a constant register index; a step shifts the ages instead of indexing a ring.
// Diagonals d-1 / d+1 are the neighbouring lanes: one DPP wave shift per operand whose "old" operand carries the
// value that crosses the slot / wave boundary (lane 63 of the previous slot via v_readlane, or the neighbouring
// wave's edge published through LDS, double-buffered by score parity).  Cells outside the current slice are
// NEG_INF, exactly what the reference's padded slices hold, so the slices need no bounds.
// The periodic trimming (every 256 scores the reference looks back over its 17-slice ring) uses one more
// register per diagonal: the last score at which the diagonal received an in-matrix value, maintained only
// during the 17 scores before a trimming point.
// Sequences sit in LDS (four byte-shifted copies, so that 8 bases at any offset are one aligned ds_read2_b32).  Traceback bytes (1 per
// cell) go to LDS for the first TBLDS cells of a problem and to an HBM scratch beyond that.
// A problem whose band leaves the 64*J*NW window, or outgrows the score / traceback tables, returns
// MGA_WFA_RETRY_TIER and is re-run by the next tier.
#include <type_traits>
#include "mga_dev.h"
#include "dev_common.h"
#include "wfa_window.h"

#define WF_NEG_INF (-0x40000000)

struct wfr_cfg_t {
	int32_t x, o1, e1, o2, e2;
	int32_t cigcap;
	int32_t tbcap; // HBM traceback bytes per workgroup (beyond the TBLDS bytes held in LDS)
	int64_t ws_stride;
};

// lane l <- src[l-1]; lane 0 keeps edge        (DPP wave_shr:1, bound_ctrl off)
__device__ __forceinline__ int32_t wfr_from_left(int32_t edge, int32_t src) { return __builtin_amdgcn_update_dpp(edge, src, 0x138, 0xf, 0xf, false); }
// lane l <- src[l+1]; lane 63 keeps edge       (DPP wave_shl:1)
__device__ __forceinline__ int32_t wfr_from_right(int32_t edge, int32_t src) { return __builtin_amdgcn_update_dpp(edge, src, 0x130, 0xf, 0xf, false); }

struct wfr_u2 { uint32_t x, y; }; // 8 bytes with 4-byte alignment: loads become ds_read2_b32

__device__ __forceinline__ int32_t wfr_max(int32_t a, int32_t b) { return a > b ? a : b; }

template<int NW, int J, int SEQCAP, int SMAX, int TBLDS, bool TBHBM>
__global__ void __launch_bounds__(64 * NW) k_wfa_r(const int *__restrict__ n_items_p, int cap, int first, const int32_t *__restrict__ list_,
												  const mga_wfa_prob_t *__restrict__ prob, const char *__restrict__ tseq, const char *__restrict__ qseq,
												  mga_wfa_res_t *__restrict__ res, uint32_t *__restrict__ pool, long long pool_cap, unsigned long long *pool_used,
												  char *__restrict__ ws_base, int *__restrict__ counter, mga_wfa_retry_t rt, wfr_cfg_t cfg)
{
	constexpr int NV = 64 * J * NW; // diagonals covered by the workgroup
	constexpr bool TRIM = SMAX >= 256; // the reference trims the band every 256 scores (miniwfa.c:139-169): tiers that stop earlier need none of it
	constexpr int NT = 64 * NW;
	// each sequence is staged four times, copy k shifted left by k bytes: any byte position is then dword-aligned in copy
	// (pos & 3), and 8 bytes come from one ds_read2_b32 (an off-alignment ds_read_b64 is replayed at ~64 cycles)
	constexpr int SEQS = SEQCAP + 16; // bytes per copy (multiple of 16)
	__shared__ __attribute__((aligned(16))) uint8_t Tb[4 * SEQS], Qb[4 * SEQS];
	__shared__ int32_t row[SMAX + 1];   // first traceback cell of score s
	__shared__ int16_t rlo[SMAX + 1];   // lowest diagonal of score s
	__shared__ uint8_t tb_lds[TBLDS];
	// xch[parity][wave+1]: [0..3] H[s-6], E1[s-2], H[s-16], E2[s-1] of the wave's LAST diagonal (read by the wave to
	// its right), [4..7] H[s-6], F1[s-2], H[s-16], F2[s-1] of its FIRST diagonal (read by the wave to its left);
	// rows 0 and NW+1 stay NEG_INF
	__shared__ __attribute__((aligned(16))) int32_t xch[2][NW + 2][8];
	__shared__ int32_t flags[6]; // [0] score+1 of the terminating slice, [1] its last state, [2+2p],[3+2p] "edge reachable" stamps
	__shared__ int32_t f_item, f_mn, f_mx;
	const int tid = threadIdx.x, lane = tid & 63;
	const int n_items = max(0, min(*n_items_p, cap) - first); // items first .. of the list (a rung's own problems run early, what arrives from below later)
	const int32_t *__restrict__ list = list_ + first;
	const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
	const int32_t oe1 = cfg.o1 + cfg.e1, oe2 = cfg.o2 + cfg.e2;
	char *wsb = ws_base + (size_t)blockIdx.x * cfg.ws_stride;
	uint32_t *cig = (uint32_t*)wsb;
	uint8_t *tbg = (uint8_t*)(cig + cfg.cigcap);
	const int32_t tbcap = TBLDS + (TBHBM ? cfg.tbcap : 0); // TBHBM false: the whole traceback fits LDS, no spill path is compiled
#define WFR_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory") /* LDS-only barrier: HBM traceback stores keep flying */
#define WFR_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

	constexpr int POOL_BLK = 512, QCHUNK_MAX = NW > 1 ? 4 : 8;
	// problems taken from the list per atomic: one when the list is shorter than the grid -- a tier above 256 diagonals sees a few hundred to a few thousand problems per launch,
	// each a millisecond of dependent score steps: handing them out four at a time left three quarters of the workgroups without work and made the launch last four problems
	const int QCHUNK = __builtin_amdgcn_readfirstlane(max(1, min(QCHUNK_MAX, n_items / (int)gridDim.x)));
	long long blk_beg = 0, blk_end = 0; // CIGAR pool block owned by wave 0
	int q_next = 0, q_end = 0;

	for (;;) {
		if (NW > 1) {
			__syncthreads(); // previous problem completely finished (its traceback reads LDS)
			if (q_next == q_end) {
				if (tid == 0) f_item = atomicAdd(counter, QCHUNK);
				__syncthreads();
				q_next = __builtin_amdgcn_readfirstlane(f_item), q_end = q_next + QCHUNK;
			}
		} else if (q_next == q_end) {
			int v = 0;
			if (lane == 0) v = atomicAdd(counter, QCHUNK);
			q_next = __builtin_amdgcn_readfirstlane(v), q_end = q_next + QCHUNK;
		}
		const int item = q_next++;
		if (item >= n_items) break;
		const int pi = list ? list[item] : item;
		const mga_wfa_prob_t pb = prob[pi];
		const int32_t tl = __builtin_amdgcn_readfirstlane(pb.tl), ql = __builtin_amdgcn_readfirstlane(pb.ql);
		int32_t status = MGA_WFA_OK, s = 0, wlo = 0, whi = 0, last_state = 0, clo = 0, chi = 0; // [clo,chi]: range of slice s
		int32_t tb_used = 1;

		if (tl > SEQCAP || ql > SEQCAP) status = MGA_WFA_RETRY_TIER;
		else {
			// Window of NV diagonals, centred between the start diagonal 0 and the end diagonal ql - tl (round 3, k_wfa_w.hip has the argument): the band follows
			// the reference's [wlo - 1, whi + 1] -- reachable edges, trimming every 256 scores -- but is CLIPPED to the window instead of leaving the tier when
			// it reaches the window's edge.  Clipped cells read NEG_INF: every value here is <= the reference's and equal on every cell of an optimal path as long
			// as the score stays below the window's bound BND (no optimal path can touch a diagonal outside); at BND the problem gives up and climbs.  The
			// round-2 tiers gave up when the BAND (2 s + 1 diagonals at score s) left the window, i.e. at half the score.
			int32_t D0;
			const int32_t BND = wfw_window(NV, tl, ql, &D0, 0x3fffffff);
			const int32_t W0 = D0 + 64 * J * wv; // first diagonal of this wave
			{ // stage the sequences; 16 bytes of padding so that the 8-byte compares may overrun
				const char *ts = tseq + pb.t_off, *qs = qseq + pb.q_off;
				if (NW == 1) WFR_LDS_FENCE(); // the previous problem's traceback has finished reading LDS (same wave, in order)
				for (int32_t i = tid; i < tl + 16; i += NT) {
					const uint8_t c = i < tl ? (uint8_t)ts[i] : (uint8_t)0;
					Tb[i] = c;
					if (i >= 1) Tb[SEQS + i - 1] = c;
					if (i >= 2) Tb[2 * SEQS + i - 2] = c;
					if (i >= 3) Tb[3 * SEQS + i - 3] = c;
				}
				for (int32_t i = tid; i < ql + 16; i += NT) {
					const uint8_t c = i < ql ? (uint8_t)qs[i] : (uint8_t)1;
					Qb[i] = c;
					if (i >= 1) Qb[SEQS + i - 1] = c;
					if (i >= 2) Qb[2 * SEQS + i - 2] = c;
					if (i >= 3) Qb[3 * SEQS + i - 3] = c;
				}
				if (tid == 0) { row[0] = 0; rlo[0] = 0; tb_lds[0] = 0; flags[0] = 0; flags[2] = flags[3] = flags[4] = flags[5] = -1; }
				if (NW > 1) {
					if (tid < 2 * (NW + 2) * 8) ((int32_t*)xch)[tid] = WF_NEG_INF;
					__syncthreads();
				} else WFR_LDS_FENCE();
			}
			int32_t H[J][18], E1[J][3], F1[J][3], E2[J][2], F2[J][2], GL[J], TBC[J];
#pragma unroll
			for (int j = 0; j < J; ++j) {
#pragma unroll
				for (int a = 0; a < 18; ++a) H[j][a] = WF_NEG_INF;
#pragma unroll
				for (int a = 0; a < 3; ++a) E1[j][a] = F1[j][a] = WF_NEG_INF;
#pragma unroll
				for (int a = 0; a < 2; ++a) E2[j][a] = F2[j][a] = WF_NEG_INF;
				GL[j] = -1, TBC[j] = 0;
				if (W0 + lane + 64 * j == 0) H[j][2] = -1, GL[j] = 0; // score 0: H[d=0] = -1 (age 0 of an even step sits at index 2)
			}

			// H of the last 16 scores is shifted by TWO registers every second step instead of by one every step: the step body exists twice
			// (P = 0: age a is H[a + 2] and the new slice goes to H[1]; P = 1: age a is H[a + 1], the new slice goes to H[0]), then
			// H[a] = H[a - 2].  16 moves per slot and pair of steps instead of 32.
#define HA(j_, a_) H[j_][(a_) + 2 - P]
			auto step = [&](auto Pc) __attribute__((always_inline)) -> bool { // false: the problem is finished (or has to leave the tier)
				constexpr int P = decltype(Pc)::value;
				const int par = s & 1;
				// Scores that no alignment can have under x=4, o1+e1=6, e1=2, o2+e2=16, e2=1 (the penalties the register ages are built for): 1, 2, 3 and the
				// odd ones below 17.  Their slices hold nothing but unreachable cells in all five arrays, whatever the sequences: the reference computes
				// them anyway (NEG_INF plus a few), here the step only keeps the books -- traceback rows, band, ages -- and writes NEG_INF.  Nothing a
				// reachable cell or its traceback byte depends on differs: an unreachable operand loses every max against a reachable one either way.
				const bool empty_cur = s < 16 && ((0xAAAEu >> s) & 1u), empty_next = s + 1 < 16 && ((0xAAAEu >> (s + 1)) & 1u);
				// ---- extension of slice s (miniwfa.c:399-411); the end cell lies on the unique diagonal ql - tl
				bool term = false;
#pragma unroll
				for (int j = 0; j < J; ++j) {
					const int32_t b0 = W0 + 64 * j;
					if (empty_cur || b0 > chi || b0 + 63 < clo) continue; // nothing to extend / slot entirely outside the slice (uniform)
					const int32_t d = b0 + lane, k0 = HA(j, 0), i0 = d + k0;
					const bool val = (uint32_t)(k0 + 1) <= (uint32_t)tl && (uint32_t)(i0 + 1) <= (uint32_t)ql; // -1 <= k0 < tl, -1 <= i0 < ql
					const int32_t tp = val ? k0 + 1 : 0, qp = val ? i0 + 1 : 0;
					const int32_t room = min(tl - tp, ql - qp);
					const wfr_u2 *tw = (const wfr_u2*)(Tb + (tp & 3) * SEQS + (tp & ~3)), *qw = (const wfr_u2*)(Qb + (qp & 3) * SEQS + (qp & ~3)); // 4-byte aligned
					int32_t n = 0, m8 = 0; // m8: matched blocks of 8 bases
					bool act = val && room > 0;
					while (__ballot(act)) { // uniform loop, no divergent branch inside: finished lanes reload their last block and add nothing
						const wfr_u2 a = tw[m8], b = qw[m8];
						const uint32_t c0 = a.x ^ b.x, c1 = a.y ^ b.y;
						const int32_t e0 = (int32_t)((c0 ? (uint32_t)__builtin_ctz(c0) : 32u) >> 3), e1 = (int32_t)((c1 ? (uint32_t)__builtin_ctz(c1) : 32u) >> 3); // equal leading bytes of each half: 0..4
						const int32_t adv = e0 < 4 ? e0 : 4 + e1; // of the block: 0..8
						n += act ? adv : 0;
						m8 += (act && adv == 8) ? 1 : 0;
						act = act && adv == 8 && n < room;
					}
					n = min(n, room);
					const int32_t k = k0 + n;
					HA(j, 0) = val ? k : k0;
					const uint64_t m = __ballot(val && k == tl - 1 && d + k == ql - 1);
					if (m) {
						const int32_t ls = __shfl(n == 0 ? (TBC[j] & 7) : 0, (int)__builtin_ctzll(m));
						if (NW == 1) { term = true; last_state = ls; }
						else if (lane == 0) { flags[1] = ls; flags[0] = s + 1; }
					}
				}
				if (NW == 1 && term) return false;
				// ---- slice s+1 (miniwfa.c:281-325,412-415).  With NW > 1 it is computed speculatively: the terminating
				// wave cannot tell the others before the barrier.
				const int32_t nlo = max(wlo > -tl ? wlo - 1 : -tl, D0);
				const int32_t nhi = min(whi < ql ? whi + 1 : ql, D0 + NV - 1);
				const int32_t width = nhi - nlo + 1;
				const bool fits = !(s + 1 >= BND || s + 1 > SMAX || tb_used + width > tbcap);
				if (NW == 1 && !fits) { status = MGA_WFA_RETRY_TIER; return false; }
				bool reach_lo = false, reach_hi = false; // uniform
				if (fits) {
					const bool track_alive = TRIM && (((s + 1) & 0xff) >= 239 || ((s + 1) & 0xff) == 0); // the trimming at score 256k looks back 17 scores only
					if (tid == 0) { row[s + 1] = tb_used; rlo[s + 1] = (int16_t)nlo; }
					int32_t eL[4], eR[4]; // what the neighbouring waves published after the previous step
					if (NW > 1) {
						const int4 l4 = *(const int4*)&xch[par][wv][0], r4 = *(const int4*)&xch[par][wv + 2][4];
						eL[0] = l4.x, eL[1] = l4.y, eL[2] = l4.z, eL[3] = l4.w;
						eR[0] = r4.x, eR[1] = r4.y, eR[2] = r4.z, eR[3] = r4.w;
					} else {
#pragma unroll
						for (int q = 0; q < 4; ++q) eL[q] = eR[q] = WF_NEG_INF;
					}
					int32_t nH[J], nE1[J], nF1[J], nE2[J], nF2[J];
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t b0 = W0 + 64 * j;
						if (empty_next || b0 > nhi || b0 + 63 < nlo) { nH[j] = nE1[j] = nF1[j] = nE2[j] = nF2[j] = WF_NEG_INF; continue; } // uniform
						const int32_t d = b0 + lane;
						// predecessors: score s+1-p is age p-1 now (ages are shifted at the end of the step)
#define WFR_L(R, a, q) wfr_from_left(j > 0 ? __builtin_amdgcn_readlane(R[j > 0 ? j - 1 : 0][a], 63) : eL[q], R[j][a])
#define WFR_R(R, a, q) wfr_from_right(j < J - 1 ? __builtin_amdgcn_readlane(R[j < J - 1 ? j + 1 : j][a], 0) : eR[q], R[j][a])
						const int32_t ho1l = WFR_L(H, 5 + 2 - P, 0), e1l = WFR_L(E1, 1, 1), ho2l = WFR_L(H, 15 + 2 - P, 2), e2l = WFR_L(E2, 0, 3);
						const int32_t ho1r = WFR_R(H, 5 + 2 - P, 0), f1r = WFR_R(F1, 1, 1), ho2r = WFR_R(H, 15 + 2 - P, 2), f2r = WFR_R(F2, 0, 3);
#undef WFR_L
#undef WFR_R
						const int32_t hx1 = HA(j, 3) + 1;
						int32_t vE1 = wfr_max(ho1l, e1l), vE2 = wfr_max(ho2l, e2l);
						int32_t vF1 = wfr_max(ho1r, f1r) + 1, vF2 = wfr_max(ho2r, f2r) + 1;
						uint32_t bits = (ho1l < e1l ? 0x08u : 0u) | (ho2l < e2l ? 0x20u : 0u) | (ho1r < f1r ? 0x10u : 0u) | (ho2r < f2r ? 0x40u : 0u);
						const int32_t e = wfr_max(vE1, vE2), f = wfr_max(vF1, vF2), h = wfr_max(e, f);
						const uint32_t ze = vE1 >= vE2 ? 1u : 3u, zf = vF1 >= vF2 ? 2u : 4u;
						uint32_t z = e >= f ? ze : zf;
						z = hx1 >= h ? 0u : z;
						int32_t vH = wfr_max(hx1, h);
						const uint32_t rel = (uint32_t)(d - nlo);
						const bool inb = rel <= (uint32_t)(nhi - nlo);
						const uint32_t tbv = bits | z;
						TBC[j] = (int32_t)tbv;
						if (inb) {
							const int32_t off = tb_used + (int32_t)rel;
							if (!TBHBM || off < TBLDS) tb_lds[off] = (uint8_t)tbv;
							else tbg[off - TBLDS] = (uint8_t)tbv;
						}
						const int32_t top = wfr_max(wfr_max(wfr_max(vH, vE1), wfr_max(vF1, vE2)), vF2);
						const uint64_t rm = __ballot(inb && top >= -1); // reachable cells of the new slice
						if (nlo >= b0 && nlo < b0 + 64) reach_lo = (rm >> (nlo - b0)) & 1;
						if (nhi >= b0 && nhi < b0 + 64) reach_hi = (rm >> (nhi - b0)) & 1;
						if (track_alive) {
#define WFR_IN(k_) ((uint32_t)((k_) + 1) <= (uint32_t)tl && (uint32_t)(d + (k_) + 1) <= (uint32_t)ql)
							if (inb && (WFR_IN(vH) || WFR_IN(vE1) || WFR_IN(vF1) || WFR_IN(vE2) || WFR_IN(vF2))) GL[j] = s + 1;
#undef WFR_IN
						}
						nH[j] = inb ? vH : WF_NEG_INF, nE1[j] = inb ? vE1 : WF_NEG_INF, nF1[j] = inb ? vF1 : WF_NEG_INF;
						nE2[j] = inb ? vE2 : WF_NEG_INF, nF2[j] = inb ? vF2 : WF_NEG_INF;
					}
#pragma unroll
					for (int j = 0; j < J; ++j) { // age shift (H: every second step, by two)
						HA(j, -1) = nH[j];
						if (P == 1) {
#pragma unroll
							for (int a = 17; a > 1; --a) H[j][a] = H[j][a - 2];
						}
						E1[j][2] = E1[j][1]; E1[j][1] = E1[j][0]; E1[j][0] = nE1[j];
						F1[j][2] = F1[j][1]; F1[j][1] = F1[j][0]; F1[j][0] = nF1[j];
						E2[j][1] = E2[j][0]; E2[j][0] = nE2[j];
						F2[j][1] = F2[j][0]; F2[j][0] = nF2[j];
					}
					if (NW > 1) { // publish what the neighbours read in the NEXT step (post-shift ages) into the other parity's buffers
						// (after this step's shift the next step is the other copy: its age a is H[a + 2 - (1 - P)])
						if (lane == 63) *(int4*)&xch[par ^ 1][wv + 1][0] = make_int4(H[J - 1][5 + 1 + P], E1[J - 1][1], H[J - 1][15 + 1 + P], E2[J - 1][0]);
						if (lane == 0) {
							*(int4*)&xch[par ^ 1][wv + 1][4] = make_int4(H[0][5 + 1 + P], F1[0][1], H[0][15 + 1 + P], F2[0][0]);
							if (reach_lo) flags[2 + 2 * (par ^ 1)] = s + 1;
							if (reach_hi) flags[3 + 2 * (par ^ 1)] = s + 1;
						}
					}
				}
				if (NW > 1) {
					WFR_BAR();
					// "== s + 1": a faster wave may already have stamped the NEXT slice as terminating
					if (flags[0] == s + 1) { last_state = flags[1]; return false; } // slice s reached the end: the speculative slice is dropped
					if (!fits) { status = MGA_WFA_RETRY_TIER; return false; }
					reach_lo = flags[2 + 2 * (par ^ 1)] == s + 1;
					reach_hi = flags[3 + 2 * (par ^ 1)] == s + 1;
				}
				++s;
				tb_used += width;
				clo = nlo, chi = nhi;
				if (reach_lo) wlo = nlo;
				if (reach_hi) whi = nhi;
				if (TRIM && (s & 0xff) == 0) { // trimming (miniwfa.c:139-169): keep [first, last] diagonal that was in the matrix during the last 17 scores
					int32_t mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
					for (int j = 0; j < J; ++j) {
						const int32_t d = W0 + lane + 64 * j;
						const uint64_t m = __ballot(d >= wlo && d <= whi && GL[j] > s - 17);
						if (m) {
							const int32_t first = W0 + 64 * j + (int32_t)__builtin_ctzll(m), last = W0 + 64 * j + 63 - (int32_t)__clzll(m);
							if (first < mn) mn = first;
							if (last > mx) mx = last;
						}
					}
					if (NW > 1) {
						if (tid == 0) { f_mn = 0x7fffffff; f_mx = -0x7fffffff; }
						WFR_BAR();
						if (lane == 0 && mn != 0x7fffffff) { atomicMin(&f_mn, mn); atomicMax(&f_mx, mx); }
						WFR_BAR();
						mn = f_mn, mx = f_mx; // (the next reset of f_mn is 256 barriers away)
					}
					if (mn != 0x7fffffff) wlo = mn, whi = mx;
					else { const int32_t e0 = whi + 1; wlo = e0; whi = e0 - 1; }
				}
				return true;
			};
			for (;;) {
				if (!step(std::integral_constant<int, 0>())) break;
				if (!step(std::integral_constant<int, 1>())) break;
			}
#undef HA
		}
		if (NW > 1) __syncthreads(); // HBM traceback rows complete and visible to wave 0
		else { WFR_LDS_FENCE(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

		// ---- traceback by wave 0 (miniwfa.c:329-377); runs of matches are found 64 bases at a time
		if (wv == 0) {
			int32_t n_cig = 0;
			int64_t cig_off = 0;
			if (status == MGA_WFA_OK) {
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
					const int32_t off = row[sc] + ((i - k) - (int32_t)rlo[sc]);
					const uint32_t x = (!TBHBM || off < TBLDS) ? tb_lds[off] : tbg[off - TBLDS];
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
					if (blk_end - blk_beg < n_cig) { // the wave takes CIGAR space from the pool one block at a time
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
				r.cig_off = cig_off, r.status = status, r.pad = 0, r.n_iter = tb_used - 1;
				res[pi] = r;
				if (status == MGA_WFA_RETRY_TIER) mga_wfa_give_up(rt, pi); // next tier's work list
				else if (status != MGA_WFA_OK) atomicAdd(rt.err, 1);
			}
		}
	}
}

// ---- host driver ---------------------------------------------------------------------------------

struct wfr_tier_t { int n_wg; int32_t cigcap, tbcap; };
static const wfr_tier_t g_rtier[7] = {
	//  workgroups  cigcap  HBM traceback scratch per workgroup
	{ 8192,    512,        0 },   // 1 wave  x 1 slot :   64 diagonals, traceback in LDS only
	{ 6144,   1024,     8192 },   // 1 wave  x 2 slots:  128
	{ 6144,   2048,    32768 },   // 1 wave  x 3 slots:  192
	{ 4096,   2048, 192 << 10 },  // 2 waves x 2 slots:  256
	{ 1280,   4096, 768 << 10 },  // 4 waves x 2 slots:  512
	{  512,   8192,   3 << 20 },  // 8 waves x 2 slots: 1024
	{   64,  16384,  12 << 20 },  // 16 waves x 2 slots: 2048 (a handful of problems per 10^5 reads; keeps them off the slow HBM kernel)
};

extern "C" int mga_dev_wfa_reg(mga_sctx_t *sc, const int *d_n, int n, int first, int slot, void *stream, const int32_t *d_list, const mga_wfa_prob_t *d_prob, const char *d_tseq, const char *d_qseq,
							   mga_wfa_res_t *d_res, uint32_t *d_pool, int64_t pool_cap, unsigned long long *d_pool_used, int tier, mga_wfa_retry_t rt)
{
	if (n <= 0) return 0;
	if (tier < 0 || tier > 6) { mga_set_error("wfa_reg: bad tier %d", tier); return -1; }
	const wfr_tier_t &T = g_rtier[tier];
	wfr_cfg_t cfg = { 4, 4, 2, 15, 1, T.cigcap, T.tbcap, 0 }; // register ages 17/3/2 are tied to these penalties (miniwfa.c:11-18)
	cfg.ws_stride = (int64_t)(((size_t)T.cigcap * 4 + (size_t)T.tbcap + 255) & ~(size_t)255);
	int wgs = T.n_wg < n - first ? T.n_wg : n - first; // (n: the list's capacity; the kernel sizes its queue chunks by the list's real length)
	if (wgs < 1) wgs = 1;
	if (mga_dbuf_reserve(&sc->wfa_ws[tier], (size_t)cfg.ws_stride * T.n_wg) < 0) return -1;
	hipStream_t st = (hipStream_t)(stream ? stream : sc->stream);
	int *d_counter = (int*)((char*)sc->wfa_cnt.p + 64 * slot);
	mga_prof_begin(st, MGA_K_WFA0 + tier);
#define LAUNCH(NW, JJ, SEQ, SM, TBL, HBM) hipLaunchKernelGGL((k_wfa_r<NW, JJ, SEQ, SM, TBL, HBM>), dim3(wgs), dim3(64 * NW), 0, st, d_n, n, first, d_list, d_prob, d_tseq, d_qseq, d_res, d_pool, (long long)pool_cap, d_pool_used, (char*)sc->wfa_ws[tier].p, d_counter, rt, cfg)
	if (tier == 0) LAUNCH(1, 1, 128, 64, 2048, false); // (g_rtier[0].tbcap == 0: traceback in LDS only)
	else if (tier == 1) LAUNCH(1, 2, 256, 128, 4096, true);
	else if (tier == 2) LAUNCH(1, 3, 256, 192, 6144, true); // [measured] 60 ns per problem against 87 ns in the two-wave 256-diagonal tier; the centre slot alone holds the first 32 scores
	else if (tier == 3) LAUNCH(2, 2, 512, 512, 8192, true);
	else if (tier == 4) LAUNCH(4, 2, 1024, 1024, 8192, true);
	else if (tier == 5) LAUNCH(8, 2, 2048, 2048, 8192, true);
	else LAUNCH(16, 2, 4096, 4096, 8192, true);
#undef LAUNCH
	mga_prof_end(st, MGA_K_WFA0 + tier);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
