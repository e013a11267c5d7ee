// dev_lcscan.h -- wave-level building blocks shared by the chaining kernels (k_lchain.hip: first pass + long-join rescue of a read; k_rmq.hip: the RMQ chainer's forward pass
// over the (segment, strand) runs of a contig): the reference's float helpers bit for bit, register scans by DPP, the replay of the predecessor scan's skip counter.
#ifndef MGA_DEV_LCSCAN_H
#define MGA_DEV_LCSCAN_H
#include "mga_dev.h"
#include "dev_common.h"

#define LC_NONE INT32_MIN

__device__ __forceinline__ float lc_log2(float x) // mgpriv.h:63-71
{
	uint32_t i = __float_as_uint(x);
	float r = (float)((int32_t)(i >> 23 & 255) - 128);
	i &= ~(255U << 23);
	i += 127U << 23;
	const float f = __uint_as_float(i);
	r += (-0.34484843f * f + 2.02466578f) * f - 0.67487759f;
	return r;
}

// ---- wave-wide inclusive scans in registers: row shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then the last lane of a row broadcast into the next row
// and lane 31 into the upper half (DPP; lanes without a source read the identity).  Six steps, no LDS crossbar (the __shfl_up form is a ds_bpermute per step).
#define LC_SCAN(name, OP) \
	__device__ __forceinline__ int32_t name(int32_t v, const int32_t ident) \
	{ \
		int32_t t; \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false); v = OP(v, t); \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false); v = OP(v, t); /* row_bcast:15 into rows 1 and 3 */ \
		t = __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false); v = OP(v, t); /* row_bcast:31 into rows 2 and 3 */ \
		return v; \
	}
#define LC_OP_ADD(a, b) ((a) + (b))
#define LC_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define LC_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
LC_SCAN(lc_scan_add, LC_OP_ADD)
LC_SCAN(lc_scan_min, LC_OP_MIN)
LC_SCAN(lc_scan_max, LC_OP_MAX)
__device__ __forceinline__ int32_t lc_prev_lane(int32_t v, int32_t first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); } // lane l <- v[l-1]; lane 0 <- first

// The skip counter of the predecessor scan (lchain.c:183-186), replayed for one block of 64 predecessors in visiting order = lane order: an improving predecessor
// takes one off the counter (never below 0), one that is already chained to a visited anchor and does not improve adds one, and the scan stops at the first
// such lane where the counter passes max_skip.  n_k = max(0, n_{k-1} + d_k) is a reflected walk: n_k = S_k - min(0, min_{j<=k} S_j) with S the running sum
// from the counter's value at the block's start -- two register scans instead of a scalar loop over the event lanes ([measured] round 1: 750 scalar
// instructions per anchor in this kernel against 370 vector ones; the scalar unit issues at the same rate per SIMD).
// Returns the lane of the cut (64: none) and leaves the counter's value behind the block in *n_skip.
__device__ __forceinline__ int lc_skip_replay(bool improve, bool hit, int32_t max_skip, int32_t *n_skip)
{
	const int32_t d = improve ? -1 : hit ? 1 : 0;
	const int32_t S = lc_scan_add(d, 0) + *n_skip;
	const int32_t M = lc_scan_min(S, 0x7fffffff);
	const int32_t nk = S - (M < 0 ? M : 0);
	const uint64_t m_cut = __ballot(hit && !improve && nk > max_skip);
	if (m_cut) return (int)__builtin_ctzll(m_cut);
	*n_skip = __builtin_amdgcn_readlane(nk, 63);
	return 64;
}

// comput_sc_simple (lchain.c:234-250)
__device__ __forceinline__ int32_t lc_score_simple(uint64_t xi, uint64_t yi, uint64_t xj, uint64_t yj, float pen_gap, float pen_skip, bool *exact, int32_t *width)
{
	const int32_t dq = (int32_t)yi - (int32_t)yj, dr = (int32_t)(xi - xj);
	const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
	const int32_t q_span = (int32_t)(yj >> 32 & 0xff);
	int32_t sc = q_span < dg ? q_span : dg;
	*width = dd, *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? lc_log2((float)(dd + 1)) : 0.0f;
		sc -= (int32_t)(lin + .5f * lg);
	}
	return sc;
}

#endif
