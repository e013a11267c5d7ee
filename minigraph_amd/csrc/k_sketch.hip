// k_sketch.hip -- (w,k)-minimizer sketch of a batch of sequences, one wavefront per sequence.
//
// Replaces mg_sketch() (reference sketch.c:56-109, hash64 at :28-38) for a whole batch.  The
// reference is a ring-buffer state machine; here it runs in its position-parallel "event timeline"
// form (derivation pinned against the reference in oracle/mgo_sketch.c):
//
//   * 64 consecutive bases per step, one per lane, read coalesced from HBM.
//   * non-ambiguous bases are compacted into an LDS code ring (2-bit base + its complement); every
//     lane rebuilds its forward / reverse k-mer from the last k ring entries, so ambiguous bases and
//     the symmetric-k-mer skip (sketch.c:76) need no sequential carry.
//   * surviving bases become events t (wave ballot + popcount = prefix sum); run length l since the
//     last ambiguous base comes from a ballot of the ambiguous lanes.
//   * events go to an LDS ring (x, y, l); each lane scans its window of w+1 entries for the rightmost
//     minimum before/after its event and derives what the reference would push at that step.
//   * emission counts are prefix-summed across the wave so output order equals the reference's.
//
// Two passes over the same kernel: count (d_mz == NULL) and write -- or ONE pass when the caller provides per-read
// capacities instead of exact offsets (the mapping pipeline: qlen/2 + 64 slots per read, ~3x the minimizer density).
#include "mga_dev.h"
#include "dev_common.h"
#include "dev_lcscan.h"
#include <stdlib.h>

#define SK_RING   512   // event ring (>= 255 window look-back + 64 new events)
#define SK_CRING  128   // code ring  (>= 27 k-mer look-back + 64 new codes)

__device__ __forceinline__ uint64_t sk_hash64(uint64_t key, uint64_t mask) // sketch.c:28-38
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

__device__ __forceinline__ int sk_nt4(unsigned char ch) // seq_nt4_table, sketch.c:9-26
{
	unsigned char u = ch & 0xdf; // fold case
	return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : (u == 'T' || u == 'U') ? 3 : 4;
}

__global__ void __launch_bounds__(64) k_sketch_v1(int n, const char *__restrict__ seq, const int64_t *__restrict__ off,
											   const uint32_t *__restrict__ rid_arr, int w, int k,
											   int32_t *__restrict__ cnt, const int64_t *__restrict__ mz_off, mg128_t *__restrict__ mz,
											   const int4 *__restrict__ items)
{
	__shared__ uint64_t ex[SK_RING], ey[SK_RING];
	__shared__ int32_t el[SK_RING];
	__shared__ uint8_t codes[SK_CRING];

	// Work item = a whole sequence, or (items != NULL) the piece [it.y, it.z) of sequence it.x.  A piece warms up on the w+k+64
	// bases before it -- every decision below looks back at most w events and k bases, and the run-length conditions
	// saturate at w+k -- and only emits for its own bases; the launcher cuts long sequences only when k is odd (no
	// symmetric k-mers, so every base is an event and the warm-up length is exact).
	const int r_item = blockIdx.x;
	if (r_item >= n) return;
	const int lane = threadIdx.x;
	const int r = items ? items[r_item].x : r_item;
	const char *s = seq + off[r];
	const int len = (int)(off[r + 1] - off[r]);
	const int own_beg = items ? items[r_item].y : 0, own_end = items ? items[r_item].z : len;
	const int warm_beg = own_beg - (w + k + 64) > 0 ? own_beg - (w + k + 64) : 0;
	const uint32_t rid = rid_arr ? rid_arr[r] : 0u;
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const uint64_t MAXV = ~0ULL;
	mg128_t *out = mz ? mz + mz_off[r_item] : 0;
	// single pass (cnt AND mz given): mz_off holds CAPACITIES, writes beyond the read's slots are dropped and the caller, who sees
	// cnt[r] > capacity, falls back to count + write
	const int cap = (mz && cnt) ? (int)(mz_off[r_item + 1] - mz_off[r_item]) : 0x7fffffff;
#define SK_PUT(o_, x_, y_) do { if ((o_) < cap) { out[(o_)].x = (x_); out[(o_)].y = (y_); } } while (0)

	// k-1 virtual "nothing yet" codes so that the first real base sits at compact index k-1
	if (lane < k - 1) codes[lane] = 0;
	int nn = k - 1;      // compact (non-ambiguous) bases so far, incl. the virtual ones
	int T = 0;           // events so far
	int lastN = -1;      // event index of the most recent ambiguous base
	int n_out = 0;
	__syncthreads();

	for (int base = warm_beg; base < own_end; base += 64) {
		const int i = base + lane;
		const bool valid = i < own_end;
		const int c = valid ? sk_nt4((unsigned char)s[i]) : 4;
		const bool nonN = valid && c < 4;
		const uint64_t m_non = __ballot(nonN);
		const int j = nn + __popcll(m_non & mga_lanemask_lt());
		if (nonN) codes[j & (SK_CRING - 1)] = (uint8_t)(c | (3 ^ c) << 2);
		__syncthreads();
		uint64_t fwd = 0, rev = 0;
		if (nonN) {
			for (int m = 0; m < k; ++m) { // base j-m sits at bit 2m of fwd and bit 2(k-1-m) of rev
				const uint32_t cc = codes[(j - m) & (SK_CRING - 1)];
				fwd |= (uint64_t)(cc & 3) << 2 * m;
				rev |= (uint64_t)(cc >> 2) << 2 * (k - 1 - m);
			}
		}
		const bool sym = nonN && fwd == rev;
		const bool isN = valid && c >= 4;
		const bool isev = valid && !sym;
		const uint64_t m_ev = __ballot(isev), m_N = __ballot(isN);
		const int t = T + __popcll(m_ev & mga_lanemask_lt());
		int l = 0;
		if (isev && !isN) {
			const uint64_t prevN = m_N & mga_lanemask_lt();
			int ln = lastN;
			if (prevN) { const int hl = 63 - __clzll(prevN); ln = T + __popcll(m_ev & ((1ULL << hl) - 1ULL)); }
			l = t - ln;
		}
		uint64_t x = MAXV, y = MAXV;
		if (isev && !isN && l >= k) {
			const int z = fwd < rev ? 0 : 1;
			x = sk_hash64(z ? rev : fwd, mask) << 8 | (uint64_t)k;
			y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z;
		}
		if (isev) { ex[t & (SK_RING - 1)] = x; ey[t & (SK_RING - 1)] = y; el[t & (SK_RING - 1)] = l; }
		__syncthreads();

		// what the reference pushes while consuming event t
		int P = -1, N = -1, c0 = 0, c1 = 0, c2 = 0; // c0: E0 duplicates, c1: old minimum (0/1), c2: E2 duplicates
		uint64_t px = MAXV, nx = MAXV;
		bool moved_out = false;
		if (isev) {
			const int lo = t - w < 0 ? 0 : t - w;
			for (int q = lo; q <= t; ++q) {
				const uint64_t v = ex[q & (SK_RING - 1)];
				if (q <= t - 1 && v <= px) px = v, P = q;                 // window [t-w, t-1]
				if (q >= t - w + 1 && v <= nx) nx = v, N = q;             // window [t-w+1, t]
			}
			if (l == w + k - 1 && px != MAXV) {                           // E0, sketch.c:84-88
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t - 1; ++q)
					if (ex[q & (SK_RING - 1)] == px && q != P) ++c0;
			}
			if (x <= px) {                                                // E1, sketch.c:89-91
				if (l >= w + k && px != MAXV) c1 = 1;
			} else if (P == t - w) {                                      // E2, sketch.c:92-104
				moved_out = true;
				if (l >= w + k - 1) {
					c1 = 1;
					if (nx != MAXV)
						for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t; ++q)
							if (ex[q & (SK_RING - 1)] == nx && q != N) ++c2;
				}
			}
		}
		if (i < own_beg) c0 = c1 = c2 = 0; // warm-up: the previous piece emits these
		const int tot = c0 + c1 + c2;
		const int incl = mga_wave_incl_scan_i32(tot);
		const int wave_tot = __shfl(incl, 63);
		if (out && tot) {
			int o = n_out + incl - tot;
			if (c0) {
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t - 1; ++q)
					if (ex[q & (SK_RING - 1)] == px && q != P) { SK_PUT(o, px, ey[q & (SK_RING - 1)]); ++o; }
			}
			if (c1) { SK_PUT(o, px, ey[P & (SK_RING - 1)]); ++o; }
			if (c2 && moved_out) {
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t; ++q)
					if (ex[q & (SK_RING - 1)] == nx && q != N) { SK_PUT(o, nx, ey[q & (SK_RING - 1)]); ++o; }
			}
		}
		n_out += wave_tot;
		if (m_N) { const int hl = 63 - __clzll(m_N); lastN = T + __popcll(m_ev & ((1ULL << hl) - 1ULL)); }
		nn += __popcll(m_non);
		T += __popcll(m_ev);
		__syncthreads(); // ring slots are reused by the next step
	}
	// the final minimum (sketch.c:107-108): rightmost minimum of the last w events -- of the sequence, i.e. of its last piece
	if (lane == 0) {
		if (T > 0 && own_end == len) {
			uint64_t nx = MAXV; int N = -1;
			for (int q = (T - w < 0 ? 0 : T - w); q <= T - 1; ++q) {
				const uint64_t v = ex[q & (SK_RING - 1)];
				if (v <= nx) nx = v, N = q;
			}
			if (nx != MAXV) {
				if (out) SK_PUT(n_out, nx, ey[N & (SK_RING - 1)]);
				++n_out;
			}
		}
		if (cnt) cnt[r_item] = n_out;
	}
}

// ---------------- round 6: the same event timeline at ~0.4 x the instructions ----------------
// [measured, rounds 3-5] k_sketch_v1 above issues ~550 vector instructions per step of 64 bases -- k LDS byte reads to rebuild each lane's k-mer, a scan of w + 1 ring entries
// for TWO windows per lane, a shuffle-based prefix sum, a byte load the step waits for -- and a SIMD retires one wavefront instruction per 4 cycles: 22.6 ms per 1.25 Gbp.
// Same state machine, same ring, same outputs; what changed:
//   * k-mers from BIT PLANES: the step's 64 2-bit codes are two ballots (or two words of the packed read, below); the scalar unit interleaves them into the packed stream
//     (s_brev_b64 + s_bitreplicate_b64_b32), a lane takes its 2k-bit window with one 64-bit funnel shift and gets the reverse strand as the reverse complement of the forward
//     one (bit reversal + pair swap).  Holds when the last 64 + k bases had no ambiguous one (else the LDS code ring of v1 serves the step: same values).
//   * ONE window scan per lane: the window [t - w, t - 1] of an event is the window [t' - w + 1, t'] of the event before it -- when every lane of the step is an event (always,
//     for odd k) its minimum comes from the neighbour lane by DPP, lane 0's from the step before.
//   * emission offsets by a DPP scan; the ring's visibility points are wave-level (one wavefront per sequence: no s_barrier); the bases of four steps are loaded at once, the
//     next four while these are processed.
// PACKED INPUT (north_star: "packed 2-bit sequence"): with `planes` the bases come as three 64-bit words per 64 bytes of the read buffer -- low bit, high bit, is-ACGT
// (k_pack2 below) -- 0.375 bytes per base instead of 1, read by SCALAR loads: the decode and the three ballots of a step disappear.
// [measured, round 6, profiles/r06n_sketch_forms.txt, 125 000 x 10 kb reads, isolated] v1 22.76 ms; this kernel on the bytes 15.04 ms; on the planes 16.78 ms INCLUDING the 1.25 GB
// pass of k_pack2 that makes them -- the kernel is bound by the vector instructions of hash + window scan (0.2 bytes of HBM per instruction), not by the 1 byte per base it reads, so the
// packed form buys nothing here and costs its packing pass (on the host it would cost the reader's threads 1-2 ns per base of a budget of 3).  ASCII stays the resident form (the WFA and
// text kernels compare bytes anyway); MGA_SKETCH_2BIT=1 keeps the packed path alive for the tests and for a host with a faster link than compute.
__device__ __forceinline__ uint64_t sk_bitrep(uint32_t x) { uint64_t r; asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(x)); return r; }
__device__ __forceinline__ uint64_t sk_prev64(uint64_t v, uint64_t first) // lane l <- v[l - 1]; lane 0 <- first
{
	const uint32_t lo = (uint32_t)lc_prev_lane((int32_t)(uint32_t)v, (int32_t)(uint32_t)first), hi = (uint32_t)lc_prev_lane((int32_t)(v >> 32), (int32_t)(first >> 32));
	return (uint64_t)hi << 32 | lo;
}

template<int RING> // event ring: a lane looks back w events, a step adds 64: 128 entries for w <= 63 (2 KB of LDS: 8 waves per SIMD), 512 up to w = 255
__global__ void __launch_bounds__(64) k_sketch(int n, const char *__restrict__ seq, const int64_t *__restrict__ off,
											   const uint32_t *__restrict__ rid_arr, int w, int k,
											   int32_t *__restrict__ cnt, const int64_t *__restrict__ mz_off, mg128_t *__restrict__ mz,
											   const int4 *__restrict__ items, const uint64_t *__restrict__ planes)
{
	__shared__ uint64_t ex[RING], ey[RING];
	__shared__ uint8_t codes[SK_CRING];
	const int r_item = blockIdx.x;
	if (r_item >= n) return;
	const int lane = threadIdx.x;
	const int r = items ? items[r_item].x : r_item;
	const int64_t s_off = off[r];
	const char *s = seq + s_off;
	const int len = (int)(off[r + 1] - s_off);
	const int own_beg = items ? items[r_item].y : 0, own_end = items ? items[r_item].z : len;
	const int warm_beg = own_beg - (w + k + 64) > 0 ? own_beg - (w + k + 64) : 0;
	const uint32_t rid = rid_arr ? rid_arr[r] : 0u;
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const uint64_t MAXV = ~0ULL;
	mg128_t *out = mz ? mz + mz_off[r_item] : 0;
	const int cap = (mz && cnt) ? (int)(mz_off[r_item + 1] - mz_off[r_item]) : 0x7fffffff;

	if (lane < k - 1) codes[lane] = 0;
	int nn = k - 1, T = 0, lastN = -1, n_out = 0;
	uint64_t f_prev_lo = 0;            // the packed forward stream of the step before (its older half is never needed: k <= 28 < 32)
	int clean = 0;                     // consecutive real bases behind the current step (saturating): the plane form needs k - 1 of them
	uint64_t c_nx = MAXV; int c_N = -1; // rightmost minimum of the window [T - w, T - 1] = what the NEXT event's "previous window" is
	mga_wave_sync();

	auto step = [&](const int base, const int ch_raw, const uint64_t q0, const uint64_t q1, const uint64_t qn) __attribute__((always_inline)) {
		const int i = base + lane;
		const int n_val = own_end - base < 64 ? own_end - base : 64;
		const uint64_t m_valid = n_val == 64 ? ~0ULL : (1ULL << n_val) - 1ULL;
		const bool valid = lane < n_val;
		int c;
		uint64_t m_non, p0, p1;
		if (planes) { // (uniform)
			m_non = qn & m_valid, p0 = q0, p1 = q1;
			c = (m_non >> lane & 1) ? (int)((p0 >> lane & 1) | (p1 >> lane & 1) << 1) : 4;
		} else {
			c = valid ? sk_nt4((unsigned char)ch_raw) : 4;
			m_non = __ballot(c < 4);
			p0 = __ballot((c & 1) != 0 && c < 4), p1 = __ballot((c & 2) != 0);
		}
		const bool nonN = valid && c < 4;
		const int j = nn + __popcll(m_non & mga_lanemask_lt());
		if (nonN) codes[j & (SK_CRING - 1)] = (uint8_t)(c | (3 ^ c) << 2);
		uint64_t fwd = 0, rev = 0;
		// the packed forward stream of this step: lane l's code at bits 2 (63 - l) (+ 1): the NEWEST base lowest, as in kmer[0] (sketch.c:72)
		// (scalar work; an ambiguous base's bits are whatever: a lane only reads the k - 1 codes before it, and `clean` says those are real)
		const bool all_real = m_non == m_valid; // (uniform) no ambiguous base among the step's bases
		const uint64_t r0 = __builtin_bitreverse64(p0), r1 = __builtin_bitreverse64(p1);
		const uint64_t f_lo = (sk_bitrep((uint32_t)r0) & 0x5555555555555555ULL) | (sk_bitrep((uint32_t)r1) & 0xaaaaaaaaaaaaaaaaULL);               // lanes 63 .. 32
		const uint64_t f_hi = (sk_bitrep((uint32_t)(r0 >> 32)) & 0x5555555555555555ULL) | (sk_bitrep((uint32_t)(r1 >> 32)) & 0xaaaaaaaaaaaaaaaaULL); // lanes 31 .. 0
		if (all_real && clean >= k - 1) { // every lane's k bases are real ones: windows of the packed stream
			const uint64_t A = lane >= 32 ? f_lo : f_hi, B = lane >= 32 ? f_hi : f_prev_lo;
			const int sh = (126 - 2 * lane) & 63;
			fwd = ((A >> sh) | ((B << 1) << (63 - sh))) & mask;
			uint64_t y = __builtin_bitreverse64(~fwd & mask);                                       // groups reversed, and the two bits of each group
			y = (y >> 1 & 0x5555555555555555ULL) | (y & 0x5555555555555555ULL) << 1;                 // ... put back in order
			rev = y >> (64 - 2 * k);
			if (!valid) fwd = 0, rev = 0;
		} else { // the code ring (k_sketch_v1): virtual codes at a sequence's start, ambiguous bases skipped
			mga_wave_sync();
			if (nonN) {
				for (int m = 0; m < k; ++m) { // base j-m sits at bit 2m of fwd and bit 2(k-1-m) of rev
					const uint32_t cc = codes[(j - m) & (SK_CRING - 1)];
					fwd |= (uint64_t)(cc & 3) << 2 * m;
					rev |= (uint64_t)(cc >> 2) << 2 * (k - 1 - m);
				}
			}
		}
		const bool sym = nonN && fwd == rev;
		const bool isN = valid && c >= 4;
		const bool isev = valid && !sym;
		const uint64_t m_ev = __ballot(isev), m_N = m_valid & ~m_non;
		const int t = T + __popcll(m_ev & mga_lanemask_lt());
		int l = 0;
		if (isev && !isN) {
			const uint64_t prevN = m_N & mga_lanemask_lt();
			int ln = lastN;
			if (prevN) { const int hl = 63 - __clzll(prevN); ln = T + __popcll(m_ev & ((1ULL << hl) - 1ULL)); }
			l = t - ln;
		}
		uint64_t x = MAXV, y = MAXV;
		if (isev && !isN && l >= k) {
			const int z = fwd < rev ? 0 : 1;
			x = sk_hash64(z ? rev : fwd, mask) << 8 | (uint64_t)k;
			y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z;
		}
		if (isev) { ex[t & (RING - 1)] = x; ey[t & (RING - 1)] = y; }
		mga_wave_sync();

		// what the reference pushes while consuming event t
		int P = -1, N = -1, c0 = 0, c1 = 0, c2 = 0; // c0: E0 duplicates, c1: old minimum (0/1), c2: E2 duplicates
		uint64_t px = MAXV, nx = MAXV;
		bool moved_out = false;
		const bool dense = m_ev == m_valid; // (uniform) lane l <-> event T + l
		if (dense) {
			// window [t-w+1, t]: w entries for every lane (one trip count for the wavefront: no per-lane loop bounds); only at a sequence's start do some not exist yet
			const int q0 = t - w + 1;
			if (T >= w - 1) { for (int it = 0; it < w; ++it) { const int q = q0 + it; const uint64_t v = ex[q & (RING - 1)]; const bool ok = v <= nx; nx = ok ? v : nx, N = ok ? q : N; } }
			else { for (int it = 0; it < w; ++it) { const int q = q0 + it; const uint64_t v = ex[q & (RING - 1)]; const bool ok = q >= 0 && v <= nx; nx = ok ? v : nx, N = ok ? q : N; } }
			if (!isev) nx = MAXV, N = -1; // (lanes beyond the sequence's end)
			px = sk_prev64(nx, c_nx), P = lc_prev_lane(N, c_N); // window [t-w, t-1] = the event before's [t'-w+1, t']
		} else if (isev) {
			const int lo = t - w < 0 ? 0 : t - w;
			for (int q = lo; q <= t; ++q) {
				const uint64_t v = ex[q & (RING - 1)];
				if (q <= t - 1 && v <= px) px = v, P = q;                 // window [t-w, t-1]
				if (q >= t - w + 1 && v <= nx) nx = v, N = q;             // window [t-w+1, t]
			}
		}
		if (isev) {
			if (l == w + k - 1 && px != MAXV) {                           // E0, sketch.c:84-88
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t - 1; ++q)
					if (ex[q & (RING - 1)] == px && q != P) ++c0;
			}
			if (x <= px) {                                                // E1, sketch.c:89-91
				if (l >= w + k && px != MAXV) c1 = 1;
			} else if (P == t - w) {                                      // E2, sketch.c:92-104
				moved_out = true;
				if (l >= w + k - 1) {
					c1 = 1;
					if (nx != MAXV)
						for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t; ++q)
							if (ex[q & (RING - 1)] == nx && q != N) ++c2;
				}
			}
		}
		if (i < own_beg) c0 = c1 = c2 = 0; // warm-up: the previous piece emits these
		const int tot = c0 + c1 + c2;
		const int incl = lc_scan_add(tot, 0);
		const int wave_tot = __builtin_amdgcn_readlane(incl, 63);
		if (out && tot) {
			int o = n_out + incl - tot;
			if (c0) {
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t - 1; ++q)
					if (ex[q & (RING - 1)] == px && q != P) { SK_PUT(o, px, ey[q & (RING - 1)]); ++o; }
			}
			if (c1) { SK_PUT(o, px, ey[P & (RING - 1)]); ++o; }
			if (c2 && moved_out) {
				for (int q = (t - w + 1 < 0 ? 0 : t - w + 1); q <= t; ++q)
					if (ex[q & (RING - 1)] == nx && q != N) { SK_PUT(o, nx, ey[q & (RING - 1)]); ++o; }
			}
		}
		n_out += wave_tot;
		if (m_ev) { // the last event's own window is the next event's previous one
			const int le = 63 - (int)__clzll(m_ev);
			c_nx = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(nx >> 32), le) << 32 | (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)nx, le);
			c_N = __builtin_amdgcn_readlane(N, le);
		}
		if (m_N) { const int hl = 63 - (int)__clzll(m_N); lastN = T + __popcll(m_ev & ((1ULL << hl) - 1ULL)); clean = __popcll(m_valid) - 1 - hl; }
		else clean = clean + n_val > 64 ? 64 : clean + n_val;
		f_prev_lo = f_lo;
		nn += __popcll(m_non);
		T += __popcll(m_ev);
		mga_wave_sync(); // ring slots are reused by the next step
	};

	if (planes) {
		for (int base = warm_beg; base < own_end; base += 64) {
			const int64_t a0 = s_off + base;
			const uint64_t *pw = planes + 3 * (a0 >> 6);
			const int sft = (int)(a0 & 63);
			uint64_t q0 = pw[0] >> sft, q1 = pw[1] >> sft, qn = pw[2] >> sft;
			if (sft) q0 |= pw[3] << (64 - sft), q1 |= pw[4] << (64 - sft), qn |= pw[5] << (64 - sft);
			step(base, 0, q0, q1, qn);
		}
	} else {
		// bases of four steps per trip, the next four in flight while these are processed
		int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#define SK_LOAD4(base_) do { const int p_ = (base_) + lane; b0 = p_ < own_end ? (unsigned char)s[p_] : 0, b1 = p_ + 64 < own_end ? (unsigned char)s[p_ + 64] : 0, \
		b2 = p_ + 128 < own_end ? (unsigned char)s[p_ + 128] : 0, b3 = p_ + 192 < own_end ? (unsigned char)s[p_ + 192] : 0; } while (0)
		SK_LOAD4(warm_beg);
		for (int base = warm_beg; base < own_end; base += 256) {
			const int a0 = b0, a1 = b1, a2 = b2, a3 = b3;
			if (base + 256 < own_end) SK_LOAD4(base + 256);
			step(base, a0, 0, 0, 0);
			if (base + 64 < own_end) step(base + 64, a1, 0, 0, 0);
			if (base + 128 < own_end) step(base + 128, a2, 0, 0, 0);
			if (base + 192 < own_end) step(base + 192, a3, 0, 0, 0);
		}
#undef SK_LOAD4
	}
	// the final minimum (sketch.c:107-108): rightmost minimum of the last w events -- of the sequence, i.e. of its last piece
	if (lane == 0) {
		if (T > 0 && own_end == len) {
			uint64_t nx = MAXV; int N = -1;
			for (int q = (T - w < 0 ? 0 : T - w); q <= T - 1; ++q) {
				const uint64_t v = ex[q & (RING - 1)];
				if (v <= nx) nx = v, N = q;
			}
			if (nx != MAXV) {
				if (out) SK_PUT(n_out, nx, ey[N & (RING - 1)]);
				++n_out;
			}
		}
		if (cnt) cnt[r_item] = n_out;
	}
}

// ASCII -> bit planes: words 3 b .. 3 b + 2 describe bytes 64 b .. 64 b + 63 of the buffer: low bit of the 2-bit code, high bit, is-ACGT (seq_nt4_table, sketch.c:9-26)
__global__ void __launch_bounds__(256) k_pack2(int64_t n_bytes, const char *__restrict__ seq, uint64_t *__restrict__ planes)
{
	const int lane = threadIdx.x & 63;
	const int64_t n_blk = (n_bytes + 63) >> 6;
	for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_blk + 2; b += (int64_t)gridDim.x * 4) { // (+ 2 words of zeros behind the end: a step reads one block ahead)
		const int64_t p = 64 * b + lane;
		const int c = p < n_bytes ? sk_nt4((unsigned char)seq[p]) : 4;
		const uint64_t q0 = __ballot((c & 1) != 0 && c < 4), q1 = __ballot((c & 2) != 0), qn = __ballot(c < 4);
		if (lane == 0) planes[3 * b] = q0, planes[3 * b + 1] = q1, planes[3 * b + 2] = qn;
	}
}

// MGA_SKETCH_V1=1: the kernel of rounds 1-5 (A/B, tests).  The packed form of a read buffer is made by mga_dev_pack2() and used by the sketch launches on the same buffer that follow.
static int sk_use_v1(void) { const char *e = getenv("MGA_SKETCH_V1"); return e && atoi(e) > 0; }
static const uint64_t *sk_planes_of(const mga_sctx_t *sc, const char *d_seq) { return sc->sk_planes_src == (const void*)d_seq && d_seq ? (const uint64_t*)sc->sk_planes.p : (const uint64_t*)0; }
extern "C" int mga_dev_pack2(mga_sctx_t *sc, const char *d_seq, int64_t n_bytes)
{
	sc->sk_planes_src = 0;
	if (n_bytes <= 0) return 0;
	const int64_t n_blk = ((n_bytes + 63) >> 6) + 2;
	if (mga_dbuf_reserve(&sc->sk_planes, (size_t)n_blk * 24 + 64) < 0) return -1;
	const int64_t wg = (n_blk + 3) / 4;
	mga_prof_begin(sc->stream, MGA_K_SKETCH);
	hipLaunchKernelGGL(k_pack2, dim3((unsigned)(wg < 16384 ? wg : 16384)), dim3(256), 0, (hipStream_t)sc->stream, n_bytes, d_seq, (uint64_t*)sc->sk_planes.p);
	mga_prof_end(sc->stream, MGA_K_SKETCH);
	MGA_HIP_CHECK(hipGetLastError());
	sc->sk_planes_src = d_seq;
	return 0;
}

extern "C" int mga_dev_sketch_items(mga_sctx_t *sc, int n_items, const int32_t *d_items, const char *d_seq, const int64_t *d_off, const uint32_t *d_rid, int w, int k,
									int32_t *d_cnt, const int64_t *d_mz_off, mg128_t *d_mz)
{
	if (n_items <= 0) return 0;
	if (w < 1 || w > 255 || k < 1 || k > 28 || !(k & 1)) { mga_set_error("sketch pieces: need 0<w<256 and an odd 0<k<=28, got w=%d k=%d", w, k); return -1; }
	mga_prof_begin(sc->stream, MGA_K_SKETCH);
	if (sk_use_v1()) hipLaunchKernelGGL(k_sketch_v1, dim3(n_items), dim3(64), 0, (hipStream_t)sc->stream, n_items, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)d_items);
	else if (w <= 63) hipLaunchKernelGGL(k_sketch<128>, dim3(n_items), dim3(64), 0, (hipStream_t)sc->stream, n_items, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)d_items, sk_planes_of(sc, d_seq));
	else hipLaunchKernelGGL(k_sketch<SK_RING>, dim3(n_items), dim3(64), 0, (hipStream_t)sc->stream, n_items, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)d_items, sk_planes_of(sc, d_seq));
	mga_prof_end(sc->stream, MGA_K_SKETCH);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" int mga_dev_sketch(mga_sctx_t *sc, int n, const char *d_seq, const int64_t *d_off, const uint32_t *d_rid, int w, int k,
							  int32_t *d_cnt, const int64_t *d_mz_off, mg128_t *d_mz)
{
	if (n <= 0) return 0;
	if (w < 1 || w > 255 || k < 1 || k > 28) { mga_set_error("sketch: need 0<w<256 and 0<k<=28 (sketch.c:62), got w=%d k=%d", w, k); return -1; }
	mga_prof_begin(sc->stream, MGA_K_SKETCH);
	if (sk_use_v1()) hipLaunchKernelGGL(k_sketch_v1, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)0);
	else if (w <= 63) hipLaunchKernelGGL(k_sketch<128>, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)0, sk_planes_of(sc, d_seq));
	else hipLaunchKernelGGL(k_sketch<SK_RING>, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)0, sk_planes_of(sc, d_seq));
	mga_prof_end(sc->stream, MGA_K_SKETCH);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
