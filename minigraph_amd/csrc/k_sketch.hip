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

__global__ void __launch_bounds__(64) k_sketch(int n, const char *__restrict__ seq, const int64_t *__restrict__ off,
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

extern "C" int mga_dev_sketch_items(mga_sctx_t *sc, int n_items, const int32_t *d_items, const char *d_seq, const int64_t *d_off, const uint32_t *d_rid, int w, int k,
									int32_t *d_cnt, const int64_t *d_mz_off, mg128_t *d_mz)
{
	if (n_items <= 0) return 0;
	if (w < 1 || w > 255 || k < 1 || k > 28 || !(k & 1)) { mga_set_error("sketch pieces: need 0<w<256 and an odd 0<k<=28, got w=%d k=%d", w, k); return -1; }
	mga_prof_begin(sc->stream, MGA_K_SKETCH);
	hipLaunchKernelGGL(k_sketch, dim3(n_items), dim3(64), 0, (hipStream_t)sc->stream, n_items, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)d_items);
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
	hipLaunchKernelGGL(k_sketch, dim3(n), dim3(64), 0, (hipStream_t)sc->stream, n, d_seq, d_off, d_rid, w, k, d_cnt, d_mz_off, d_mz, (const int4*)0);
	mga_prof_end(sc->stream, MGA_K_SKETCH);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}
