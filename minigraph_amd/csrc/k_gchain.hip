// k_gchain.hip -- graph chaining on the device: one wavefront per read runs gc_map_read() (gc_core.h) on its lane 0, from the
// linear chains k_lchain left in HBM to the filtered graph chains (chain records, clean-up, DP with shortest-walk reachability,
// GWFA / shortest-walk bridging, ordering, primary/secondary, filters).  Reference: map-algo.c:422-474, gchain1.c, shortk.c,
// gfa-ed.c, gcmisc.c.
//
// Execution model: the routine runs REPLICATED on the 64 lanes of the read's wavefront -- the control flow is a chain of dependent,
// data-driven decisions over a handful of records (1.9 linear chains, <1 graph search, <1 GWFA per 10 kb read), identical on every
// lane -- and its per-element loops (GWFA extension and wavefront construction, dedup, anchor copies, minimizer ranks) are split over
// the lanes, 64 loads in flight instead of one dependent chain.  Persistent wavefronts pull reads from an atomic counter, each with a
// private scratch ARENA in HBM (1 MiB; a read that outgrows it is re-run by a second launch with 256 MiB arenas, beyond that the job
// stops with an error).  [measured] see DESIGN.md 4.
//
// The same source runs on the host (mga_gchain_host_read below: -x asm where the chainer is host code, and the CPU parity tests).
#include <stdio.h>
#include <unistd.h>
#include <math.h>
#include "mga_dev.h"
#include "dev_common.h"
#define GC_PARSORT 5   /* [measured] gathering the sorted subset by lanes (bit 2 / 8) faults on the device; the split and the merge by rank are fine */
#include "gc_core.h"

static_assert(sizeof(gc_arc_t) == sizeof(gfa_arc_t) && sizeof(gc_arc_t) == 32, "gc_arc_t must mirror gfa_arc_t");

// ---- reverse complements of all segments, same offsets as the forward copy ----
__constant__ unsigned char c_gc_comp[256];

__global__ void __launch_bounds__(256) k_revcomp(int n_seg, const char *__restrict__ fw, const int64_t *__restrict__ off, char *__restrict__ rc)
{
	for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
		const int64_t b = off[s], len = off[s + 1] - b;
		for (int64_t i = threadIdx.x; i < len; i += blockDim.x) rc[b + len - 1 - i] = (char)c_gc_comp[(unsigned char)fw[b + i]];
	}
}

extern "C" int mga_dev_graph_upload(mga_sctx_t *sc, const gfa_t *g, const unsigned char *comp, mga_didx_t *ix)
{
	const uint32_t n_vtx = gfa_n_vtx(g);
	const int64_t tot = 0;
	(void)tot;
	MGA_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_gc_comp), comp, 256));
	ix->d_arc = mga_dmalloc((size_t)(g->n_arc + 1) * sizeof(gfa_arc_t));
	ix->d_arc_idx = (uint64_t*)mga_dmalloc((size_t)(n_vtx + 1) * 8);
	if (ix->d_arc == 0 || ix->d_arc_idx == 0) return -1;
	if (mga_h2d(ix->d_arc, g->arc, (size_t)g->n_arc * sizeof(gfa_arc_t)) < 0 || mga_h2d(ix->d_arc_idx, g->idx, (size_t)n_vtx * 8) < 0) return -1;
	{
		int64_t total = 0;
		if (mga_d2h(&total, ix->d_gseq_off + ix->n_seg, 8) < 0) return -1;
		ix->d_gseq_rc = (char*)mga_dmalloc((size_t)total + 64);
		if (ix->d_gseq_rc == 0) return -1;
		MGA_HIP_CHECK(hipMemsetAsync(ix->d_gseq_rc + total, 0, 64, (hipStream_t)sc->stream));
		const int nb = ix->n_seg < 65536 ? (ix->n_seg > 0 ? ix->n_seg : 1) : 65536;
		hipLaunchKernelGGL(k_revcomp, dim3(nb), dim3(256), 0, (hipStream_t)sc->stream, ix->n_seg, (const char*)ix->d_gseq, (const int64_t*)ix->d_gseq_off, ix->d_gseq_rc);
		MGA_HIP_CHECK(hipGetLastError());
		if (mga_ssync(sc) < 0) return -1;
	}
	if (mga_dev_gaf_names_upload(g, ix) < 0) return -1; // (GAF lines on the device, k_gaf.hip)
	return 0;
}

// ---- the kernel ----
struct gck_in_t {
	int n;
	const int32_t *list;             // reads to do (NULL: 0..n)
	const int64_t *a_off;            // anchors / chains of read i at a_off[i]
	const int32_t *nu, *nb;
	const uint64_t *u;
	const mg128_t *b;
	const int64_t *mini_off; const int32_t *mini;
	const int64_t *q_off; const char *seq;
	const uint32_t *hash;
	const int32_t *rflag;            // k_lchain's verdict on the long-join rescue: 2 = this read's chains come from the host tree, not from here
};
struct gck_out_t {
	mga_gc_hdr_t *hdr;
	gc_rec_t *gc_pool; int64_t gc_cap;
	mg_llchain_t *lc_pool; int64_t lc_cap;
	mg128_t *a_pool; int64_t a_cap;
	unsigned long long *ctl;         // [0] next read, [1] gc pool used, [2] lc pool used, [3] reads to retry, [4] gwfa calls, [5] shortest-walk calls, [6] anchor pool used, [7] peak arena
	int32_t *retry;
};

#define GCK_FAST_BYTES 16384 // LDS scratch per wavefront: two waves per SIMD = eight per CU = 128 KB of the CU's 160

__device__ __forceinline__ void gck_copy_words(void *dst, const void *src, int64_t bytes, int lane) // both 4-byte aligned; 16-byte records at 16-byte addresses move as such
{
	if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes) & 15) == 0) {
		uint4 *d = (uint4*)dst;
		const uint4 *s = (const uint4*)src;
		for (int64_t i = lane, n = bytes >> 4; i < n; i += 64) d[i] = s[i];
		return;
	}
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	for (int64_t i = lane, n = bytes >> 2; i < n; i += 64) d[i] = s[i];
}

// a read's records leave its arena for the chunk's pools: one lane reserves, all 64 copy; a read that did not make it goes to the retry list
__device__ __forceinline__ void gck_publish(const gck_out_t &out, int r, int lane, int32_t status, const gc_result_t *R, const mg128_t *res_a, long long peak)
{
	int32_t n_gc = 0, n_lc = 0, n_a = 0;
	long long gc_off = 0, lc_off = 0, a_off = 0;
	mga_gc_hdr_t *H = &out.hdr[r];
	if (status == GC_OK) n_gc = R->n_gc, n_lc = R->n_lc, n_a = R->n_a;
	mga_wave_sync();
	if (lane == 0) { // one lane talks to the chunk's pools and counters
		if (status == GC_OK) {
			gc_off = (long long)atomicAdd(&out.ctl[1], (unsigned long long)n_gc);
			lc_off = (long long)atomicAdd(&out.ctl[2], (unsigned long long)n_lc);
			a_off = (long long)atomicAdd(&out.ctl[6], (unsigned long long)n_a);
			if (gc_off + n_gc > out.gc_cap || lc_off + n_lc > out.lc_cap || a_off + n_a > out.a_cap) status = MGA_GC_E_POOL;
			atomicAdd(&out.ctl[4], (unsigned long long)R->n_gwfa);
			atomicAdd(&out.ctl[8], (unsigned long long)R->n_fast);
			atomicAdd(&out.ctl[5], (unsigned long long)R->n_shortk);
			atomicMax(&out.ctl[7], (unsigned long long)peak);
		}
		if (status != GC_OK) out.retry[atomicAdd(&out.ctl[3], 1ULL)] = r, n_gc = n_lc = n_a = 0;
		H->n_gc = n_gc, H->n_lc = n_lc, H->n_a = n_a, H->status = status, H->gc_off = gc_off, H->lc_off = lc_off, H->a_off = a_off;
	}
	// the records leave the arena on all 64 lanes
	status = __shfl(status, 0), n_gc = __shfl(n_gc, 0), n_lc = __shfl(n_lc, 0), n_a = __shfl(n_a, 0);
	gc_off = __shfl(gc_off, 0), lc_off = __shfl(lc_off, 0), a_off = __shfl(a_off, 0);
	if (status == GC_OK) {
		gck_copy_words(out.gc_pool + gc_off, R->gc, (int64_t)n_gc * (int64_t)sizeof(gc_rec_t), lane);
		gck_copy_words(out.lc_pool + lc_off, R->lc, (int64_t)n_lc * (int64_t)sizeof(mg_llchain_t), lane);
		gck_copy_words(out.a_pool + a_off, res_a, (int64_t)n_a * 16, lane);
	}
}

#ifndef GC_AB_WAVES_PER_EU
#define GC_AB_WAVES_PER_EU 2
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GC_AB_WAVES_PER_EU, 8))) k_gchain(gck_in_t in, gck_out_t out, gc_graph_t G_arg, gc_par_t P_arg, char *arena_mem, int64_t arena_bytes, int fast_bytes)
{
	extern __shared__ __attribute__((aligned(16))) char fast_lds[]; // fast_bytes of dynamic LDS (0: none)
	const int lane = threadIdx.x;
	char *my_arena = arena_mem + (int64_t)blockIdx.x * arena_bytes;
#ifndef GC_AB_NO_LDS_STATE
	// the routine's top-level state -- arena header, graph view, parameters, read, result -- once per wavefront in LDS (GC_STATE, gc_core.h): handed on by address, a
	// private copy per lane would live in scratch memory
	__shared__ gc_graph_t G_lds;
	__shared__ gc_par_t P_lds;
	__shared__ gc_arena_t A_lds;
	__shared__ gc_read_t rd_lds;
	__shared__ gc_result_t R_lds;
	G_lds = G_arg, P_lds = P_arg;
	mga_wave_sync();
	const gc_graph_t &G = G_lds;
	const gc_par_t &P = P_lds;
#else
	const gc_graph_t &G = G_arg;
	const gc_par_t &P = P_arg;
#endif
	const long long t_wave0 = out.ctl[15] ? (long long)clock64() : 0;
	for (;;) {
		int slot = 0;
		const long long t_read0 = out.ctl[15] ? (long long)clock64() : 0;
		if (lane == 0) slot = (int)atomicAdd(&out.ctl[0], 1ULL);
		slot = __shfl(slot, 0);
		if (slot >= in.n) break;
		const int r = in.list ? in.list[slot] : slot;
		const int64_t off = in.a_off[r];
		const int32_t n_u = in.nu[r], n_b = in.nb[r];
		mga_gc_hdr_t *H = &out.hdr[r];
		if (in.rflag && in.rflag[r] == 2) { if (lane == 0) { H->n_gc = H->n_lc = H->n_a = 0, H->status = MGA_GC_HOST, H->gc_off = H->lc_off = H->a_off = 0; } continue; }
		if (n_u <= 0 || n_b <= 0) { if (lane == 0) { H->n_gc = H->n_lc = H->n_a = 0, H->status = 0, H->gc_off = H->lc_off = H->a_off = 0; } continue; }
#ifndef GC_AB_NO_LDS_STATE
		gc_arena_t &A = A_lds;
		gc_read_t &rd = rd_lds;
		gc_result_t &R = R_lds;
#else
		gc_arena_t A;
		gc_read_t rd;
		gc_result_t R;
#endif
		gc_arena_init(&A, my_arena, arena_bytes, 0);
		// MGA_GC_LDS=1: the scratch of ONE graph search / GWFA call at a time in LDS; a call that outgrows the block is run again in the main arena (see the launcher)
		if (fast_bytes > 0) A.fast_base = (char*)fast_lds, A.fast_cap = fast_bytes;
		if (out.ctl[15]) A.ticks = out.ctl + 16, A.tick_last = (long long)clock64(); // profiling: ctl[15] != 0 asks for per-stage cycle sums in ctl[16..31]
		mg128_t *work = (mg128_t*)gc_alloc(&A, (int64_t)n_b * 16); // the chains' anchors: flags and minimizer ranks are written into this copy
		mg128_t *res_a = (mg128_t*)gc_alloc(&A, (int64_t)n_b * 16); // anchors of the graph chains
		if (work && res_a) gck_copy_words(work, in.b + off, (int64_t)n_b * 16, lane);
		mga_wave_sync();
		int32_t status = GC_E_ARENA;
		R.gc = 0, R.lc = 0;
		if (work && res_a) { // every lane runs the routine on the same values (replicated execution, gc_core.h); its hot loops are split over the lanes
			rd.qlen = (int32_t)(in.q_off[r + 1] - in.q_off[r]), rd.hash = in.hash[r];
			rd.n_u = n_u, rd.u = in.u + off, rd.a = work;
			rd.n_mini = (int32_t)(in.mini_off[r + 1] - in.mini_off[r]), rd.mini_pos = in.mini + in.mini_off[r];
			rd.qseq = in.seq + in.q_off[r];
			R.a = res_a;
			status = gc_map_read(&A, &G, &P, &rd, &R);
			if (status == GC_E_BUG) status = GC_OK, R.n_gc = R.n_lc = R.n_a = 0; // the reference's own bail-outs: the read gets no chains
		}
		gck_publish(out, r, lane, status, &R, res_a, (long long)A.peak);
		mga_wave_sync();
		if (out.ctl[15] && lane == 0) atomicMax(&out.ctl[16 + 15], (unsigned long long)((long long)clock64() - t_read0)); // profiling: the longest read ...
	}
	if (out.ctl[15] && lane == 0) atomicMax(&out.ctl[16 + 7], (unsigned long long)((long long)clock64() - t_wave0)); // ... and the longest wavefront of the launch
}

// ---- the three-kernel form: a read's bridges on wavefronts of their own -------------------------------------------------------------------------
// k_gchain lasts as long as its longest read, and what makes one read take fifty times another are its bridges: independent GWFA calls / graph searches
// (gc_job_run) that depend on two chain records, the graph and the query only.  So a chunk goes through three launches on its stream:
//   k_gchain_p1  a wavefront per read: chain records, clean-up, minimizer ranks, DP + reachability, the assembly's first half (gc_read_p1).  A read that ends up
//                with graph chains leaves its state -- anchors with flags and ranks, chain records, the DP's grouping, the graph-chain records with score and
//                hash -- as ONE block in the chunk's state pool, written by a copy at the END of its turn, and lists its bridges in the chunk's job array;
//   k_gchain_p2  a wavefront per bridge, longest query gaps first: gc_job_run in a scratch arena, the walk's inner vertices into the chunk's vertex pool;
//   k_gchain_p3  a wavefront per read of the state pool: the assembly's second half consuming the bridges in order, measuring, ordering, parents, filters
//                (gc_read_p3), records published to the chunk's pools exactly as k_gchain does.
// Nothing is read across a launch boundary that was not written by a block copy at the end of the previous launch's turn (the pattern k_gchain -> k_plan already
// relies on); every wavefront works in its own scratch arena, as in the one-kernel form.  A read that runs out of arena / pool room in any part goes to the retry
// list and is redone by k_gchain in a large arena.
struct gcs_hdr_t { // a read's state between the parts: 64 bytes, followed by a[n_b] | c[n_c] | u2[n_u2] | kept[n_u2] | gc[n_gc], each padded to 16 bytes
	int32_t read, n_b, n_c, n_u2, n_gc, n_jobs, n_shortk, span;
	int64_t job0, bytes;
	int64_t pad_[2];
};
static_assert(sizeof(gcs_hdr_t) == 64, "gcs_hdr_t");
struct gck_split_t {
	char *state; int64_t state_cap;            // state pool (bytes)
	int64_t *p3list;                           // offsets of the states, in the order part 1 finished them
	gc_job_t *jobs; int64_t jobs_cap;
	int32_t *mid; int64_t mid_cap;             // inner vertices of the bridges' walks
	unsigned long long *sctl;                  // [0] state bytes used, [1] jobs listed, [2] vertices used, [3] states listed, [4] next state of part 3, [8..15] next job of part 2 per length class
};
#define GCS_A16(x) (((int64_t)(x) + 15) & ~(int64_t)15)
#define GCS_N_CLASS 8
struct gcs_view_t { gcs_hdr_t *h; mg128_t *a; gc_chain_t *c; uint64_t *u2; int32_t *kept; gc_rec_t *gc; int64_t bytes; };
// where the arrays of a block with these counts lie (one definition for the writer, the readers and the host's emulation of the hand-over)
__host__ __device__ inline void gcs_view(char *blk, int32_t n_b, int32_t n_c, int32_t n_u2, int32_t n_gc, gcs_view_t *v)
{
	const int64_t o_a = (int64_t)sizeof(gcs_hdr_t), o_c = o_a + (int64_t)n_b * 16, o_u = o_c + GCS_A16((int64_t)n_c * (int64_t)sizeof(gc_chain_t)), o_k = o_u + GCS_A16((int64_t)n_u2 * 8),
				  o_g = o_k + GCS_A16((int64_t)n_u2 * 4);
	v->h = (gcs_hdr_t*)blk, v->a = (mg128_t*)(blk + o_a), v->c = (gc_chain_t*)(blk + o_c), v->u2 = (uint64_t*)(blk + o_u), v->kept = (int32_t*)(blk + o_k), v->gc = (gc_rec_t*)(blk + o_g);
	v->bytes = o_g + GCS_A16((int64_t)n_gc * (int64_t)sizeof(gc_rec_t));
}
__host__ __device__ inline void gcs_copy_words(void *dst, const void *src, int64_t bytes, int lane, int n_lane) // both 4-byte aligned
{
	if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes) & 15) == 0) {
		typedef struct __attribute__((aligned(16))) { uint64_t a, b; } w16_t;
		w16_t *d16 = (w16_t*)dst;
		const w16_t *s16 = (const w16_t*)src;
		for (int64_t i = lane, n = bytes >> 4; i < n; i += n_lane) d16[i] = s16[i];
		return;
	}
	uint32_t *d = (uint32_t*)dst;
	const uint32_t *s = (const uint32_t*)src;
	for (int64_t i = lane, n = bytes >> 2; i < n; i += n_lane) d[i] = s[i];
}
// part 1's state into the block (arrays by all lanes; the header by the caller's lane 0 once the arrays are visible)
__host__ __device__ inline void gcs_write_arrays(const gcs_view_t *v, int32_t n_b, const mg128_t *work, const gc_split_t *sp, const gc_result_t *R, int lane, int n_lane)
{
	gcs_copy_words(v->a, work, (int64_t)n_b * 16, lane, n_lane);
	gcs_copy_words(v->c, sp->c, (int64_t)sp->n_c * (int64_t)sizeof(gc_chain_t), lane, n_lane);
	gcs_copy_words(v->u2, sp->u2, (int64_t)sp->n_u2 * 8, lane, n_lane);
	gcs_copy_words(v->kept, sp->kept, (int64_t)sp->n_u2 * 4, lane, n_lane);
	gcs_copy_words(v->gc, R->gc, (int64_t)R->n_gc * (int64_t)sizeof(gc_rec_t), lane, n_lane);
}
__host__ __device__ inline void gcs_write_header(const gcs_view_t *v, int32_t read, int32_t n_b, const gc_split_t *sp, const gc_result_t *R, int64_t job0)
{
	gcs_hdr_t *h = v->h;
	h->read = read, h->n_b = n_b, h->n_c = sp->n_c, h->n_u2 = sp->n_u2, h->n_gc = R->n_gc, h->n_jobs = sp->n_jobs, h->n_shortk = R->n_shortk, h->span = GC_ASPAN(v->a[0]);
	h->job0 = job0, h->bytes = v->bytes, h->pad_[0] = h->pad_[1] = 0;
}
// what part 3 starts from: the read's arrays in the block, records to be completed in place, anchors of the graph chains into res_a
__host__ __device__ inline void gcs_setup_p3(char *blk, mg128_t *res_a, gc_read_t *rd, gc_result_t *R, gc_split_t *sp)
{
	gcs_view_t v;
	const gcs_hdr_t *h = (const gcs_hdr_t*)blk;
	gcs_view(blk, h->n_b, h->n_c, h->n_u2, h->n_gc, &v);
	rd->a = v.a; // (part 3 only reads the anchors)
	R->n_gc = h->n_gc, R->n_lc = R->n_a = 0, R->gc = v.gc, R->lc = 0, R->a = res_a, R->n_gwfa = R->n_fast = 0, R->n_shortk = h->n_shortk;
	sp->c = v.c, sp->u2 = v.u2, sp->kept = v.kept, sp->n_c = h->n_c, sp->n_u2 = h->n_u2, sp->n_jobs = h->n_jobs, sp->done = 0;
}
__device__ __forceinline__ int gcs_job_class(const gc_job_t *q) // by the query gap of the bridge, longest first (the gap is what is known beforehand of a bridge's cost)
{
	const int32_t ql = (q->c1->qs + q->span) - (q->c0->qe - q->span);
	return ql >= 1600 ? 0 : ql >= 1100 ? 1 : ql >= 800 ? 2 : ql >= 600 ? 3 : ql >= 450 ? 4 : ql >= 300 ? 5 : ql >= 150 ? 6 : 7;
}

// A value every lane holds alike, as a SCALAR: branches on it are scalar branches.  The persistent loops below fetch their work with one lane's atomic; when the fetched index
// stayed a per-lane value (__shfl), the compiler treated every `continue` / `break` of the loop as divergent and restructured it into nested exec-mask loops -- [measured, round 4]
// in k_gchain_p1 the path of an early `continue` then came back WITHOUT a new fetch (the launch never ended on workloads with chain-less reads).
__device__ __forceinline__ int32_t gck_uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long gck_uni64(long long v) { const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)((unsigned long long)v >> 32)); return (long long)((unsigned long long)hi << 32 | lo); }
// The launch parameters (several hundred bytes of pointers and sizes) are copied to LDS at the start and read from there: as kernel arguments they are all loaded into scalar
// registers up front and stay live across the whole routine -- [measured, round 4] 130-430 spilled SGPRs per kernel, and a part-1 wavefront that took an early `continue`
// never fetched its next read (workloads with chain-less reads hung the launch).
#define GCK_ARGS_TO_LDS(has_io, has_split) GCK_ARGS_TO_LDS_##has_io
#define GCK_ARGS_TO_LDS_1 __shared__ gck_in_t in_lds_; __shared__ gck_out_t out_lds_; __shared__ gck_split_t S_lds_; in_lds_ = in_arg, out_lds_ = out_arg, S_lds_ = S_arg; \
	const gck_in_t &in = in_lds_; const gck_out_t &out = out_lds_; const gck_split_t &S = S_lds_
#define GCK_ARGS_TO_LDS_0 __shared__ gck_split_t S_lds_; S_lds_ = S_arg; const gck_split_t &S = S_lds_
#ifndef GC_AB_P1_WAVES
#define GC_AB_P1_WAVES 2
#endif
#ifndef GC_AB_P2_WAVES
#define GC_AB_P2_WAVES 2
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GC_AB_P1_WAVES, 8))) k_gchain_p1(gck_in_t in_arg, gck_out_t out_arg, gck_split_t S_arg, gc_graph_t G_arg, gc_par_t P_arg, char *arena_mem, int64_t arena_bytes)
{
	GCK_ARGS_TO_LDS(1, 1);
	const int lane = threadIdx.x;
	char *my_arena = arena_mem + (int64_t)blockIdx.x * arena_bytes;
	__shared__ gc_graph_t G_lds;
	__shared__ gc_par_t P_lds;
	__shared__ gc_arena_t A_lds;
	__shared__ gc_read_t rd_lds;
	__shared__ gc_result_t R_lds;
	__shared__ gc_split_t sp_lds;
	G_lds = G_arg, P_lds = P_arg;
	mga_wave_sync();
	const gc_graph_t &G = G_lds;
	const gc_par_t &P = P_lds;
	gc_arena_t &A = A_lds;
	gc_read_t &rd = rd_lds;
	gc_result_t &R = R_lds;
	gc_split_t &sp = sp_lds;
	for (;;) { // (one way through the body, no early `continue`: see gck_uni above)
		int slot = 0;
		if (lane == 0) slot = (int)atomicAdd(&out.ctl[0], 1ULL);
		slot = gck_uni(slot);
		if (slot >= gck_uni(in.n)) break;
		const int r = gck_uni(in.list ? in.list[slot] : slot);
		const int64_t off = gck_uni64(in.a_off[r]);
		const int32_t n_u = gck_uni(in.nu[r]), n_b = gck_uni(in.nb[r]);
		int32_t none_status = GC_OK, none_shortk = 0; // what the read's header says when it leaves no state behind
		int save = 0;
		mg128_t *work = 0;
		if (gck_uni(in.rflag && in.rflag[r] == 2)) none_status = MGA_GC_HOST; // its chains come from the host tree, not from here
		else if (n_u > 0 && n_b > 0) {
			gc_arena_init(&A, my_arena, arena_bytes, 0);
			if (out.ctl[15]) A.ticks = out.ctl + 16, A.tick_last = (long long)clock64();
			work = (mg128_t*)gc_alloc(&A, (int64_t)n_b * 16); // the chains' anchors: flags and minimizer ranks are written into this copy
			if (gck_uni(work == 0)) none_status = GC_E_ARENA;
			else {
				gck_copy_words(work, in.b + off, (int64_t)n_b * 16, lane);
				mga_wave_sync();
				rd.qlen = (int32_t)(in.q_off[r + 1] - in.q_off[r]), rd.hash = in.hash[r];
				rd.n_u = n_u, rd.u = in.u + off, rd.a = work;
				rd.n_mini = (int32_t)(in.mini_off[r + 1] - in.mini_off[r]), rd.mini_pos = in.mini + in.mini_off[r];
				rd.qseq = in.seq + in.q_off[r];
				R.gc = 0, R.lc = 0, R.a = 0;
				const int32_t status = gck_uni(gc_read_p1(&A, &G, &P, &rd, &R, &sp));
				if (lane == 0) atomicMax(&out.ctl[7], (unsigned long long)A.peak);
				if (status == GC_E_BUG) none_shortk = gck_uni(R.n_shortk);          // the reference's own bail-outs: the read gets no chains
				else if (status != GC_OK) none_status = status;                      // out of arena: the retry list
				else if (gck_uni(sp.done || R.n_gc == 0)) none_shortk = gck_uni(R.n_shortk);
				else save = 1;
			}
		}
		if (save) { // ---- the state leaves the arena as one block ----
			const int32_t n_jobs = gck_uni(sp.n_jobs);
			gcs_view_t v;
			gcs_view(0, n_b, sp.n_c, sp.n_u2, R.n_gc, &v); // (sizes first)
			const int64_t bytes = v.bytes;
			long long s_off = 0, job0 = 0, pos = 0;
			int ok = 1;
			if (lane == 0) {
				s_off = (long long)atomicAdd(&S.sctl[0], (unsigned long long)bytes);
				job0 = (long long)atomicAdd(&S.sctl[1], (unsigned long long)n_jobs);
				ok = s_off + bytes <= S.state_cap && job0 + n_jobs <= S.jobs_cap;
				if (ok) pos = (long long)atomicAdd(&S.sctl[3], 1ULL);
			}
			ok = gck_uni(ok), s_off = gck_uni64(s_off), job0 = gck_uni64(job0), pos = gck_uni64(pos);
			if (!ok) none_status = MGA_GC_E_POOL, save = 0;
			else {
				char *blk = S.state + s_off;
				gcs_view(blk, n_b, sp.n_c, sp.n_u2, R.n_gc, &v);
				gcs_write_arrays(&v, n_b, work, &sp, &R, lane, 64);
				mga_wave_sync();
				if (lane == 0) {
					gcs_write_header(&v, r, n_b, &sp, &R, job0);
					S.p3list[pos] = s_off;
					if (n_jobs > 0) gc_assemble_jobs(sp.n_u2, v.u2, v.kept, v.c, r, v.h->span, rd.qseq, S.jobs + job0); // the bridges, pointing into the block's copy of the chain records
				}
			}
		}
		if (!save && lane == 0) { // header of a read without graph chains, or of one that goes to the host / to the retry list
			mga_gc_hdr_t *H = &out.hdr[r];
			if (none_status != GC_OK && none_status != MGA_GC_HOST) out.retry[atomicAdd(&out.ctl[3], 1ULL)] = r;
			else if (none_shortk) atomicAdd(&out.ctl[5], (unsigned long long)none_shortk);
			H->n_gc = H->n_lc = H->n_a = 0, H->status = none_status, H->gc_off = H->lc_off = H->a_off = 0;
		}
		mga_wave_sync();
	}
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GC_AB_P2_WAVES, 8))) k_gchain_p2(gck_split_t S_arg, gc_graph_t G_arg, gc_par_t P_arg, char *arena_mem, int64_t arena_bytes, unsigned long long *ctl, long long only_job)
{
	GCK_ARGS_TO_LDS(0, 1);
	const int lane = threadIdx.x;
	char *my_arena = arena_mem + (int64_t)blockIdx.x * arena_bytes;
	__shared__ gc_graph_t G_lds;
	__shared__ gc_par_t P_lds;
	__shared__ gc_arena_t A_lds;
	G_lds = G_arg, P_lds = P_arg;
	mga_wave_sync();
	const gc_graph_t &G = G_lds;
	const gc_par_t &P = P_lds;
	gc_arena_t &A = A_lds;
	long long n_jobs = gck_uni64((long long)S.sctl[1]);
	if (n_jobs > S.jobs_cap) n_jobs = gck_uni64(S.jobs_cap);
	for (int cls = 0; cls < GCS_N_CLASS; ++cls) { // long query gaps first: the launch ends with the short ones
		const int quantum = cls < 4 ? 64 : cls < 6 ? 16 : 4; // jobs looked at per reservation (a class's jobs are a fraction of them; the last classes must not hand a wavefront a long run)
		for (;;) {
			long long base = 0;
			if (lane == 0) base = (long long)atomicAdd(&S.sctl[8 + cls], (unsigned long long)quantum);
			base = gck_uni64(base);
			if (base >= n_jobs) break;
			uint64_t todo = __ballot(lane < quantum && base + lane < n_jobs && gcs_job_class(&S.jobs[base + lane]) == cls && (only_job < 0 || base + lane == only_job));
			while (todo) {
				gc_job_t *q = &S.jobs[base + (__ffsll((long long)todo) - 1)];
				todo &= todo - 1;
				gc_bres_t b;
				const long long t_job0 = (long long)clock64();
				gc_arena_init(&A, my_arena, arena_bytes, 0);
				if (ctl[15]) A.ticks = ctl + 16, A.tick_last = (long long)clock64();
				gc_job_run(&A, &G, &P, q, &b); // (its status lands in the job)
				int32_t st = gck_uni(q->status);
				if (st == GC_JOB_OK && gck_uni(b.n_mid) > 0) {
					long long mo = 0;
					if (lane == 0) mo = (long long)atomicAdd(&S.sctl[2], (unsigned long long)b.n_mid);
					mo = gck_uni64(mo);
					if (mo + b.n_mid > S.mid_cap) st = GC_JOB_ARENA;
					else { gck_copy_words(S.mid + mo, b.mid, (int64_t)b.n_mid * 4, lane); if (lane == 0) q->mid_off = mo; }
				}
				if (lane == 0) { q->status = st; q->n_fast = (int32_t)(((long long)clock64() - t_job0) >> 10); atomicMax(&ctl[7], (unsigned long long)A.peak); } // (n_fast: the job's duration in 1024-cycle units, a profiling aid)
				mga_wave_sync();
			}
		}
	}
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GC_AB_WAVES_PER_EU, 8))) k_gchain_p3(gck_in_t in_arg, gck_out_t out_arg, gck_split_t S_arg, gc_graph_t G_arg, gc_par_t P_arg, char *arena_mem, int64_t arena_bytes)
{
	GCK_ARGS_TO_LDS(1, 1);
	const int lane = threadIdx.x;
	char *my_arena = arena_mem + (int64_t)blockIdx.x * arena_bytes;
	__shared__ gc_graph_t G_lds;
	__shared__ gc_par_t P_lds;
	__shared__ gc_arena_t A_lds;
	__shared__ gc_read_t rd_lds;
	__shared__ gc_result_t R_lds;
	__shared__ gc_split_t sp_lds;
	G_lds = G_arg, P_lds = P_arg;
	mga_wave_sync();
	const gc_graph_t &G = G_lds;
	const gc_par_t &P = P_lds;
	gc_arena_t &A = A_lds;
	gc_read_t &rd = rd_lds;
	gc_result_t &R = R_lds;
	gc_split_t &sp = sp_lds;
	const long long n_state = gck_uni64((long long)S.sctl[3]);
	for (;;) {
		long long k = 0;
		if (lane == 0) k = (long long)atomicAdd(&S.sctl[4], 1ULL);
		k = gck_uni64(k);
		if (k >= n_state) break;
		char *blk = S.state + gck_uni64(S.p3list[k]);
		const gcs_hdr_t *h = (const gcs_hdr_t*)blk;
		const int r = gck_uni(h->read);
		gc_arena_init(&A, my_arena, arena_bytes, 0);
		if (out.ctl[15]) A.ticks = out.ctl + 16, A.tick_last = (long long)clock64();
		mg128_t *res_a = (mg128_t*)gc_alloc(&A, (int64_t)h->n_b * 16); // anchors of the graph chains
		int32_t status = GC_E_ARENA;
		R.gc = 0, R.lc = 0;
		if (res_a) {
			rd.qlen = (int32_t)(in.q_off[r + 1] - in.q_off[r]), rd.hash = in.hash[r];
			rd.n_u = in.nu[r], rd.u = in.u + in.a_off[r];
			rd.n_mini = (int32_t)(in.mini_off[r + 1] - in.mini_off[r]), rd.mini_pos = in.mini + in.mini_off[r];
			rd.qseq = in.seq + in.q_off[r];
			gcs_setup_p3(blk, res_a, &rd, &R, &sp);
			mga_wave_sync();
			status = gc_read_p3(&A, &G, &P, &rd, &R, &sp, S.jobs + h->job0, S.mid);
			if (status == GC_E_BUG) status = GC_OK, R.n_gc = R.n_lc = R.n_a = 0;
		}
		gck_publish(out, r, lane, status, &R, res_a, (long long)A.peak);
		mga_wave_sync();
	}
}

extern "C" size_t mga_dev_gchain_arena_bytes(int tier)
{ // MGA_GC_ARENA_KB / MGA_GC_ARENA1_KB: scratch per wavefront of the first launch / of the retry launch (tests: small values push reads through the retry and on to the host)
	const char *e = getenv(tier == 0 ? "MGA_GC_ARENA_KB" : "MGA_GC_ARENA1_KB");
	if (e && atoi(e) > 0) return (size_t)atoi(e) << 10;
	return tier == 0 ? (size_t)1 << 20 : (size_t)256 << 20;
}
extern "C" int mga_dev_gchain_waves(int tier) { static int w0 = 0; if (w0 == 0) { const char *e = getenv("MGA_GC_WAVES"); w0 = e && atoi(e) > 0 ? atoi(e) : 2048; } return tier == 0 ? w0 : 24; } /* tier 0: 256 CUs x 4 SIMDs x 2 resident waves (246 VGPRs, no spills: [measured] same kernel time as 4 waves with 513 spills) */

static void gc_par_from_opt(const mg_mapopt_t *opt, int k, float pen_gap, gc_par_t *P)
{
	memset(P, 0, sizeof *P);
	P->k = k, P->bw = opt->bw, P->bw_long = opt->bw_long, P->max_gap = opt->max_gap;
	P->min_lc_cnt = opt->min_lc_cnt, P->lc_max_occ = opt->lc_max_occ, P->lc_max_trim = opt->lc_max_trim;
	P->max_gc_skip = opt->max_gc_skip, P->ref_bonus = opt->ref_bonus, P->min_gc_cnt = opt->min_gc_cnt, P->min_gc_score = opt->min_gc_score, P->gdp_max_ed = opt->gdp_max_ed;
	P->best_n = opt->best_n, P->sub_diff = opt->sub_diff;
	P->chn_pen_gap = pen_gap, P->mask_level = opt->mask_level, P->pri_ratio = opt->pri_ratio;
}

// One launch over n reads (d_list == NULL: all of them; otherwise the listed ones, tier 1 arenas).  Device pointers throughout.
// ctl: 8 x uint64 (zeroed by the caller before the FIRST launch of a chunk; a retry launch resets only the read counter).
extern "C" int mga_dev_gchain(mga_sctx_t *sc, const mga_didx_t *ix, const mg_mapopt_t *opt, int k, float pen_gap, int n, const int32_t *d_list, int tier,
							  const int64_t *d_a_off, const int32_t *d_nu, const int32_t *d_nb, const uint64_t *d_u, const mg128_t *d_b,
							  const int64_t *d_mini_off, const int32_t *d_mini, const int64_t *d_q_off, const char *d_seq, const uint32_t *d_hash, const int32_t *d_rflag,
							  mga_gc_hdr_t *d_hdr, void *d_gc_pool, int64_t gc_cap, mg_llchain_t *d_lc_pool, int64_t lc_cap, mg128_t *d_a_pool, int64_t a_cap,
							  unsigned long long *d_ctl, int32_t *d_retry)
{
	if (n <= 0) return 0;
	if (ix->d_arc == 0 || ix->d_gseq_rc == 0) { mga_set_error("graph chaining on the device needs the graph replica (mga_dev_graph_upload)"); return -1; }
	const size_t ab = mga_dev_gchain_arena_bytes(tier);
	int waves = mga_dev_gchain_waves(tier);
	if (waves > n) waves = n;
	mga_dbuf_t *arena = &sc->gc_arena[tier ? 1 : 0];
	if (mga_dbuf_reserve(arena, ab * (size_t)waves) < 0) return -1; /* (grow-only: a context that only ever maps single reads keeps a single arena) */
	gck_in_t in;
	gck_out_t out;
	gc_graph_t G;
	gc_par_t P;
	in.n = n, in.list = d_list, in.a_off = d_a_off, in.nu = d_nu, in.nb = d_nb, in.u = d_u, in.b = d_b, in.mini_off = d_mini_off, in.mini = d_mini;
	in.q_off = d_q_off, in.seq = d_seq, in.hash = d_hash, in.rflag = d_rflag;
	out.hdr = d_hdr, out.gc_pool = (gc_rec_t*)d_gc_pool, out.gc_cap = gc_cap, out.lc_pool = d_lc_pool, out.lc_cap = lc_cap, out.a_pool = d_a_pool, out.a_cap = a_cap;
	out.ctl = d_ctl, out.retry = d_retry;
	memset(&G, 0, sizeof G);
	G.arc = (const gc_arc_t*)ix->d_arc, G.idx = ix->d_arc_idx, G.seg_len = ix->d_seg_len, G.es = 0, G.seq_fw = ix->d_gseq, G.seq_rc = ix->d_gseq_rc, G.seq_off = ix->d_gseq_off;
	gc_par_from_opt(opt, k, pen_gap, &P);
	// MGA_GC_SPLIT=1: the three-launch form below for a chunk's first launch.  The default is the one-kernel form: in the pipeline (bench.py, 125000 reads per step,
	// device placement, three interleaved repetitions) it maps 2.94 / 2.99 / 3.03 Gbp/s against 2.67 / 2.64 / 2.76 for the three launches -- each of the three ends in its
	// own tail (p2's is its longest bridge) and the two per-read kernels run at the occupancy of the one-kernel form, see DESIGN.md "Graph chaining in three launches"
	const int split = getenv("MGA_GC_SPLIT") ? atoi(getenv("MGA_GC_SPLIT")) : 0; // (read per launch: one launch per chunk)
	if (tier == 0 && d_list == 0 && split) { // ---- three launches: per read / per bridge / per read (see above) ----
		static int w2 = 0;
		if (w2 == 0) { const char *e = getenv("MGA_GC_WAVES2"); w2 = e && atoi(e) > 0 ? atoi(e) : 2048; }
		const int waves1 = waves, waves2 = w2, waves3 = waves;
		const int wmax = waves1 > waves2 ? waves1 : waves2;
		// pools of the chunk: every read's state fits (anchors + records of all its chains), jobs <= linear chains <= graph-chain records; vertices of walks: by count
		const int64_t jobs_cap = gc_cap, mid_cap = gc_cap * 8 + (1 << 20);
		const int64_t state_cap = a_cap * 16 + gc_cap * (int64_t)(sizeof(gc_chain_t) + 16 + sizeof(gc_rec_t) + 48) + (int64_t)n * 128 + 4096;
		const size_t o_list = 256, o_jobs = o_list + (((size_t)n * 8 + 255) & ~(size_t)255), o_mid = o_jobs + (((size_t)jobs_cap * sizeof(gc_job_t) + 255) & ~(size_t)255),
					 o_state = o_mid + (((size_t)mid_cap * 4 + 255) & ~(size_t)255), total = o_state + (size_t)state_cap;
		if (mga_dbuf_reserve(arena, ab * (size_t)wmax) < 0 || mga_dbuf_reserve(&sc->gc_split, total) < 0) return -1;
		gck_split_t S;
		char *sb = (char*)sc->gc_split.p;
		S.sctl = (unsigned long long*)sb, S.p3list = (int64_t*)(sb + o_list), S.jobs = (gc_job_t*)(sb + o_jobs), S.jobs_cap = jobs_cap, S.mid = (int32_t*)(sb + o_mid), S.mid_cap = mid_cap;
		S.state = sb + o_state, S.state_cap = state_cap;
		static int dbg = -1; // MGA_GC_SPLIT_DEBUG=1: wait behind every part and print the chunk's counters
		if (dbg < 0) { const char *e = getenv("MGA_GC_SPLIT_DEBUG"); dbg = e ? atoi(e) : 0; }
#define GCS_DBG(what) do { if (dbg) { unsigned long long c_[16], k_[4] = { 0, 0, 0, 0 }; const double t_ = mga_wtime(); hipError_t e_ = hipErrorNotReady; \
			while (mga_wtime() - t_ < 4.0 && (e_ = hipStreamQuery((hipStream_t)sc->stream)) == hipErrorNotReady) {} \
			if (e_ == hipErrorNotReady) { hipStream_t s2_; (void)hipStreamCreateWithFlags(&s2_, hipStreamNonBlocking); (void)hipMemcpyAsync(k_, d_ctl, 32, hipMemcpyDeviceToHost, s2_); (void)hipMemcpyAsync(c_, sb, sizeof c_, hipMemcpyDeviceToHost, s2_); (void)hipStreamSynchronize(s2_); \
				fprintf(stderr, "[gc-split] %s: STILL RUNNING after 4 s; n %d waves %d/%d ctl[0..3] %llu %llu %llu %llu; state bytes %llu jobs %llu vertices %llu states %llu next3 %llu next2 %llu/%llu/%llu\n", what, n, waves1, waves2, k_[0], k_[1], k_[2], k_[3], c_[0], c_[1], c_[2], c_[3], c_[4], c_[8], c_[9], c_[15]); _exit(3); } \
			if (e_ == hipSuccess) e_ = hipMemcpy(c_, sb, sizeof c_, hipMemcpyDeviceToHost); \
			fprintf(stderr, "[gc-split] %s: %s after %.1f ms; n %d state bytes %llu jobs %llu vertices %llu states %llu next3 %llu next2 %llu/%llu/%llu\n", what, hipGetErrorString(e_), (mga_wtime() - t_) * 1e3, n, c_[0], c_[1], c_[2], c_[3], c_[4], c_[8], c_[9], c_[15]); } } while (0)
		MGA_HIP_CHECK(hipMemsetAsync(sb, 0, 256, (hipStream_t)sc->stream));
		GCS_DBG("start");
		mga_prof_begin(sc->stream, MGA_K_GCHAIN);
		hipLaunchKernelGGL(k_gchain_p1, dim3(waves1), dim3(64), 0, (hipStream_t)sc->stream, in, out, S, G, P, (char*)arena->p, (int64_t)ab);
		mga_prof_end(sc->stream, MGA_K_GCHAIN);
		GCS_DBG("part 1");
		mga_prof_begin(sc->stream, MGA_K_GCHAIN2);
		hipLaunchKernelGGL(k_gchain_p2, dim3(waves2), dim3(64), 0, (hipStream_t)sc->stream, S, G, P, (char*)arena->p, (int64_t)ab, d_ctl, -1LL);
		mga_prof_end(sc->stream, MGA_K_GCHAIN2);
		GCS_DBG("part 2");
		if (dbg >= 2) { // where part 2's time goes: the bridges by duration
			unsigned long long c_[4];
			if (hipMemcpy(c_, sb, sizeof c_, hipMemcpyDeviceToHost) == hipSuccess && c_[1] > 0) {
				const size_t nj = (size_t)(c_[1] < (unsigned long long)jobs_cap ? c_[1] : (unsigned long long)jobs_cap);
				gc_job_t *hj = (gc_job_t*)malloc(nj * sizeof(gc_job_t));
				gc_chain_t *hc = (gc_chain_t*)malloc(nj * 2 * sizeof(gc_chain_t));
				if (hipMemcpy(hj, S.jobs, nj * sizeof(gc_job_t), hipMemcpyDeviceToHost) == hipSuccess) {
					double sum = 0; long long mx = 0; size_t imx = 0, hist[12] = { 0 };
					for (size_t i_ = 0; i_ < nj; ++i_) { const long long t_ = (long long)hj[i_].n_fast << 10; sum += (double)t_; if (t_ > mx) mx = t_, imx = i_; int b_ = 0; long long x_ = t_ >> 14; while (x_ > 0 && b_ < 11) x_ >>= 1, ++b_; hist[b_]++; }
					(void)hipMemcpy(&hc[0], hj[imx].c0, sizeof(gc_chain_t), hipMemcpyDeviceToHost); (void)hipMemcpy(&hc[1], hj[imx].c1, sizeof(gc_chain_t), hipMemcpyDeviceToHost);
					fprintf(stderr, "[gc-split] part 2: %zu bridges, %.1f Mcycles in all (%.0f per wavefront of %d), longest %.2f Mcycles (query gap %d, ed %d, %d inner vertices, %d graph searches); by duration (<16k, <32k, ... cycles):", nj, sum * 1e-6, sum / waves2, waves2,
							mx * 1e-6, (hc[1].qs + hj[imx].span) - (hc[0].qe - hj[imx].span), hj[imx].ed, hj[imx].n_mid, hj[imx].n_shortk);
					for (int b_ = 0; b_ < 12; ++b_) fprintf(stderr, " %zu", hist[b_]);
					fprintf(stderr, "\n");
					{ // the longest bridge once more, ALONE on the device, with per-stage cycle sums and counts
						unsigned long long z_[32], one_ = 1, tk_[16];
						memset(z_, 0, sizeof z_);
						(void)hipMemcpy(sb + 64, z_, 64, hipMemcpyHostToDevice);          // job counters of the classes
						(void)hipMemcpy(d_ctl + 16, z_, 128, hipMemcpyHostToDevice); (void)hipMemcpy(d_ctl + 15, &one_, 8, hipMemcpyHostToDevice);
						const double t1_ = mga_wtime();
						hipLaunchKernelGGL(k_gchain_p2, dim3(1), dim3(64), 0, (hipStream_t)sc->stream, S, G, P, (char*)arena->p, (int64_t)ab, d_ctl, (long long)imx);
						(void)hipStreamSynchronize((hipStream_t)sc->stream);
						const double t2_ = mga_wtime();
						(void)hipMemcpy(tk_, d_ctl + 16, 128, hipMemcpyDeviceToHost);
						z_[0] = 0; (void)hipMemcpy(d_ctl + 15, z_, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_ctl + 16, z_, 128, hipMemcpyHostToDevice);
						fprintf(stderr, "[gc-split] the longest bridge alone: %.2f ms; steps %llu, cells in %llu, runs %llu, head cells %llu (%llu with those added on the way), cells out %llu, sorts %llu, finished ranges %llu + %llu new, %llu dedups on LDS; Mcycles: clear %.2f runs %.2f heads %.2f ranges %.2f dedup %.2f rest %.2f\n",
								(t2_ - t1_) * 1e3, tk_[1], tk_[2], tk_[3], tk_[4], tk_[9], tk_[6], tk_[10] & 0xfffffULL, tk_[7], tk_[15], tk_[10] >> 20, tk_[11] * 1e-6, tk_[12] * 1e-6, tk_[13] * 1e-6, tk_[0] * 1e-6, tk_[14] * 1e-6, tk_[8] * 1e-6);
					}
				}
				free(hj); free(hc);
			}
		}
		mga_prof_begin(sc->stream, MGA_K_GCHAIN3);
		hipLaunchKernelGGL(k_gchain_p3, dim3(waves3), dim3(64), 0, (hipStream_t)sc->stream, in, out, S, G, P, (char*)arena->p, (int64_t)ab);
		mga_prof_end(sc->stream, MGA_K_GCHAIN3);
		GCS_DBG("part 3");
		MGA_HIP_CHECK(hipGetLastError());
		return 0;
	}
	mga_prof_begin(sc->stream, MGA_K_GCHAIN);
	const int fast_bytes = 0; // (round 3's MGA_GC_LDS experiment -- the scratch of one graph search / GWFA call in 16 KB of LDS -- cost 66 -> 92 Gcycles per 16k reads: removed from the launch; the hook in gc_core.h stays)
	hipLaunchKernelGGL(k_gchain, dim3(waves), dim3(64), (size_t)fast_bytes, (hipStream_t)sc->stream, in, out, G, P, (char*)arena->p, (int64_t)ab, fast_bytes);
	mga_prof_end(sc->stream, MGA_K_GCHAIN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

extern "C" size_t mga_gc_rec_bytes(void) { return sizeof(gc_rec_t); }
static_assert(sizeof(gc_rec_t) == sizeof(mga_gc_rec_t) && offsetof(gc_rec_t, n_anchor) == offsetof(mga_gc_rec_t, n_anchor) && offsetof(gc_rec_t, qs) == offsetof(mga_gc_rec_t, qs)
			  && offsetof(gc_rec_t, ps) == offsetof(mga_gc_rec_t, ps) && offsetof(gc_rec_t, id) == offsetof(mga_gc_rec_t, id) && offsetof(gc_rec_t, parent) == offsetof(mga_gc_rec_t, parent)
			  && offsetof(gc_rec_t, hash) == offsetof(mga_gc_rec_t, hash), "mga_gc_rec_t (mga_dev.h) must mirror gc_rec_t");

// ---- flat records -> mg_gchains_t (host).  div (gchain1.c:299) and MAPQ (gcmisc.c:190-223) are computed here: they are the only
// places of the path that go through libm (log, logf), which has to be the host's. ----
static void gc_fill_public(mg_gchain_t *g, const gc_rec_t *r)
{
	memset(g, 0, sizeof *g);
	g->id = r->id, g->parent = r->parent, g->off = r->off, g->cnt = r->cnt, g->n_anchor = r->n_anchor, g->score = r->score;
	g->qs = r->qs, g->qe = r->qe, g->plen = r->plen, g->ps = r->ps, g->pe = r->pe, g->blen = r->blen, g->mlen = r->mlen;
	g->hash = r->hash, g->subsc = r->subsc, g->n_sub = r->n_sub, g->flt = (uint32_t)r->flt;
	g->div = -1.0f;
	if (r->cnt > 0) { // ratio of minimizers spanned to minimizers chained, per base of a minimizer
		const double ratio = r->n_mini >= r->n_anchor ? (double)r->n_mini / r->n_anchor : (double)r->n_anchor / r->n_mini;
		g->div = (float)(log(ratio) / r->q_span);
	}
}

static void gc_mapq(mg_gchains_t *gs, int qlen, int n_mz, int min_gc_score)
{
	if (gs->n_gc == 0) return;
	const int cap_sc = qlen < 100 ? qlen : 100;
	int cap_cnt = n_mz < 10 ? n_mz : 10;
	if (cap_cnt < 5) cap_cnt = 5;
	const float inv_sc = 1.0 / cap_sc, inv_cnt = 1.0 / cap_cnt;
	int64_t prim_sum = 0;
	for (int i = 0; i < gs->n_gc; ++i) if (gs->gc[i].parent == gs->gc[i].id) prim_sum += gs->gc[i].score;
	const float uniq = (float)prim_sum / (prim_sum + gs->rep_len);
	for (int i = 0; i < gs->n_gc; ++i) {
		mg_gchain_t *r = &gs->gc[i];
		int q = 0;
		if (r->parent == r->id) {
			const float by_score = (r->score > cap_sc ? 1.0f : r->score * inv_sc) * uniq;
			const float by_cnt = r->n_anchor > cap_cnt ? 1.0f : r->n_anchor * inv_cnt;
			const float pen = by_score < by_cnt ? by_score : by_cnt;
			const int sub = r->subsc > min_gc_score ? r->subsc : min_gc_score;
			const float x = (float)sub / r->score;
			q = (int)(pen * 40.0f * (1.0f - x) * logf(r->score));
			q -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			if (q < 0) q = 0;
			if (r->score > sub && q == 0) q = 1;
			if (q > 60) q = 60;
		}
		r->mapq = (uint32_t)q;
	}
}

extern "C" mg_gchains_t *mga_gchains_from_flat(int32_t n_gc, const void *gc_recs, int32_t n_lc, const mg_llchain_t *lc, int32_t n_a, const mg128_t *a,
											   int32_t rep_len, int32_t qlen, int32_t n_mz, int32_t min_gc_score)
{
	mg_gchains_t *gs = (mg_gchains_t*)calloc(1, sizeof(mg_gchains_t));
	const gc_rec_t *r = (const gc_rec_t*)gc_recs;
	gs->rep_len = rep_len;
	if (n_gc <= 0) return gs; // gchain1.c:460: a valid object without chains
	gs->n_gc = n_gc, gs->n_lc = n_lc, gs->n_a = n_a;
	gs->gc = (mg_gchain_t*)malloc((size_t)n_gc * sizeof(mg_gchain_t));
	gs->lc = (mg_llchain_t*)malloc((size_t)(n_lc > 0 ? n_lc : 1) * sizeof(mg_llchain_t));
	gs->a = (mg128_t*)malloc((size_t)(n_a > 0 ? n_a : 1) * sizeof(mg128_t));
	for (int32_t i = 0; i < n_gc; ++i) gc_fill_public(&gs->gc[i], &r[i]);
	memcpy(gs->lc, lc, (size_t)n_lc * sizeof(mg_llchain_t));
	if (n_a > 0) memcpy(gs->a, a, (size_t)n_a * sizeof(mg128_t));
	gc_mapq(gs, qlen, n_mz, min_gc_score);
	return gs;
}

// ---- the same routine on a host thread ----
extern "C" mg_gchains_t *mga_gchain_host_read(const mg_idx_t *gi, const int32_t *seg_len, const mg_mapopt_t *opt, float pen_gap, int32_t qlen, uint32_t hash,
											  int32_t n_u, const uint64_t *u, mg128_t *a, int32_t n_a, int32_t n_mini, const int32_t *mini_pos, const char *qseq,
											  int32_t rep_len, int32_t n_mz, int32_t *n_gwfa, int32_t *n_shortk)
{
	static __thread char *t_mem = 0;
	static __thread int64_t t_first = 1 << 20; /* the thread's arena: grows to what its reads have needed (a read that outgrew it chained malloc'ed blocks -- released below --, and
	                                            * the next one starts with room for that: [measured, round 4] block malloc / free per read was part of 15 % "rest" in the host profile) */
	const int64_t first = t_first;
	gc_arena_t A;
	gc_graph_t G;
	gc_par_t P;
	gc_read_t rd;
	gc_result_t R;
	if (t_mem == 0) t_mem = (char*)malloc((size_t)first);
	gc_arena_init(&A, t_mem, first, 1);
	memset(&G, 0, sizeof G);
	G.arc = (const gc_arc_t*)gi->g->arc, G.idx = gi->g->idx, G.seg_len = seg_len, G.es = gi->es;
	gc_par_from_opt(opt, gi->k, pen_gap, &P);
	rd.qlen = qlen, rd.hash = hash, rd.n_u = n_u, rd.u = u, rd.a = a, rd.n_mini = n_mini, rd.mini_pos = mini_pos, rd.qseq = qseq;
	R.a = (mg128_t*)malloc((size_t)(n_a > 0 ? n_a : 1) * sizeof(mg128_t));
	int rc;
	static int split_test = -1; // MGA_GC_SPLIT_TEST (CPU tests) = 1: the three-part form -- part 1, the bridges as jobs in REVERSE order in an arena of their own, part 3; = 3: + the redo path
	                            // of part 3; = 2: the device's hand-over -- part 1's state leaves through a block (gcs_*), its arena is scrubbed, the jobs and part 3 work from the block
	char *blk = 0;
	if (split_test < 0) { const char *e = getenv("MGA_GC_SPLIT_TEST"); split_test = e ? atoi(e) : 0; }
	if (!split_test) rc = gc_map_read(&A, &G, &P, &rd, &R);
	else {
		gc_split_t sp;
		rc = gc_read_p1(&A, &G, &P, &rd, &R, &sp);
		if (rc == GC_OK && !sp.done && (split_test != 2 || R.n_gc > 0)) {
			gc_job_t *jobs = (gc_job_t*)calloc((size_t)sp.n_jobs + 1, sizeof(gc_job_t));
			int32_t *pool = 0;
			int64_t n_pool = 0, m_pool = 0;
			if (split_test == 2) { // what k_gchain_p1 / p2 / p3 do, with one lane
				gcs_view_t v;
				gcs_view(0, n_a, sp.n_c, sp.n_u2, R.n_gc, &v);
				blk = (char*)malloc((size_t)v.bytes);
				gcs_view(blk, n_a, sp.n_c, sp.n_u2, R.n_gc, &v);
				gcs_write_arrays(&v, n_a, rd.a, &sp, &R, 0, 1);
				gcs_write_header(&v, 0, n_a, &sp, &R, 0);
				gc_assemble_jobs(sp.n_u2, v.u2, v.kept, v.c, 0, v.h->span, rd.qseq, jobs);
				memset(A.base, 0xA5, (size_t)A.top); A.top = 0; // part 1's arena is gone (its current block at least)
				memset(rd.a, 0xA5, (size_t)n_a * 16);           // ... and so is its working copy of the anchors
				memset(&sp, 0xA5, sizeof sp);
				gcs_setup_p3(blk, R.a, &rd, &R, &sp);
			} else gc_assemble_jobs(sp.n_u2, sp.u2, sp.kept, sp.c, 0, GC_ASPAN(rd.a[0]), rd.qseq, jobs);
			for (int32_t k = sp.n_jobs - 1; k >= 0 && rc == GC_OK; --k) {
				gc_arena_t A2;
				gc_bres_t b;
				char *m2 = (char*)malloc(1 << 16);
				gc_arena_init(&A2, m2, 1 << 16, 1);
				rc = gc_job_run(&A2, &G, &P, &jobs[k], &b);
				if (rc == GC_OK && jobs[k].status == GC_JOB_OK) {
					if (n_pool + b.n_mid > m_pool) { m_pool = (n_pool + b.n_mid) * 2 + 64; pool = (int32_t*)realloc(pool, (size_t)m_pool * 4); }
					jobs[k].mid_off = n_pool;
					for (int32_t t = 0; t < b.n_mid; ++t) pool[n_pool++] = b.mid[t];
				}
				gc_arena_free_blocks(&A2);
				free(m2);
			}
			if (split_test == 3) // (tests) every bridge between NEIGHBOURING chains is reported as "no walk": part 3 must then redo it where it meets it -- the reference's bytes again
				for (int32_t k = 0; k < sp.n_jobs; ++k) if (jobs[k].c1 == jobs[k].c0 + 1) jobs[k].status = GC_JOB_FAILED;
			if (rc == GC_OK) rc = gc_read_p3(&A, &G, &P, &rd, &R, &sp, jobs, pool);
			free(jobs); free(pool);
		}
	}
	if (rc != GC_OK) R.n_gc = R.n_lc = R.n_a = 0; // GC_E_BUG: no chains; GC_E_ARENA cannot happen on a growable arena short of malloc failing
	mg_gchains_t *gs = mga_gchains_from_flat(R.n_gc, R.gc, R.n_lc, R.lc, R.n_a, R.a, rep_len, qlen, n_mz, opt->min_gc_score);
	if (n_gwfa) *n_gwfa = R.n_gwfa;
	if (n_shortk) *n_shortk = R.n_shortk;
	free(R.a); free(blk);
	if (A.blocks) { /* outgrown: next time the first block holds everything */
		int64_t tot = first;
		for (const gc_block_t *bq = A.blocks; bq; bq = bq->prev) tot += bq->cap;
		gc_arena_free_blocks(&A);
		if (tot < ((int64_t)1 << 30)) { free(t_mem); t_mem = 0; t_first = tot + (tot >> 2); }
	}
	return gs;
}

#if defined(GC_HOST_PROF) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" void mga_gc_host_prof_dump(void)
{
	static const char *nm[16] = { "", "records", "cleanup", "index", "dp+shortk", "assemble(rest)", "post", "", "gwfa(rest)", "measure", "order", "gw:clear", "gw:runs", "gw:heads", "gw:dedup", "" };
	unsigned long long tot = 0;
	for (int q = 1; q < 15; ++q) tot += gc_host_ticks[q];
	fprintf(stderr, "[gc-host-prof] Mcycles:");
	for (int q = 1; q < 15; ++q) if (nm[q][0]) fprintf(stderr, " %s %.1f (%.0f%%)", nm[q], gc_host_ticks[q] * 1e-6, 100.0 * gc_host_ticks[q] / (tot ? tot : 1));
	fprintf(stderr, "\n");
}
#endif
#if defined(GC_STATS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" void mga_gc_stats_dump(void) { gc_stats_dump(); }
#endif

// ---- stage-level host entry points over the same core (CPU parity tests against the reference's mg_shortest_k / gfa_ed_step) ----
typedef struct { // mg_path_dst_t, mgpriv.h:40-52
	uint32_t v;
	int32_t target_dist;
	uint32_t target_hash;
	uint32_t meta:30, check_hash:1, inner:1;
	int32_t qlen;
	uint32_t n_path:31, is_0:1;
	int32_t path_end;
	int32_t dist;
	uint32_t hash;
} mga_path_dst_t;
typedef struct { uint32_t v, d; int32_t pre; } mga_pathv_t; // mg_pathv_t, mgpriv.h:54-57

static int32_t *gc_host_seg_len(const gfa_t *g)
{
	int32_t *len = (int32_t*)malloc((size_t)(g->n_seg + 1) * 4);
	for (uint32_t s = 0; s < g->n_seg; ++s) len[s] = g->seg[s].len;
	return len;
}

extern "C" mga_pathv_t *mga_shortest_k(const gfa_t *g, uint32_t src, int32_t n_dst, mga_path_dst_t *dst, int32_t max_dist, int32_t max_k, int32_t *n_pathv)
{
	gc_arena_t A;
	gc_graph_t G;
	gc_walkv_t *w = 0;
	int32_t n_w = 0;
	mga_pathv_t *ret = 0;
	if (n_pathv) *n_pathv = 0;
	if (n_dst <= 0) return 0;
	int32_t *seg_len = gc_host_seg_len(g);
	gc_dst_t *d = (gc_dst_t*)calloc((size_t)n_dst, sizeof(gc_dst_t));
	gc_arena_init(&A, malloc(1 << 16), 1 << 16, 1);
	char *first = A.base;
	memset(&G, 0, sizeof G);
	G.arc = (const gc_arc_t*)g->arc, G.idx = g->idx, G.seg_len = seg_len;
	for (int32_t i = 0; i < n_dst; ++i) {
		d[i].v = dst[i].v, d[i].target_dist = dst[i].target_dist, d[i].target_hash = dst[i].target_hash, d[i].meta = (int32_t)dst[i].meta;
		d[i].check_hash = dst[i].check_hash, d[i].inner = dst[i].inner, d[i].n_path = (int32_t)dst[i].n_path, d[i].is_0 = dst[i].is_0;
		d[i].path_end = dst[i].path_end, d[i].dist = dst[i].dist, d[i].hash = dst[i].hash;
	}
	const int rc = gc_shortest_k(&A, &G, src, n_dst, d, max_dist, max_k, n_pathv ? &w : 0, n_pathv ? &n_w : 0);
	for (int32_t i = 0; i < n_dst; ++i) {
		dst[i].n_path = (uint32_t)d[i].n_path, dst[i].is_0 = (uint32_t)d[i].is_0, dst[i].path_end = d[i].path_end, dst[i].dist = d[i].dist, dst[i].hash = d[i].hash;
	}
	if (rc == GC_OK && n_pathv && n_w > 0) {
		ret = (mga_pathv_t*)malloc((size_t)n_w * sizeof(mga_pathv_t));
		for (int32_t i = 0; i < n_w; ++i) ret[i].v = w[i].v, ret[i].d = w[i].d, ret[i].pre = w[i].pre;
		*n_pathv = n_w;
	}
	gc_arena_free_blocks(&A);
	free(first); free(d); free(seg_len);
	return ret;
}

extern "C" int32_t mga_gwfa_bridge(const gfa_t *g, const gfa_edseq_t *es, int32_t ql, const char *q, uint32_t v0, int32_t off0, uint32_t v1, int32_t off1,
								   int32_t max_lag, int32_t s_term, int32_t **path, int32_t *nv)
{
	gc_arena_t A;
	gc_graph_t G;
	int32_t ed = -1, *p = 0, n = 0;
	*path = 0, *nv = 0;
	int32_t *seg_len = gc_host_seg_len(g);
	gc_arena_init(&A, malloc(1 << 16), 1 << 16, 1);
	char *first = A.base;
	memset(&G, 0, sizeof G);
	G.arc = (const gc_arc_t*)g->arc, G.idx = g->idx, G.seg_len = seg_len, G.es = es;
	if (gc_gwfa(&A, &G, ql, q, v0, off0, v1, off1, max_lag, s_term, &ed, &p, &n) != GC_OK) ed = -1, n = 0;
	if (n > 0) { *path = (int32_t*)malloc((size_t)n * 4); memcpy(*path, p, (size_t)n * 4); *nv = n; }
	gc_arena_free_blocks(&A);
	free(first); free(seg_len);
	return ed;
}

// the assembly alone (mg_gchain_gen, gchain1.c:443-520: records, junctions, bridges, measuring, ordering) from given chain records: CPU parity tests against the reference's function
extern "C" mg_gchains_t *mga_gchain_gen_host(const gfa_t *g, const gfa_edseq_t *es, int32_t n_u, const uint64_t *u, const mg_lchain_t *lc, const mg128_t *a, int32_t n_a, uint32_t hash,
											 int32_t min_gc_cnt, int32_t min_gc_score, int32_t gdp_max_ed, const char *qseq, int32_t *rc_out)
{
	gc_arena_t A;
	gc_graph_t G;
	gc_par_t P;
	gc_result_t R;
	int32_t n_c = 0;
	for (int32_t i = 0; i < n_u; ++i) n_c += (int32_t)(uint32_t)u[i];
	int32_t *seg_len = gc_host_seg_len(g);
	gc_arena_init(&A, malloc(1 << 16), 1 << 16, 1);
	char *first = A.base;
	memset(&G, 0, sizeof G); memset(&P, 0, sizeof P); memset(&R, 0, sizeof R);
	G.arc = (const gc_arc_t*)g->arc, G.idx = g->idx, G.seg_len = seg_len, G.es = es;
	P.min_gc_cnt = min_gc_cnt, P.min_gc_score = min_gc_score, P.gdp_max_ed = gdp_max_ed;
	gc_chain_t *c = (gc_chain_t*)calloc((size_t)n_c + 1, sizeof(gc_chain_t));
	for (int32_t i = 0; i < n_c; ++i) {
		c[i].off = lc[i].off, c[i].cnt = lc[i].cnt, c[i].v = lc[i].v, c[i].rs = lc[i].rs, c[i].re = lc[i].re, c[i].qs = lc[i].qs, c[i].qe = lc[i].qe, c[i].score = lc[i].score;
		c[i].dist_pre = lc[i].dist_pre, c[i].hash_pre = lc[i].hash_pre, c[i].inner_pre = lc[i].inner_pre;
	}
	R.a = (mg128_t*)malloc((size_t)(n_a > 0 ? n_a : 1) * sizeof(mg128_t));
	int rc = gc_assemble(&A, &G, &P, n_u, u, c, a, hash, qseq, &R);
	if (rc_out) *rc_out = rc;
	if (rc != GC_OK) R.n_gc = R.n_lc = R.n_a = 0;
	mg_gchains_t *gs = mga_gchains_from_flat(R.n_gc, R.gc, R.n_lc, R.lc, R.n_a, R.a, 0, 100, 10, min_gc_score);
	free(R.a); free(c);
	gc_arena_free_blocks(&A);
	free(first); free(seg_len);
	return gs;
}

extern "C" void mg_gchain_free(mg_gchains_t *gs) // mgpriv.h:101 / gchain1.c:522-535: everything is malloc-owned (km == NULL)
{
	if (gs == 0) return;
	for (int32_t i = 0; i < gs->n_gc; ++i) { free(gs->gc[i].p); free(gs->gc[i].ds.ds); free(gs->gc[i].ds.off); }
	free(gs->gc); free(gs->a); free(gs->lc);
	free(gs);
}
