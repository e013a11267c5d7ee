// dev_rt.hip -- HIP runtime plumbing behind the C interface of mga_dev.h (gfx950 only).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mga_dev.h"
#include "dev_common.h"
#include <mutex>

static __thread char g_err[512];
static int g_dev_ok = -1, g_dev_id = 0;

extern "C" void mga_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	if (mg_verbose >= 1) fprintf(stderr, "[E::minigraph_amd] %s\n", g_err);
}

extern "C" const char *mga_last_error(void) { return g_err; }

extern "C" int mga_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int mga_dev_init(void)
{
	if (g_dev_ok > 0) return 0;
	if (g_dev_ok == 0) { mga_set_error("no HIP device available: the MI355X path cannot run and there is no CPU fallback"); return -1; }
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		mga_set_error("no HIP device available (%s): the MI355X path cannot run and there is no CPU fallback", hipGetErrorString(e));
		g_dev_ok = 0;
		return -1;
	}
	// one process per GPU: honour LOCAL_RANK when launched under torch.distributed.run
	int dev = 0;
	const char *lr = getenv("MGA_DEVICE");
	if (lr == 0) lr = getenv("LOCAL_RANK");
	if (lr) dev = atoi(lr) % n;
	if (hipSetDevice(dev) != hipSuccess) { mga_set_error("hipSetDevice(%d) failed", dev); g_dev_ok = 0; return -1; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, dev) == hipSuccess && mg_verbose >= 3)
		fprintf(stderr, "[M::minigraph_amd] device %d: %s (%s), %d CUs, %.1f GB\n", dev, prop.name, prop.gcnArchName,
				prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);
	g_dev_ok = 1, g_dev_id = dev;
	return 0;
}

extern "C" int mga_dev_bind_thread(void)
{
	if (mga_dev_init() < 0) return -1;
	MGA_HIP_CHECK(hipSetDevice(g_dev_id));
	return 0;
}

// pinned staging of small read-backs, see mga_d2h_s()
static size_t g_stage_bytes = 1 << 20, g_stage_max = 256 << 10; // MGA_STAGE_KB=<n>: copies of up to n KB are staged (block of 4n KB)
#define MGA_STAGE_BYTES g_stage_bytes
#define MGA_STAGE_MAX   g_stage_max
#define MGA_STAGE_SLOTS 64
struct stage_ent_t { void *dst; size_t off, bytes; };
struct stage_t { char *buf; size_t used; int n; stage_ent_t e[MGA_STAGE_SLOTS]; };

extern "C" mga_sctx_t *mga_sctx_create(void)
{
	if (mga_dev_init() < 0) return 0;
	mga_sctx_t *sc = (mga_sctx_t*)calloc(1, sizeof(mga_sctx_t));
	hipStream_t st;
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { free(sc); mga_set_error("hipStreamCreate failed"); return 0; }
	sc->stream = (void*)st;
	for (int i = 0; i < MGA_WFA_MAX_TIER; ++i) {
		hipStream_t t; hipEvent_t e;
		if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { mga_set_error("hipStreamCreate failed"); return 0; }
		sc->tier_stream[i] = (void*)t, sc->ev_done[i] = (void*)e;
	}
	{ hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 0; sc->ev_ready = (void*)e; }
	{ hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 0; sc->ev_sync = (void*)e; }
	{
		stage_t *S = (stage_t*)calloc(1, sizeof(stage_t));
		{ const char *e = getenv("MGA_STAGE_KB"); if (e && atoi(e) >= 0) g_stage_max = (size_t)atoi(e) << 10, g_stage_bytes = g_stage_max * 4 + 4096; }
		S->buf = (char*)mga_hmalloc_pinned(MGA_STAGE_BYTES);
		if (S->buf == 0) { free(S); return 0; }
		sc->stage = S;
	}
	return sc;
}

extern "C" void mga_sctx_destroy(mga_sctx_t *sc)
{
	if (sc == 0) return;
	(void)hipStreamSynchronize((hipStream_t)sc->stream);
	for (int i = 0; i < 10; ++i) mga_dbuf_free(&sc->wfa_ws[i]);
	for (int i = 0; i < 8; ++i) mga_dbuf_free(&sc->wfa_tbuf[i]);
	mga_dbuf_free(&sc->wfa_cnt);
	mga_dbuf_free(&sc->scan_tmp); mga_dbuf_free(&sc->txt_cnt); mga_dbuf_free(&sc->txt_off); mga_dbuf_free(&sc->txt_vwb); mga_dbuf_free(&sc->txt_el);
	mga_dbuf_free(&sc->wfa_list[0]); mga_dbuf_free(&sc->wfa_list[1]); mga_dbuf_free(&sc->wfa_key); mga_dbuf_free(&sc->wfa_ctl); mga_dbuf_free(&sc->wfa_fb);
	mga_dbuf_free(&sc->fb_prob); mga_dbuf_free(&sc->fb_res); mga_dbuf_free(&sc->sk_planes);
	mga_dbuf_free(&sc->gc_arena[0]); mga_dbuf_free(&sc->gc_arena[1]); mga_dbuf_free(&sc->gc_split);
	for (int i = 0; i < MGA_WFA_MAX_TIER; ++i) { (void)hipStreamDestroy((hipStream_t)sc->tier_stream[i]); (void)hipEventDestroy((hipEvent_t)sc->ev_done[i]); }
	(void)hipEventDestroy((hipEvent_t)sc->ev_ready); (void)hipEventDestroy((hipEvent_t)sc->ev_sync);
	if (sc->stage) { mga_hfree_pinned(((stage_t*)sc->stage)->buf); free(sc->stage); }
	(void)hipStreamDestroy((hipStream_t)sc->stream);
	free(sc);
}

// The tiers of one chunk run one after another on the context's stream: [measured] 1.95 vs 1.85 Gbp/s against running them
// concurrently on their own streams -- co-resident tiers halve each other's occupancy, while the tails of the wide tiers are
// filled by the kernels of the OTHER chunks in the pipeline anyway.  MGA_WFA_CONCURRENT=1 brings the per-tier streams back.
static int wfa_serial(void) { return 1; } // (round 3: a sweep of the ladder is one chain of launches -- a rung reads what the rungs before it appended)
extern "C" int mga_wfa_tiers_serial(void) { return wfa_serial(); }
extern "C" void *mga_wfa_stream(mga_sctx_t *sc, int slot) { return wfa_serial() ? sc->stream : sc->tier_stream[slot % MGA_WFA_MAX_TIER]; }

extern "C" int mga_wfa_fork(mga_sctx_t *sc)
{
	MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_ready, (hipStream_t)sc->stream));
	for (int i = 0; i < MGA_WFA_MAX_TIER; ++i) MGA_HIP_CHECK(hipStreamWaitEvent((hipStream_t)sc->tier_stream[i], (hipEvent_t)sc->ev_ready, 0));
	return 0;
}

extern "C" int mga_wfa_join(mga_sctx_t *sc)
{
	for (int i = 0; i < MGA_WFA_MAX_TIER; ++i) {
		MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_done[i], (hipStream_t)sc->tier_stream[i]));
		MGA_HIP_CHECK(hipStreamWaitEvent((hipStream_t)sc->stream, (hipEvent_t)sc->ev_done[i], 0));
	}
	return 0;
}

extern "C" mga_sctx_t *mga_sctx_default(void)
{
	static mga_sctx_t *g_def = 0;
	if (g_def == 0) g_def = mga_sctx_create();
	return g_def;
}

extern "C" int mga_h2d_s(mga_sctx_t *sc, void *d, const void *h, size_t bytes)
{
	if (bytes == 0) return 0;
	MGA_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)sc->stream));
	return 0;
}

// Small read-backs (counters, offsets) usually land in ordinary host variables.  An async copy to PAGEABLE memory is
// staged and waited for inside the runtime -- a spinning wait for everything queued before it on the stream.  So copies
// of up to MGA_STAGE_MAX bytes go to a pinned staging block of the context first and are handed to their destination
// by the next mga_ssync(); larger destinations must be pinned (mga_hbuf_t).

extern "C" int mga_d2h_s(mga_sctx_t *sc, void *h, const void *d, size_t bytes)
{
	if (bytes == 0) return 0;
	stage_t *S = (stage_t*)sc->stage;
	if (S && bytes <= MGA_STAGE_MAX && S->n < MGA_STAGE_SLOTS && S->used + bytes <= MGA_STAGE_BYTES) {
		stage_ent_t *e = &S->e[S->n++];
		e->dst = h, e->off = S->used, e->bytes = bytes;
		S->used += (bytes + 63) & ~(size_t)63;
		MGA_HIP_CHECK(hipMemcpyAsync(S->buf + e->off, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)sc->stream));
		return 0;
	}
	MGA_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)sc->stream));
	return 0;
}

static void stage_deliver(mga_sctx_t *sc)
{
	stage_t *S = (stage_t*)sc->stage;
	if (S == 0) return;
	for (int i = 0; i < S->n; ++i) memcpy(S->e[i].dst, S->buf + S->e[i].off, S->e[i].bytes);
	S->n = 0, S->used = 0;
}

extern "C" int mga_dmemset_s(mga_sctx_t *sc, void *d, int v, size_t bytes)
{
	if (bytes == 0) return 0;
	MGA_HIP_CHECK(hipMemsetAsync(d, v, bytes, (hipStream_t)sc->stream));
	return 0;
}

// Waiting without burning a core: hipStreamSynchronize() spins, and [measured] so does hipEventSynchronize() on an event
// created with hipEventBlockingSync under this runtime (4 pipeline threads waiting = 2 CPU-s per 0.55 s step).  The
// host stages of the other chunks need every core the (often CPU-quota-limited) box gives us, so poll the event and
// sleep in between; a sync happens ~12 times per chunk of ~45 ms, the added latency (<= 100 us each) is noise.
extern "C" void mga_cpu_note(int which, int64_t ns);
extern "C" int64_t mga_cpu_now(void);
extern "C" int mga_ssync(mga_sctx_t *sc)
{
	struct cpu_scope_t { int64_t t0; cpu_scope_t() : t0(mga_cpu_now()) {} ~cpu_scope_t() { if (t0) mga_cpu_note(12, mga_cpu_now() - t0); } } cpu_scope; // MGA_DEBUG_PIPE accounting
	MGA_HIP_CHECK(hipEventRecord((hipEvent_t)sc->ev_sync, (hipStream_t)sc->stream));
	struct timespec ts = { 0, 20000 };
	for (int spin = 0;; ++spin) {
		const hipError_t e = hipEventQuery((hipEvent_t)sc->ev_sync);
		if (e == hipSuccess) break;
		if (e != hipErrorNotReady) { mga_set_error("HIP error while waiting for the stream: %s", hipGetErrorString(e)); return -1; }
		if (spin < 3) continue; // the copy or kernel may be just about done
		nanosleep(&ts, 0);
		if (ts.tv_nsec < 100000) ts.tv_nsec += 20000;
	}
	stage_deliver(sc);
	return 0;
}

// error path of a pipeline stage: wait for whatever is still queued on the stream and DROP the staged read-backs -- their
// destinations (stack variables, per-chunk arrays) are about to go away, and the next mga_ssync() on this context must not
// deliver into them
extern "C" void mga_sctx_abort(mga_sctx_t *sc)
{
	if (sc == 0) return;
	(void)hipStreamSynchronize((hipStream_t)sc->stream);
	stage_t *S = (stage_t*)sc->stage;
	if (S) S->n = 0, S->used = 0;
}

extern "C" int mga_hbuf_reserve(mga_hbuf_t *b, size_t bytes)
{
	if (bytes <= b->cap && b->p) return 0;
	if (b->p) (void)hipHostFree(b->p);
	b->p = 0, b->cap = 0;
	size_t want = bytes + (bytes >> 2) + 4096;
	if (hipHostMalloc(&b->p, want, hipHostMallocDefault) != hipSuccess) { b->p = 0; mga_set_error("hipHostMalloc(%zu) failed", want); return -1; }
	b->cap = want;
	return 0;
}

extern "C" void mga_hbuf_free(mga_hbuf_t *b) { if (b->p) (void)hipHostFree(b->p); b->p = 0, b->cap = 0; }

extern "C" void *mga_dmalloc(size_t bytes)
{
	void *p = 0;
	if (bytes == 0) bytes = 16;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) { mga_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return 0; }
	return p;
}

extern "C" void mga_dfree(void *p) { if (p) (void)hipFree(p); }

extern "C" int mga_h2d(void *d, const void *h, size_t bytes)
{
	if (bytes == 0) return 0;
	MGA_HIP_CHECK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
	return 0;
}

extern "C" int mga_d2h(void *h, const void *d, size_t bytes)
{
	if (bytes == 0) return 0;
	MGA_HIP_CHECK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int mga_dmemset(void *d, int v, size_t bytes)
{
	if (bytes == 0) return 0;
	MGA_HIP_CHECK(hipMemset(d, v, bytes));
	return 0;
}

extern "C" int mga_dsync(void)
{
	MGA_HIP_CHECK(hipDeviceSynchronize());
	return 0;
}

extern "C" void *mga_hmalloc_pinned(size_t bytes)
{
	void *p = 0;
	if (bytes == 0) bytes = 16;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return 0;
	return p;
}

extern "C" void mga_hfree_pinned(void *p) { if (p) (void)hipHostFree(p); }

// ---- page-locking of host buffers the LIBRARY owns (the GAF output buffer of mga_map_files_shard, handed back for reuse from step to step): the registry
// lives here so that every path that frees or reallocates such a buffer -- mga_free(), the writer's realloc, a failed job -- unregisters it first
// (hipHostRegister from Python on a malloc'ed block the C side later realloc'ed or freed left stale registrations behind) ----
static struct { void *p; size_t n; } g_pins[64];
static int g_n_pins = 0;
static std::mutex g_pin_mtx;

extern "C" void mga_host_unpin(void *p)
{
	if (p == 0) return;
	std::lock_guard<std::mutex> lk(g_pin_mtx);
	for (int i = 0; i < g_n_pins; ++i)
		if (g_pins[i].p == p) { (void)hipHostUnregister(p); g_pins[i] = g_pins[--g_n_pins]; return; }
}

extern "C" int mga_host_pin(void *p, size_t bytes) // 0: registered (or already was, with at least `bytes`); -1: left pageable
{
	if (p == 0 || bytes == 0 || mga_dev_init() < 0) return -1;
	std::lock_guard<std::mutex> lk(g_pin_mtx);
	for (int i = 0; i < g_n_pins; ++i)
		if (g_pins[i].p == p) {
			if (g_pins[i].n >= bytes) return 0;
			(void)hipHostUnregister(p); g_pins[i] = g_pins[--g_n_pins];
			break;
		}
	if (g_n_pins == 64 || hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
	g_pins[g_n_pins].p = p, g_pins[g_n_pins].n = bytes, ++g_n_pins;
	return 0;
}

extern "C" double mga_wtime(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

extern "C" int mga_dbuf_reserve(mga_dbuf_t *b, size_t bytes)
{
	if (bytes <= b->cap && b->p) return 0;
	if (b->p) (void)hipFree(b->p);
	b->p = 0, b->cap = 0;
	size_t want = bytes + (bytes >> 3) + 256;
	b->p = mga_dmalloc(want);
	if (b->p == 0) return -1;
	b->cap = want;
	return 0;
}

extern "C" void mga_dbuf_free(mga_dbuf_t *b)
{
	if (b->p) (void)hipFree(b->p);
	b->p = 0, b->cap = 0;
}

// ---- exclusive scan int32 -> int64 (single workgroup of 256 threads; n up to a few 10^4: the per-read counts of a chunk) ----
// 256 threads, not 1024: in the pipeline this kernel is launched while persistent WFA workgroups of other chunks hold most wave slots, and a 16-wave
// workgroup then waits for a whole CU to drain ([measured] round 2: 1.2-1.7 ms per call, 3.6 % of the stream time, for 20 us of work); four waves fit
// into the gaps.  Each thread takes 4 consecutive counts per trip, so a trip still covers 1024.
__global__ void __launch_bounds__(256) k_scan_i32_i64(const int32_t *__restrict__ cnt, int64_t n, int64_t *__restrict__ off)
{
	__shared__ int64_t wsum[4];
	__shared__ int64_t carry;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int64_t base = 0; base < n; base += 1024) {
		const int64_t i0 = base + (int64_t)threadIdx.x * 4;
		int64_t v[4], s = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) { v[r] = i0 + r < n ? (int64_t)cnt[i0 + r] : 0; s += v[r]; }
		int64_t x = s;
		for (int d = 1; d < 64; d <<= 1) { // inclusive scan of the threads' sums inside the wave
			int64_t y = __shfl_up(x, d);
			if (lane >= d) x += y;
		}
		if (lane == 63) wsum[wid] = x;
		__syncthreads();
		int64_t run = carry + x - s;
		for (int w = 0; w < wid; ++w) run += wsum[w];
#pragma unroll
		for (int r = 0; r < 4; ++r) { if (i0 + r < n) off[i0 + r] = run; run += v[r]; }
		__syncthreads();
		if (threadIdx.x == 255) carry = run;
		__syncthreads();
	}
	if (threadIdx.x == 0) off[n] = carry;
}

// large inputs: tiles of 8192 counts per workgroup -- tile sums, scan of the tile sums (one small workgroup), tile-local
// scans with the tile's base.  (The single-workgroup kernel above took 1.6 ms for the 5*10^5 gap-filling problems of a chunk.)
#define SCAN_TILE 8192
__global__ void __launch_bounds__(1024) k_scan_tile_sum(const int32_t *__restrict__ cnt, int64_t n, int64_t *__restrict__ bsum)
{
	__shared__ int64_t ws[16];
	const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
	int64_t s = 0;
#pragma unroll
	for (int r = 0; r < 8; ++r) if (t0 + r < n) s += cnt[t0 + r];
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) { int64_t t = 0; for (int w = 0; w < 16; ++w) t += ws[w]; bsum[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(1024) k_scan_bsum(int64_t *__restrict__ bsum, int nb) // in place, exclusive; total at bsum[nb]
{
	__shared__ int64_t wsum[16];
	__shared__ int64_t carry;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = 0; base < nb; base += 1024) {
		const int i = base + threadIdx.x;
		int64_t v = i < nb ? bsum[i] : 0, x = v;
		for (int d = 1; d < 64; d <<= 1) { int64_t y = __shfl_up(x, d); if (lane >= d) x += y; }
		if (lane == 63) wsum[wid] = x;
		__syncthreads();
		if (wid == 0) {
			int64_t s = lane < 16 ? wsum[lane] : 0, t = s;
			for (int d = 1; d < 16; d <<= 1) { int64_t y = __shfl_up(t, d); if (lane >= d) t += y; }
			if (lane < 16) wsum[lane] = t - s;
		}
		__syncthreads();
		const int64_t c = carry;
		if (i < nb) bsum[i] = c + wsum[wid] + x - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry = c + wsum[wid] + x;
		__syncthreads();
	}
	if (threadIdx.x == 0) bsum[nb] = carry;
}

__global__ void __launch_bounds__(1024) k_scan_tile_write(const int32_t *__restrict__ cnt, int64_t n, const int64_t *__restrict__ bsum, int nb, int64_t *__restrict__ off)
{
	__shared__ int64_t wsum[16];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
	int32_t v[8];
	int64_t s = 0;
#pragma unroll
	for (int r = 0; r < 8; ++r) { v[r] = t0 + r < n ? cnt[t0 + r] : 0; s += v[r]; }
	int64_t x = s;
	for (int d = 1; d < 64; d <<= 1) { int64_t y = __shfl_up(x, d); if (lane >= d) x += y; }
	if (lane == 63) wsum[wid] = x;
	__syncthreads();
	int64_t run = bsum[blockIdx.x] + x - s;
	for (int w = 0; w < wid; ++w) run += wsum[w];
#pragma unroll
	for (int r = 0; r < 8; ++r) { if (t0 + r < n) off[t0 + r] = run; run += v[r]; }
	if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = bsum[nb];
}

extern "C" int mga_dev_scan_i32_to_i64(mga_sctx_t *sc, const int32_t *d_cnt, int64_t n, int64_t *d_off)
{
	hipStream_t st = (hipStream_t)sc->stream;
	mga_prof_begin(sc->stream, MGA_K_SCAN);
	if (n <= 4 * SCAN_TILE) hipLaunchKernelGGL(k_scan_i32_i64, dim3(1), dim3(256), 0, st, d_cnt, n, d_off);
	else {
		const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
		if (mga_dbuf_reserve(&sc->scan_tmp, (size_t)(nb + 1) * 8) < 0) return -1;
		hipLaunchKernelGGL(k_scan_tile_sum, dim3(nb), dim3(1024), 0, st, d_cnt, n, (int64_t*)sc->scan_tmp.p);
		hipLaunchKernelGGL(k_scan_bsum, dim3(1), dim3(1024), 0, st, (int64_t*)sc->scan_tmp.p, nb);
		hipLaunchKernelGGL(k_scan_tile_write, dim3(nb), dim3(1024), 0, st, d_cnt, n, (const int64_t*)sc->scan_tmp.p, nb, d_off);
	}
	mga_prof_end(sc->stream, MGA_K_SCAN);
	MGA_HIP_CHECK(hipGetLastError());
	return 0;
}

// ---- per-kernel HIP-event timing, recorded on the stream the kernel is launched on ----
#define PROF_MAX_PENDING 4096
static struct {
	int enabled;
	hipEvent_t ev[PROF_MAX_PENDING][2];
	int kid[PROF_MAX_PENDING];
	int n_pending, n_created;
	double ms[MGA_K_N];
	int64_t launches[MGA_K_N];
	std::mutex mtx;
} g_prof;
static __thread int t_prof_slot = -1;

extern "C" void mga_prof_enable(int on) { g_prof.enabled = on; }

static void prof_collect_locked(void)
{
	for (int i = 0; i < g_prof.n_pending; ++i) {
		float ms = 0.f;
		if (hipEventSynchronize(g_prof.ev[i][1]) == hipSuccess && hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]) == hipSuccess)
			g_prof.ms[g_prof.kid[i]] += ms, ++g_prof.launches[g_prof.kid[i]];
	}
	g_prof.n_pending = 0;
}

extern "C" void mga_prof_collect(void) { std::lock_guard<std::mutex> lk(g_prof.mtx); prof_collect_locked(); }

extern "C" void mga_prof_begin(void *stream, int kid)
{
	t_prof_slot = -1;
	if (!g_prof.enabled) return;
	std::lock_guard<std::mutex> lk(g_prof.mtx);
	if (g_prof.n_pending == PROF_MAX_PENDING) return; // full: this launch goes untimed (collected at the next mga_prof_get)
	int i = g_prof.n_pending;
	if (i >= g_prof.n_created) {
		if (hipEventCreate(&g_prof.ev[i][0]) != hipSuccess || hipEventCreate(&g_prof.ev[i][1]) != hipSuccess) { g_prof.enabled = 0; return; }
		g_prof.n_created = i + 1;
	}
	g_prof.kid[i] = kid;
	(void)hipEventRecord(g_prof.ev[i][0], (hipStream_t)stream);
	t_prof_slot = i;
	++g_prof.n_pending;
}

extern "C" void mga_prof_end(void *stream, int kid)
{
	(void)kid;
	if (t_prof_slot < 0) return;
	std::lock_guard<std::mutex> lk(g_prof.mtx);
	(void)hipEventRecord(g_prof.ev[t_prof_slot][1], (hipStream_t)stream);
	t_prof_slot = -1;
}

extern "C" void mga_prof_get(double *ms, int64_t *launches, int reset)
{
	std::lock_guard<std::mutex> lk(g_prof.mtx);
	prof_collect_locked();
	for (int k = 0; k < MGA_K_N; ++k) { ms[k] = g_prof.ms[k]; launches[k] = g_prof.launches[k]; }
	if (reset) { memset(g_prof.ms, 0, sizeof g_prof.ms); memset(g_prof.launches, 0, sizeof g_prof.launches); }
}
