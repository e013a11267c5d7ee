/*
 * mga_host.h -- internal declarations shared by the host-side C sources (not part of the public ABI).
 */
#ifndef MGA_HOST_H
#define MGA_HOST_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/minigraph_amd.h"
#include "mga_dev.h"

#define MGA_MALLOC(type, n) ((type*)malloc((size_t)(n) * sizeof(type)))
#define MGA_CALLOC(type, n) ((type*)calloc((size_t)(n), sizeof(type)))
#define MGA_REALLOC(type, p, n) ((type*)realloc((p), (size_t)(n) * sizeof(type)))
#define MGA_GROW(type, p, n, m) do { if ((n) >= (m)) { (m) = (m) ? (m) + ((m) >> 1) + 8 : 16; (p) = MGA_REALLOC(type, (p), (m)); } } while (0)

/* ---- exact klib radix-sort permutation on the host (ksortx.c) ----
 * The reference's in-place byte radix sort (ksort.h:112-162) is unstable for n > 64; wherever the
 * reference sorts records whose keys may tie, the product must apply the same permutation.
 * mga_ksort_perm() computes it: on return perm[i] = index (in the input order) of the record that ends
 * up at position i.  key_bytes = sizeof_key of the reference instantiation (8, or 4 for radix_sort_gc). */
void mga_ksort_perm(int64_t n, const uint64_t *key, int key_bytes, int64_t *perm);
void mga_ksort_128x(int64_t n, mg128_t *a);          /* radix_sort_128x, misc.c:9-10 */
void mga_ksort_u64(int64_t n, uint64_t *a);          /* radix_sort_gfa64, gfa-base.c:13-14 (ties are identical values) */

/* ---- hashes (khashl.h:321-346) ---- */
static inline uint32_t mga_hash_u32(uint32_t key)
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
	key ^= (key >> 6);   key += ~(key << 11); key ^= (key >> 16);
	return key;
}
static inline uint32_t mga_hash_str(const char *s)
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}

/* mg_log2 of mgpriv.h:63-71 (valid for x >= 2) */
static inline float mga_log2f(float x)
{
	union { float f; uint32_t i; } z;
	float r;
	z.f = x;
	r = (float)((int32_t)(z.i >> 23 & 255) - 128);
	z.i &= ~(255U << 23);
	z.i += 127U << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

extern unsigned char mga_comp_table[256]; /* IUPAC complement, gfa-base.c:509-526 */
extern unsigned char mga_nt4_table[256];  /* seq_nt4_table, sketch.c:9-26 */
void mga_tables_init(void);

/* ---- hidden part of mg_idx_t (index.c) ---- */
struct mg_idx_bucket_s {
	mga_didx_t dev;          /* replica in HBM */
	int64_t n_keys, n_mz;    /* distinct minimizers / total occurrences */
	int64_t *occ_hist;       /* occ_hist[c] = number of distinct minimizers occurring c times, c <= max_occ_seen */
	int64_t max_occ_seen;
	mga_stats_t st;
	char *gaf_out;           /* GAF text of the last mga_map_reads() pass; grow-only, owned by the index */
	int64_t gaf_cap;
	void *stream;            /* mga_stream_t of the single-batch entry points (mapper.c), created on first use */
	void *mf_cache;          /* read batches (pinned) and output buffers of mg_map_files jobs on this index (mapfiles.c), reused from job to job */
	/* an index loaded from a graph image (image.c) owns these: */
	gfa_t *img_g; void *img_map; size_t img_map_bytes; char *img_rc; int64_t *img_off;
};
void mga_graph_image_release(struct mg_idx_bucket_s *B);
mg_idx_t *mga_idx_from_cat(gfa_t *g, const mg_idxopt_t *io, int n_threads, const char *cat, const int64_t *off, const int32_t *seg_len, int64_t tot, char *es_rc);
mg_idx_t *mga_idx_hostpart_blob(gfa_t *g, const mg_idxopt_t *io, int n_threads, const int64_t *off, char *rc);
int mga_h2d_big(void *d, const void *h, size_t bytes, int n_threads); /* pageable host memory -> HBM through pinned staging blocks filled by several threads */

/* ---- the chunk pipeline as a persistent object (mapper.c) ---- */
typedef struct mga_stream_s mga_stream_t;
#define MGA_SB_FIRST 1   /* first batch of a job: small chunks first (pipeline fill) */
#define MGA_SB_LAST  2   /* last batch: small chunks last (pipeline drain) */
mga_stream_t *mga_stream_open(const mg_idx_t *gi, const mg_mapopt_t *opt, int n_threads);
int mga_stream_submit(mga_stream_t *S, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs, int want_gaf,
					  const char *d_seq, const int64_t *q_off, int seqs_pinned, int flags, char *out, int64_t out_cap, void *user);
int mga_stream_collect(mga_stream_t *S, char **out, int64_t *out_len, int64_t *out_cap, void **user);
void mga_stream_close(mga_stream_t *S);
void mga_idx_stream_close(mg_idx_t *gi);
mga_stream_t *mga_idx_stream_acquire(const mg_idx_t *gi, const mg_mapopt_t *opt, int n_threads);
void mga_idx_stream_release(mga_stream_t *S);
void mga_idx_mf_free(mg_idx_t *gi);

typedef struct { const char *cg, *ds; int32_t cg_len, ds_len, mlen, blen; } mga_chain_text_t; /* cg == NULL: format from the chain itself */
#define MGA_KS_WINDOW 0xffffffffu /* kstring_t::m of a window into another buffer: never reallocated, never NUL-terminated (gaf.c) */
void mga_gaf_window_limit(size_t bytes); /* the calling thread's next window holds this many bytes: a write past it aborts before it happens (gaf.c) */
void mga_write_gaf_append(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag,
						  const mga_chain_text_t *txt);
int mga_gaf_chain_rev(const gfa_t *g, const mg_gchains_t *gs, const mg_gchain_t *p, uint64_t flag);
mg_idx_t *mga_idx_hostpart(gfa_t *g, const mg_idxopt_t *io);

/* ---- simple parallel-for over [0,n) on n_threads pthreads, dynamic chunks (par.c) ---- */
typedef void (*mga_for_f)(void *data, int64_t i, int tid);
void mga_parallel_for(int n_threads, int64_t n, mga_for_f f, void *data);

#endif
