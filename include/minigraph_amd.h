/*
 * minigraph_amd.h -- C ABI of the MI355X-native seed-chain-align engine.
 *
 * Drop-in surface for the mg_map() hot path of lh3/minigraph (reference @ v0.21-r606).  Every type
 * below is layout-compatible with the reference type of the same name and every function in
 * section 2 has the reference's name, argument meaning and error behaviour, so that code written
 * against minigraph.h links against libminigraph_amd.so unchanged.  Section 3 is ADDITIVE: the
 * batched entry point that replaces the per-read kt_for() at gmap.c:99 / ggen.c:64, and the
 * stage-level entry points (plain pointers + sizes) the parity tests and foreign-language
 * bindings call.  No torch / C++ types cross this boundary.
 *
 * Reference citations are file:line in lh3/minigraph.
 */
#ifndef MINIGRAPH_AMD_H
#define MINIGRAPH_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * 1. Types (ABI mirrors)
 * ---------------------------------------------------------------------------------------------- */

/* mapping flags: minigraph.h:9-31 */
#define MG_M_SPLICE       0x10
#define MG_M_SR           0x20
#define MG_M_FRAG_MODE    0x40
#define MG_M_FRAG_MERGE   0x80
#define MG_M_FOR_ONLY     0x100
#define MG_M_REV_ONLY     0x200
#define MG_M_HEAP_SORT    0x400
#define MG_M_VERTEX_COOR  0x800
#define MG_M_ALL_CHAINS   0x1000
#define MG_M_PRINT_2ND    0x2000
#define MG_M_CAL_COV      0x4000
#define MG_M_RMQ          0x8000
#define MG_M_COPY_COMMENT 0x10000
#define MG_M_INDEPEND_SEG 0x20000
#define MG_M_NO_QUAL      0x40000
#define MG_M_2_IO_THREADS 0x80000
#define MG_M_SHOW_UNMAP   0x100000
#define MG_M_NO_COMP_PATH 0x200000
#define MG_M_NO_DIAG      0x400000
#define MG_M_WRITE_LCHAIN 0x800000
#define MG_M_WRITE_MZ     0x1000000
#define MG_M_SKIP_GCHECK  0x2000000
#define MG_M_CIGAR        0x4000000

/* anchor flag bits in mg128_t::y: mgpriv.h:18-27 */
#define MG_SEED_IGNORE     (1ULL<<41)
#define MG_SEED_TANDEM     (1ULL<<42)
#define MG_SEED_FIXED      (1ULL<<43)
#define MG_SEED_SEG_SHIFT  48
#define MG_SEED_SEG_MASK   (0xffULL<<(MG_SEED_SEG_SHIFT))
#define MG_SEED_OCC_SHIFT  56
#define MG_MAX_SEG         255
#define MG_MAX_SHORT_K     15

/* minigraph.h:41-42.  Universal 16-byte record.
 *   minimizer: x = hash<<8 | span                 y = rid<<32 | lastPos<<1 | strand   (sketch.c:50-52)
 *   anchor:    x = seg<<33 | rev<<32 | rpos       y = occ<<56 | qseg<<48 | flags | span<<32 | qpos
 *   after mg_update_anchors: x = minimizer_index<<32 | rpos                            (lchain.c:424-441) */
typedef struct { uint64_t x, y; } mg128_t;
typedef struct { size_t n, m; mg128_t *a; } mg128_v;

/* ---- graph model: gfa.h:33-106 ---- */
typedef struct {
	uint64_t v_lv;   /* vertex_id<<32 | lv */
	uint32_t w;
	int32_t rank;
	int32_t ov, ow;
	uint64_t link_id:61, strong:1, del:1, comp:1;
} gfa_arc_t;

typedef struct { uint32_t m_aux, l_aux; uint8_t *aux; } gfa_aux_t;

struct gfa_utg_s;
typedef struct {
	int32_t len;
	uint32_t del:16, circ:16;
	int32_t snid, soff, rank;
	char *name, *seq;
	struct gfa_utg_s *utg;
	gfa_aux_t aux;
} gfa_seg_t;

typedef struct { char *name; int32_t min, max, rank; } gfa_sseq_t;

typedef struct {
	uint32_t m_seg, n_seg, max_rank;
	gfa_seg_t *seg;
	void *h_names;
	uint32_t m_sseq, n_sseq;
	gfa_sseq_t *sseq;
	void *h_snames;
	uint64_t m_arc, n_arc;
	gfa_arc_t *arc;
	gfa_aux_t *link_aux;
	uint64_t *idx;
} gfa_t;

typedef struct { const char *seq; int32_t len; } gfa_edseq_t;

#define gfa_n_vtx(g) ((g)->n_seg << 1)
#define gfa_arc_n(g, v) ((uint32_t)(g)->idx[(v)])
#define gfa_arc_a(g, v) (&(g)->arc[(g)->idx[(v)]>>32])

/* ---- options: minigraph.h:46-91 ---- */
typedef struct { int w, k; int bucket_bits; } mg_idxopt_t;

typedef struct {
	uint64_t flag;
	int64_t mini_batch_size;
	int seed;
	int max_qlen;
	int pe_ori;
	int occ_max1, occ_max1_cap;
	float occ_max1_frac;
	int bw, bw_long;
	int rmq_size_cap;
	int rmq_rescue_size;
	float rmq_rescue_ratio;
	int max_gap_pre, max_gap, max_gap_ref, max_frag_len;
	float div;
	float chn_pen_gap, chn_pen_skip;
	int max_lc_skip, max_lc_iter, max_gc_skip;
	int min_lc_cnt, min_lc_score;
	int min_gc_cnt, min_gc_score;
	int gdp_max_ed, lc_max_trim, lc_max_occ;
	float mask_level;
	int sub_diff;
	int best_n;
	float pri_ratio;
	int ref_bonus;
	int64_t cap_kalloc;
	int min_cov_mapq, min_cov_blen;
} mg_mapopt_t;

typedef struct {
	uint64_t flag;
	int algo;
	int min_mapq;
	int min_map_len, min_depth_len;
	int min_var_len, match_pen;
	int ggs_shrink_pen;
	int ggs_min_end_cnt;
	float ggs_min_end_frac;
	float ggs_max_iden, ggs_min_inv_iden;
} mg_ggopt_t;

/* ---- index handle: minigraph.h:93-98 (public fields identical; B is opaque there too) ---- */
typedef struct {
	const gfa_t *g;
	gfa_edseq_t *es;
	int32_t b, w, k, flag, n_seg;
	struct mg_idx_bucket_s *B; /* opaque: here a flat host table + its replica in HBM */
} mg_idx_t;

/* ---- results: minigraph.h:100-146 ---- */
typedef struct {
	int32_t off, cnt:31, inner_pre:1;
	uint32_t v;
	int32_t rs, re, qs, qe;
	int32_t score, dist_pre;
	uint32_t hash_pre;
} mg_lchain_t;

typedef struct { int32_t off, cnt; uint32_t v; int32_t score; int32_t ed; } mg_llchain_t;

typedef struct {
	int32_t n_cigar, mlen, blen, aplen, ss, ee;
	uint64_t cigar[]; /* len<<4 | op ; ops: I=1 D=2 '='=7 X=8 */
} mg_cigar_t;

typedef struct { int32_t len, n_off, *off; char *ds; } mg_ds_t;

typedef struct {
	int32_t id, parent;
	int32_t off, cnt;
	int32_t n_anchor, score;
	int32_t qs, qe;
	int32_t plen, ps, pe;
	int32_t blen, mlen;
	float div;
	uint32_t hash;
	int32_t subsc, n_sub;
	uint32_t mapq:8, flt:1, dummy:23;
	mg_cigar_t *p;
	mg_ds_t ds;
} mg_gchain_t;

typedef struct {
	void *km;
	int32_t n_gc, n_lc, n_a, rep_len;
	mg_gchain_t *gc;
	mg_llchain_t *lc;
	mg128_t *a;
} mg_gchains_t;

typedef struct mg_tbuf_s mg_tbuf_t;

#ifndef KSTRING_T
#define KSTRING_T kstring_t
typedef struct __kstring_t { unsigned l, m; char *s; } kstring_t; /* mgpriv.h:31-37 */
#endif

extern int mg_verbose, mg_dbg_flag;   /* minigraph.h:150 */
extern double mg_realtime0;           /* minigraph.h:151 */

/* ------------------------------------------------------------------------------------------------
 * 2. Reference API (same names and semantics)
 * ---------------------------------------------------------------------------------------------- */

/* minigraph.h:158-160 / options.c:65-134 */
int mg_opt_set(const char *preset, mg_idxopt_t *io, mg_mapopt_t *mo, mg_ggopt_t *go);
int mg_opt_check(const mg_idxopt_t *io, const mg_mapopt_t *mo, const mg_ggopt_t *go);
void mg_opt_update(const mg_idx_t *gi, mg_mapopt_t *mo, mg_ggopt_t *go);

/* minigraph.h:163-164 / index.c:211-230.  Builds the minimizer index of g (upper-cases the segment
 * sequences in place like the reference), uploads graph + index to the current GPU, and applies
 * mg_opt_update() when mo != NULL.  Returns NULL if any link has a non-zero overlap. */
mg_idx_t *mg_index(gfa_t *g, const mg_idxopt_t *io, int n_threads, mg_mapopt_t *mo);
void mg_idx_destroy(mg_idx_t *gi);

/* minigraph.h:167-170 / map-algo.c:14-32,340-502.  mg_map() maps ONE read through the same GPU
 * pipeline as a batch of one.  seq must be upper-case ASCII; the result is heap-allocated
 * (km == NULL) and is released with mg_gchain_free(). */
mg_tbuf_t *mg_tbuf_init(void);
void mg_tbuf_destroy(mg_tbuf_t *b);
mg_gchains_t *mg_map(const mg_idx_t *gi, int qlen, const char *seq, mg_tbuf_t *b, const mg_mapopt_t *opt, const char *qname);
void mg_map_frag(const mg_idx_t *gi, int n_segs, const int *qlens, const char **seqs, mg_gchains_t **gcs, mg_tbuf_t *b, const mg_mapopt_t *opt, const char *qname);
void mg_gchain_free(mg_gchains_t *gs);                         /* mgpriv.h:101 / gchain1.c:522-535 */

/* minigraph.h:173 / gmap.c:186-211: index g, map every file, write GAF to stdout. */
int mg_map_files(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads);

/* mgpriv.h:117 / format.c:121-291 */
void mg_write_gaf(kstring_t *s, const gfa_t *g, const mg_gchains_t *gs, int32_t n_seg, const int32_t *qlens, const char *qname, uint64_t flag, void *km);

void mg_sprintf_lite(kstring_t *s, const char *fmt, ...);     /* mgpriv.h:118 / format.c:75-80: %d %u %s %c, appends */
extern unsigned char seq_nt4_table[256];                      /* sketch.c:9-26 */

/* gfa.h:125-128 / gfa-io.c:294-340: rGFA or FASTA (one segment per record) reader */
gfa_t *gfa_read(const char *fn);
void gfa_destroy(gfa_t *g);

/* ------------------------------------------------------------------------------------------------
 * 3. Additive batched API
 * ---------------------------------------------------------------------------------------------- */

/* The GPU replacement for kt_for(n_threads, worker_for, ...) at gmap.c:99: maps n reads in one call.
 * seqs[i] is upper-case ASCII of length qlens[i]; gcs[i] receives what mg_map() would return for
 * read i (NULL for qlen==0 or qlen>opt->max_qlen, map-algo.c:356-360).  n_threads host threads run
 * the host-side stages.  Returns 0, or <0 on a device error (message on stderr). */
int mg_map_batch(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames,
				 mg_gchains_t **gcs, const mg_mapopt_t *opt, int n_threads);

/* mg_map_files() writing to an arbitrary stream instead of stdout. */
int mg_map_files_fp(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads, FILE *out);

/* same, to a file path (convenience for bindings that cannot pass a FILE*) */
int mga_map_files_to_path(gfa_t *g, int n_fn, const char **fn, const mg_idxopt_t *ipt, const mg_mapopt_t *opt0, int n_threads, const char *out_path);

/* the mapping phase of mg_map_files() alone, against an existing index: GAF (input order) to fp, or (fp == NULL) into one malloc'ed
 * buffer *mem / *mem_len (release with mga_free()).  *t_map (optional): wall seconds from the first byte read to the last byte handed
 * to the sink -- the interval between the reference's mg_opt_update and its last worker_pipeline log line (gmap.c:186-211). */
int mga_map_files_idx(const mg_idx_t *gi, int n_fn, const char **fn, const mg_mapopt_t *opt, int n_threads, FILE *fp, char **mem, int64_t *mem_len, double *t_map);
/* one shard of the same job (one process per GPU, gmap.c:98-100 fanned out over devices): rank shard_rank of shard_world maps a
 * contiguous part of every output SEGMENT -- a plain FASTA file is one segment, cut by byte range at record starts; any other input is
 * parsed by every rank and each mini-batch is a segment, cut by read index.  seg_len[0..n_seg) (malloc'ed) are this rank's bytes of each
 * segment: concatenating, segment by segment, the ranks' parts in rank order gives the single-process output. */
int mga_map_files_shard(const mg_idx_t *gi, int n_fn, const char **fn, const mg_mapopt_t *opt, int n_threads, int shard_rank, int shard_world,
						FILE *fp, char **mem, int64_t *mem_len, int64_t *mem_cap /* in: capacity of a buffer passed in *mem for reuse; out: capacity of *mem */,
						int64_t **seg_len, int *n_seg, double *t_map);
int mga_reads_parse_x(const char *fn, int64_t batch_bases, int n_threads, int64_t *n_reads, int64_t *n_bases, uint64_t *hash);
/* the reads of shard rank/world, exactly as mga_map_files_shard() cuts them, as one-line FASTA in out_path; seg_n[0..n_seg) (malloc'ed) =
 * records per output segment.  No device needed. */
int mga_reads_shard_dump(const char *fn, int64_t batch_bases, int n_threads, int rank, int world, const char *out_path, int64_t **seg_n, int *n_seg);

/* a read set kept resident in HBM across calls (repeated passes over one batch, as the benchmark does) */
typedef struct mga_reads_s mga_reads_t;
mga_reads_t *mga_reads_load(const char *fn, int64_t max_reads);   /* FASTA/FASTQ(.gz) -> host copy + HBM copy; NULL on error */
void mga_reads_free(mga_reads_t *rd);
/* the FASTA/FASTQ(.gz) reader alone (no device): record count, bases, FNV-1a hash over "name\nSEQ\n" after upper-casing and U->T */
int mga_reads_parse(const char *fn, int64_t *n_reads, int64_t *n_bases, uint64_t *hash);
int mga_reads_count(const mga_reads_t *rd);
int64_t mga_reads_bases(const mga_reads_t *rd);
/* mg_map_batch() + mg_write_gaf() for a resident read set; *gaf (NUL-terminated, input order) points into a buffer owned by the
 * index: valid until the next mga_map_reads() on this index or mg_idx_destroy(); do NOT free it */
int mga_map_reads(const mg_idx_t *gi, const mga_reads_t *rd, const mg_mapopt_t *opt, int n_threads, char **gaf, int64_t *gaf_len);
int mga_map_batch_resident(const mg_idx_t *gi, int n, const int *qlens, const char **seqs, const char **qnames, mg_gchains_t **gcs,
						   const mg_mapopt_t *opt, int n_threads, const char *d_seq, const int64_t *q_off);

/* counters of the last mg_map_batch() calls on this index (for the bench's algorithmic-bytes figure) */
typedef struct {
	int64_t n_reads, n_bases, n_mz, n_probe, n_hit, n_anchor_chained;
	int64_t n_wfa, wfa_t_bases, wfa_q_bases, wfa_cells, gaf_bytes;
	double t_sketch, t_seed, t_lchain, t_host_chain, t_wfa, t_host_post, t_gaf; /* seconds, summed over pipeline threads */
	int64_t n_rescue_dev, n_rescue_host; /* long-join rescues done by the chaining kernel / deferred to the host tree */
	int64_t n_gwfa, n_shortk, n_gc_retry, gc_arena_peak; /* graph chaining on the device: GWFA bridges, shortest-walk searches, reads re-run in the large arenas, largest arena use (bytes) */
	int64_t n_wfa_dev_plan;  /* gaps (of n_wfa) listed by the device itself (k_plan.hip) rather than by host threads */
	int64_t n_gc_host;       /* reads whose graph chaining the device gave up on even in the large arenas: chained by the host threads */
} mga_stats_t;
void mga_get_stats(const mg_idx_t *gi, mga_stats_t *st, int reset);

/* ---- ADDITIVE: the graph as one binary image (csrc/image.c).  Written once from a gfa_t (after or before mg_index), then MAPPED instead of parsed:
 * replaces gfa_read (gfa-io.c:294) + gfa_finalize (gfa-base.c:421-430) + gfa_edseq_init (gfa-ed.c:24-42) + the host half of mg_index (index.c:186-230) for
 * repeated runs and for the N ranks of a node.  The minimizer table is rebuilt on the device from the sequence (k, w stay load-time options).
 * The loaded index OWNS its graph (gi->g): mg_idx_destroy() releases it; do not gfa_destroy() it. ---- */
int mga_graph_image_save(const gfa_t *g, const char *path);

/* ---- ADDITIVE: mg_gchains_t across ranks (csrc/gcpack.c; SURVEY 8e, BASELINE configs[4]).  Under -x asm the contigs of a query file are sharded over the ranks of a node;
 * what consumes the mappings -- mg_call_asm (asm-call.c:21), mg_ggsimple (ggsimple.c), mg_cov_asm -- wants all of the file's results, in input order, on one rank (what
 * ggen_map fills r->gcs[] with, ggen.c:39-71).  pack: n results (NULL entries allowed) -> one malloc'ed, pointer-free buffer (*out, release with mga_free()), returns its
 * size or -1; unpack: the objects again, each malloc-owned like mg_map()'s (release every one with mg_gchain_free(), the array with mga_free()); NULL on a corrupt buffer. ---- */
int64_t mga_gchains_pack(int n, mg_gchains_t *const *gcs, void **out);
mg_gchains_t **mga_gchains_unpack(const void *buf, int64_t bytes, int *n);
/* One rank's part of ggen_map (ggen.c:39-71: the kt_for over a query file's sequences at ggen.c:64) when the file's contigs are sharded over `world` ranks, and the
 * destination rank's other half.  mga_ggen_map_shard(): the contiguous shard of the n_seq sequences that belongs to `rank` -- cut by BASES, not by count (a file is a few
 * chromosome-scale contigs of very different lengths); every rank computes the same cut from qlens[] alone: mga_ggen_shard_range() -- is mapped through mg_map_batch() and
 * comes back packed (*packed, *packed_bytes: mga_gchains_pack's buffer, release with mga_free()); returns 0, or -1 with mga_last_error().  The buffers of all ranks travel to the
 * rank that runs what consumes a file's mappings (any transport: minigraph_amd.dist.gather_bytes = RCCL point to point on the GPU box, MPI_Gatherv in an MPI build);
 * mga_ggen_assemble() there turns the `world` buffers, in rank order, into the n_seq objects of the file in INPUT order -- what ggen_map's r->gcs[] holds before
 * mg_call_asm (asm-call.c:21) / mg_ggsimple / mg_cov_asm run -- each released with mg_gchain_free(), the array with mga_free(); NULL when a part is corrupt or the
 * parts do not add up to n_seq. */
void mga_ggen_shard_range(int n_seq, const int *qlens, int rank, int world, int *beg, int *end);
int mga_ggen_map_shard(const mg_idx_t *gi, int n_seq, const int *qlens, const char **seqs, const char **qnames, const mg_mapopt_t *opt, int n_threads, int rank, int world,
					   void **packed, int64_t *packed_bytes);
mg_gchains_t **mga_ggen_assemble(int world, const void *const *parts, const int64_t *part_bytes, int n_seq);
mg_idx_t *mga_index_load_image(const char *path, const mg_idxopt_t *io, int n_threads, mg_mapopt_t *mo);

/* ---- stage-level entry points (host pointers in, host pointers out; device work inside) ----
 * Each replaces the per-read reference routine named in its comment for a whole batch.  Outputs are
 * malloc()'ed by the callee and released with mga_free().  All return 0 on success, <0 on error. */
void mga_free(void *p);
int mga_device_count(void);
const char *mga_last_error(void);

/* mg_sketch (sketch.c:56-109) for n sequences: seq[off[i]..off[i+1]) ASCII, rid[i] (NULL => 0).
 * Out: mz = concatenated minimizers, mz_off[n+1]. */
int mga_sketch_batch(int n, const char *seq, const int64_t *off, const uint32_t *rid, int w, int k,
					 mg128_t **mz, int64_t **mz_off);

/* collect_matches + collect_seed_hits (map-algo.c:58-91,152-192) for n reads against gi's index:
 * in: minimizers of each read (mz, mz_off) ; out per read: x-sorted anchors, rep_len, mini_pos. */
int mga_seed_batch(const mg_idx_t *gi, int n, const mg128_t *mz, const int64_t *mz_off, int max_occ,
				   mg128_t **a, int64_t **a_off, int32_t **rep_len, int32_t **mini_pos, int64_t **mini_off);

/* mg_lchain_dp + mg_chain_backtrack + compact_a (lchain.c:9-219) for n reads.
 * in: x-sorted anchors per read ; out per read: u[] (score<<32|cnt) and the compacted anchors. */
typedef struct {
	int32_t max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc;
	float chn_pen_gap, chn_pen_skip;
} mga_lchain_par_t;
int mga_lchain_batch(int n, const mg128_t *a, const int64_t *a_off, const mga_lchain_par_t *par,
					 uint64_t **u, int64_t **u_off, mg128_t **b, int64_t **b_off);
/* radix_sort_128x (ksort.h:112-162 through misc.c:9) as the chaining kernels run it: array i = a[a_off[i]..a_off[i+1]) is sorted in place by x with the
 * reference's exact permutation of equal keys -- one wavefront per array, the LDS form up to 1024 elements, the in-memory form beyond (tests). */
int mga_sort128x_batch(int n, mg128_t *a, const int64_t *a_off);
/* the dv:f: field of a GAF line as the device prints it (k_gaf.hip; format.c:200-203: "%.4f", "0" for zero): out = n x 8 bytes, NUL-padded (tests) */
int mga_gaf_div_batch(int n, const float *div, char *out);

/* mwf_wfa_auto's exact mode (miniwfa.c:380-435,603-615,824-828) for n independent problems:
 * target i = tseq[t_off[i]..t_off[i+1]), query i = qseq[q_off[i]..q_off[i+1]) (raw ASCII compare).
 * out: score[i] (-1: exceeded max_iter), cigar ops (len<<4|op) concatenated, cig_off[n+1]. */
int mga_wfa_batch(int n, const char *tseq, const int64_t *t_off, const char *qseq, const int64_t *q_off,
				  int32_t **score, uint32_t **cigar, int64_t **cig_off);

#ifdef __cplusplus
}
#endif
#endif
