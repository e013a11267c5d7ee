/*
 * gcpack.c -- mg_gchains_t across ranks (BASELINE configs[4]): under -x asm the contigs of a query file are sharded over the ranks of a node
 * (one GPU each), and what consumes the mappings -- mg_call_asm (asm-call.c:21), the graph generation of ggsimple.c, mg_cov_asm -- wants ALL of a file's
 * mg_gchains_t in input order on one rank (ggen_map, ggen.c:39-71 fills r->gcs[] for exactly that).  mga_gchains_pack() turns a rank's results into one
 * flat, pointer-free buffer (what a gather moves); mga_gchains_unpack() rebuilds malloc-owned objects the reference's code reads and mg_gchain_free() releases.
 *
 * Layout (little endian, 8-byte aligned records):
 *   uint64 magic, n
 *   per read: int32 present, n_gc, n_lc, n_a, rep_len, pad[3]
 *             gc[n_gc] as mg_gchain_t with the three pointers replaced by byte counts of what follows (0: absent)
 *             lc[n_lc], a[n_a]
 *             per chain: mg_cigar_t header + cigar[n_cigar] (if p), ds.off[n_off] + ds.ds[len + 1] (if ds.ds)
 */
#include "mga_host.h"

#define GCP_MAGIC 0x31504347414d4d47ULL /* "GMMAGCP1" */

typedef struct { int32_t present, n_gc, n_lc, n_a, rep_len, pad[3]; } gcp_read_t;
typedef struct { char *p; size_t n, m; } gcp_buf_t;

static int gcp_put(gcp_buf_t *b, const void *src, size_t bytes)
{
	const size_t pad = (8 - (bytes & 7)) & 7;
	if (b->n + bytes + pad > b->m) {
		size_t m = (b->n + bytes + pad) * 2 + 4096;
		char *q = (char*)realloc(b->p, m);
		if (q == 0) return -1;
		b->p = q, b->m = m;
	}
	if (bytes) memcpy(b->p + b->n, src, bytes);
	if (pad) memset(b->p + b->n + bytes, 0, pad);
	b->n += bytes + pad;
	return 0;
}

int64_t mga_gchains_pack(int n, mg_gchains_t *const *gcs, void **out)
{
	gcp_buf_t b = { 0, 0, 0 };
	uint64_t hdr[2] = { GCP_MAGIC, (uint64_t)(n > 0 ? n : 0) };
	int i, k;
	*out = 0;
	if (gcp_put(&b, hdr, sizeof hdr) < 0) goto fail;
	for (i = 0; i < n; ++i) {
		const mg_gchains_t *gs = gcs[i];
		gcp_read_t r;
		memset(&r, 0, sizeof r);
		if (gs == 0) { if (gcp_put(&b, &r, sizeof r) < 0) goto fail; continue; } /* map-algo.c:356-360: no object for an empty / over-long read */
		r.present = 1, r.n_gc = gs->n_gc, r.n_lc = gs->n_lc, r.n_a = gs->n_a, r.rep_len = gs->rep_len;
		if (gcp_put(&b, &r, sizeof r) < 0) goto fail;
		for (k = 0; k < gs->n_gc; ++k) { /* records with pointer fields turned into sizes */
			mg_gchain_t g = gs->gc[k];
			g.p = (mg_cigar_t*)(uintptr_t)(gs->gc[k].p ? sizeof(mg_cigar_t) + (size_t)gs->gc[k].p->n_cigar * 8 : 0);
			g.ds.off = (int32_t*)(uintptr_t)(gs->gc[k].ds.ds ? (size_t)gs->gc[k].ds.n_off * 4 : 0);
			g.ds.ds = (char*)(uintptr_t)(gs->gc[k].ds.ds ? (size_t)gs->gc[k].ds.len + 1 : 0);
			if (gcp_put(&b, &g, sizeof g) < 0) goto fail;
		}
		if (gcp_put(&b, gs->lc, (size_t)gs->n_lc * sizeof(mg_llchain_t)) < 0 || gcp_put(&b, gs->a, (size_t)gs->n_a * sizeof(mg128_t)) < 0) goto fail;
		for (k = 0; k < gs->n_gc; ++k) {
			const mg_gchain_t *g = &gs->gc[k];
			if (g->p && gcp_put(&b, g->p, sizeof(mg_cigar_t) + (size_t)g->p->n_cigar * 8) < 0) goto fail;
			if (g->ds.ds && (gcp_put(&b, g->ds.off, (size_t)g->ds.n_off * 4) < 0 || gcp_put(&b, g->ds.ds, (size_t)g->ds.len + 1) < 0)) goto fail;
		}
	}
	*out = b.p;
	return (int64_t)b.n;
fail:
	free(b.p);
	mga_set_error("mga_gchains_pack: out of memory");
	return -1;
}

/* cursor over an untrusted buffer: every read is bounds-checked */
typedef struct { const char *p; size_t n, at; } gcp_cur_t;
static const void *gcp_get(gcp_cur_t *c, size_t bytes)
{
	const size_t pad = (8 - (bytes & 7)) & 7;
	const void *r;
	if (bytes > c->n - c->at || pad > c->n - c->at - bytes) return 0;
	r = c->p + c->at;
	c->at += bytes + pad;
	return r;
}

static void *gcp_dup(const void *src, size_t bytes) { void *p = malloc(bytes ? bytes : 1); if (p && bytes) memcpy(p, src, bytes); return p; }

mg_gchains_t **mga_gchains_unpack(const void *buf, int64_t bytes, int *n_out)
{
	gcp_cur_t c = { (const char*)buf, bytes > 0 ? (size_t)bytes : 0, 0 };
	const uint64_t *hdr = (const uint64_t*)gcp_get(&c, 16);
	mg_gchains_t **gcs = 0;
	int64_t n, i;
	int32_t k;
	*n_out = 0;
	if (hdr == 0 || hdr[0] != GCP_MAGIC || hdr[1] > 0x7fffffffULL) { mga_set_error("mga_gchains_unpack: not a packed chain buffer"); return 0; }
	n = (int64_t)hdr[1];
	if ((uint64_t)n > c.n / sizeof(gcp_read_t)) { mga_set_error("mga_gchains_unpack: truncated buffer"); return 0; }
	gcs = (mg_gchains_t**)calloc(n > 0 ? (size_t)n : 1, sizeof(mg_gchains_t*));
	if (gcs == 0) { mga_set_error("mga_gchains_unpack: out of memory"); return 0; }
	for (i = 0; i < n; ++i) {
		const gcp_read_t *r = (const gcp_read_t*)gcp_get(&c, sizeof *r);
		const mg_gchain_t *g;
		const void *lc, *a;
		mg_gchains_t *gs;
		if (r == 0) goto bad;
		if (!r->present) continue;
		if (r->n_gc < 0 || r->n_lc < 0 || r->n_a < 0) goto bad;
		g = (const mg_gchain_t*)gcp_get(&c, (size_t)r->n_gc * sizeof(mg_gchain_t));
		lc = gcp_get(&c, (size_t)r->n_lc * sizeof(mg_llchain_t));
		a = gcp_get(&c, (size_t)r->n_a * sizeof(mg128_t));
		if (g == 0 || lc == 0 || a == 0) goto bad;
		/* the consumers (mg_call_asm, the GAF writer) index lc[] through gc[].off/cnt and a[] through lc[].off/cnt, so a peer buffer is only accepted when
		 * every one of those ranges lies inside its array; an object without chains owns no arrays (gchain1.c:460) and therefore may not claim any records */
		if (r->n_gc == 0 && (r->n_lc != 0 || r->n_a != 0)) goto bad;
		for (k = 0; k < r->n_gc; ++k) if (g[k].off < 0 || g[k].cnt < 0 || (int64_t)g[k].off + g[k].cnt > r->n_lc || g[k].n_anchor < 0) goto bad;
		for (k = 0; k < r->n_lc; ++k) { const mg_llchain_t *l = (const mg_llchain_t*)lc + k; if (l->off < 0 || l->cnt < 0 || (int64_t)l->off + l->cnt > r->n_a) goto bad; }
		gs = gcs[i] = MGA_CALLOC(mg_gchains_t, 1);
		if (gs == 0) goto oom;
		gs->rep_len = r->rep_len;
		if (r->n_gc == 0) continue; /* gchain1.c:460: a valid object without chains owns no arrays */
		gs->gc = (mg_gchain_t*)gcp_dup(g, (size_t)r->n_gc * sizeof(mg_gchain_t));
		if (gs->gc == 0) goto oom;
		gs->n_gc = r->n_gc;
		for (k = 0; k < r->n_gc; ++k) gs->gc[k].p = 0, gs->gc[k].ds.off = 0, gs->gc[k].ds.ds = 0; /* (sizes so far: owned pointers from here on; mg_gchain_free walks gc[0 .. n_gc)) */
		gs->lc = (mg_llchain_t*)gcp_dup(lc, (size_t)r->n_lc * sizeof(mg_llchain_t));
		gs->a = (mg128_t*)gcp_dup(a, (size_t)r->n_a * sizeof(mg128_t));
		if (gs->lc == 0 || gs->a == 0) goto oom;
		gs->n_lc = r->n_lc, gs->n_a = r->n_a;
		for (k = 0; k < r->n_gc; ++k) {
			const size_t pb = (size_t)(uintptr_t)g[k].p, ob = (size_t)(uintptr_t)g[k].ds.off, db = (size_t)(uintptr_t)g[k].ds.ds;
			if (pb) {
				const mg_cigar_t *p = (const mg_cigar_t*)gcp_get(&c, pb);
				if (p == 0 || pb < sizeof(mg_cigar_t) || p->n_cigar < 0 || pb != sizeof(mg_cigar_t) + (size_t)p->n_cigar * 8) goto bad;
				if ((gs->gc[k].p = (mg_cigar_t*)gcp_dup(p, pb)) == 0) goto oom;
			}
			if (db) {
				const void *off = gcp_get(&c, ob), *ds = gcp_get(&c, db);
				if (off == 0 || ds == 0 || g[k].ds.n_off < 0 || g[k].ds.len < 0 || ob != (size_t)g[k].ds.n_off * 4 || db != (size_t)g[k].ds.len + 1) goto bad;
				gs->gc[k].ds.off = (int32_t*)gcp_dup(off, ob);
				gs->gc[k].ds.ds = (char*)gcp_dup(ds, db);
				if (gs->gc[k].ds.off == 0 || gs->gc[k].ds.ds == 0) goto oom;
			} else if (ob) goto bad;
		}
	}
	*n_out = (int)n;
	return gcs;
bad:
	for (i = 0; i < n; ++i) mg_gchain_free(gcs[i]);
	free(gcs);
	mga_set_error("mga_gchains_unpack: truncated or corrupt buffer");
	return 0;
oom:
	for (i = 0; i < n; ++i) mg_gchain_free(gcs[i]);
	free(gcs);
	mga_set_error("mga_gchains_unpack: out of memory");
	return 0;
}

/* ---- ggen_map (ggen.c:39-71) over ranks: see include/minigraph_amd.h ---- */
void mga_ggen_shard_range(int n_seq, const int *qlens, int rank, int world, int *beg, int *end)
{
	int64_t tot = 0, acc = 0;
	int i, b = -1, e = -1;
	if (world < 1) world = 1;
	if (rank < 0) rank = 0;
	if (rank >= world) rank = world - 1;
	for (i = 0; i < n_seq; ++i) tot += qlens[i] > 0 ? qlens[i] : 0;
	/* sequence i belongs to the rank r with tot * r / world <= (bases before i) < tot * (r + 1) / world: contiguous, order preserving, every sequence exactly once; an empty
	 * sequence is placed by the bases BEFORE it like any other, i.e. it goes with the sequence that follows it (at a cut: with the successor's rank) */
	for (i = 0; i < n_seq; ++i) {
		int r = tot > 0 ? (int)((__int128)acc * world / tot) : 0;
		if (r >= world) r = world - 1;
		if (r >= rank && b < 0) b = i;
		if (r > rank && e < 0) e = i;
		acc += qlens[i] > 0 ? qlens[i] : 0;
	}
	if (b < 0) b = n_seq;
	if (e < 0) e = n_seq;
	if (e < b) e = b;
	*beg = b, *end = e;
}

int mga_ggen_map_shard(const mg_idx_t *gi, int n_seq, const int *qlens, const char **seqs, const char **qnames, const mg_mapopt_t *opt, int n_threads, int rank, int world,
					   void **packed, int64_t *packed_bytes)
{
	int beg, end, i, n;
	mg_gchains_t **gcs;
	int64_t nb;
	*packed = 0, *packed_bytes = 0;
	if (world < 1 || rank < 0 || rank >= world) { mga_set_error("mga_ggen_map_shard: rank %d of %d", rank, world); return -1; }
	mga_ggen_shard_range(n_seq, qlens, rank, world, &beg, &end);
	n = end - beg;
	gcs = (mg_gchains_t**)calloc(n > 0 ? (size_t)n : 1, sizeof *gcs);
	if (gcs == 0) { mga_set_error("mga_ggen_map_shard: out of memory"); return -1; }
	if (n > 0 && mg_map_batch(gi, n, qlens + beg, seqs + beg, qnames ? qnames + beg : 0, gcs, opt, n_threads) < 0) { /* (the message is mg_map_batch's) */
		for (i = 0; i < n; ++i) mg_gchain_free(gcs[i]);
		free(gcs);
		return -1;
	}
	nb = mga_gchains_pack(n, gcs, packed);
	for (i = 0; i < n; ++i) mg_gchain_free(gcs[i]);
	free(gcs);
	if (nb < 0) return -1;
	*packed_bytes = nb;
	return 0;
}

mg_gchains_t **mga_ggen_assemble(int world, const void *const *parts, const int64_t *part_bytes, int n_seq)
{
	mg_gchains_t **all = (mg_gchains_t**)calloc(n_seq > 0 ? (size_t)n_seq : 1, sizeof *all);
	int r, at = 0, i;
	if (all == 0) { mga_set_error("mga_ggen_assemble: out of memory"); return 0; }
	for (r = 0; r < world; ++r) {
		int k = 0;
		mg_gchains_t **part = mga_gchains_unpack(parts[r], part_bytes[r], &k);
		if (part == 0) goto bad; /* (mga_gchains_unpack set the message) */
		if (at + k > n_seq) { for (i = 0; i < k; ++i) mg_gchain_free(part[i]); free(part); mga_set_error("mga_ggen_assemble: the parts hold more than %d sequences", n_seq); goto bad; }
		for (i = 0; i < k; ++i) all[at++] = part[i];
		free(part);
	}
	if (at != n_seq) { mga_set_error("mga_ggen_assemble: the parts hold %d sequences, the file has %d", at, n_seq); goto bad; }
	return all;
bad:
	for (i = 0; i < at; ++i) mg_gchain_free(all[i]);
	free(all);
	return 0;
}
