/*
 * align.c -- host half of base alignment: turning a graph chain into gap-filling problems for the
 * WFA kernel, stitching the returned CIGARs, and the ds:Z difference string.
 *
 * Reference: mg_gchain_cigar (galign.c:39-145) calls mwf_wfa_auto() once per pair of consecutive kept
 * anchors; here that loop is split around the GPU:
 *   mga_plan_cigar()   walks the anchors exactly like the reference, emits either a ready-made
 *                      operator (pure match / pure insertion / pure deletion shortcuts, galign.c:98-100)
 *                      or a WFA problem whose target is spliced from the oriented vertex sequences
 *                      (galign.c:66-93) into a per-thread byte pool;
 *   [k_wfa.hip solves every problem of the batch]
 *   mga_apply_cigar()  replays the plan with the reference's run-merging rules (append_cigar1 /
 *                      append_cigar, galign.c:11-37) and fills mg_cigar_t (galign.c:127-140).
 * mga_gen_ds() is mg_gchain_gen_ds (galign.c:147-293).
 */
#include <stdio.h>
#include <assert.h>
#include "hchain.h"
#include "align.h"

static inline void pool_item(mga_tpool_t *tp, int32_t op, int32_t val)
{
	MGA_GROW(mga_cigitem_t, tp->item, tp->n_item, tp->m_item);
	tp->item[tp->n_item].op = op, tp->item[tp->n_item].val = val, ++tp->n_item;
}

static char *pool_target(mga_tpool_t *tp, int64_t len)
{
	if (tp->n_t + len + 16 > tp->m_t) { tp->m_t = (tp->n_t + len + 16) * 3 / 2 + 4096; tp->tseq = (char*)realloc(tp->tseq, (size_t)tp->m_t); }
	return tp->tseq + tp->n_t;
}

void mga_plan_cigar(const gfa_t *g, const gfa_edseq_t *es, const mg_gchains_t *gt, int32_t gc_idx, int64_t q_base, mga_tpool_t *tp, int64_t vert_beg)
{
	const mg_gchain_t *gc = &gt->gc[gc_idx];
	int32_t l0 = gc->off, off_a0 = gt->lc[l0].off, j, j0 = 0, k, l;
	pool_item(tp, 7, (int32_t)(gt->a[off_a0].y >> 32 & 0xff));
	for (j = 1; j < gc->n_anchor; ++j) {
		const mg128_t *q, *p = &gt->a[off_a0 + j];
		int32_t l_seq, qlen;
		if ((p->y & MG_SEED_IGNORE) && j != gc->n_anchor - 1) continue;
		q = &gt->a[off_a0 + j0];
		for (l = l0; l < gc->off + gc->cnt; ++l) { /* the vertex holding anchor j */
			const mg_llchain_t *r = &gt->lc[l];
			if (off_a0 + j >= r->off && off_a0 + j < r->off + r->cnt) break;
		}
		assert(l < gc->off + gc->cnt);
		if (l == l0) l_seq = (int32_t)p->x - (int32_t)q->x;
		else {
			l_seq = g->seg[gt->lc[l0].v>>1].len - (int32_t)q->x - 1;
			for (k = l0 + 1; k < l; ++k) l_seq += es[gt->lc[k].v].len;
			l_seq += (int32_t)p->x + 1;
		}
		qlen = (int32_t)p->y - (int32_t)q->y;
		assert(l_seq > 0 || qlen > 0);
		if (l_seq == 0) pool_item(tp, 1, qlen);
		else if (qlen == 0) pool_item(tp, 2, l_seq);
		else if (l_seq == qlen && qlen <= (int32_t)(q->y >> 32 & 0xff)) pool_item(tp, 7, qlen);
		else { /* a gap for the WFA kernel: target spliced across vertices, query = read[q.y+1 .. p.y] */
			char *seq = tp->want_src ? 0 : pool_target(tp, l_seq);
			mga_wfa_prob_t *pb;
			if (tp->want_src) { /* the device splices the target from its own segment images: say where it lies on the chain's walk */
				mga_plan_src_t *sr;
				if (tp->n_prob == tp->m_src) { tp->m_src = tp->m_src ? tp->m_src + (tp->m_src >> 1) : 1024; tp->src = (mga_plan_src_t*)realloc(tp->src, (size_t)tp->m_src * sizeof(mga_plan_src_t));
					if (tp->src == 0) { fprintf(stderr, "[E::%s] out of memory (%ld gap descriptors)\n", __func__, (long)tp->m_src); abort(); } }
				sr = &tp->src[tp->n_prob];
				sr->lc0 = vert_beg + (l0 - gc->off), sr->n_lc = l - l0, sr->x0 = (int32_t)q->x, sr->x1 = (int32_t)p->x, sr->pad = 0;
			} else if (l == l0) memcpy(seq, &es[gt->lc[l0].v].seq[(int32_t)q->x + 1], (size_t)l_seq);
			else {
				uint32_t v = gt->lc[l0].v;
				int32_t n = g->seg[v>>1].len - (int32_t)q->x - 1;
				memcpy(seq, &es[v].seq[(int32_t)q->x + 1], (size_t)n);
				for (k = l0 + 1; k < l; ++k) {
					v = gt->lc[k].v;
					memcpy(&seq[n], es[v].seq, (size_t)es[v].len);
					n += es[v].len;
				}
				memcpy(&seq[n], es[gt->lc[l].v].seq, (size_t)((int32_t)p->x + 1));
			}
			MGA_GROW(mga_wfa_prob_t, tp->prob, tp->n_prob, tp->m_prob);
			pb = &tp->prob[tp->n_prob];
			pb->t_off = tp->n_t, pb->tl = l_seq;
			pb->q_off = q_base + (int32_t)q->y + 1, pb->ql = qlen;
			tp->n_t += l_seq;
			tp->wfa_t_bases += l_seq, tp->wfa_q_bases += qlen;
			pool_item(tp, -1, (int32_t)tp->n_prob++);
		}
		j0 = j, l0 = l;
	}
}

int mga_apply_cigar(mg_gchains_t *gt, int32_t gc_idx, const mga_cigitem_t *item, int64_t n_item, int64_t prob_base, const mga_cigsrc_t *src)
{
	mg_gchain_t *gc = &gt->gc[gc_idx];
	int64_t t, cap = 0;
	int32_t j, l, n = 0, off_a0 = gt->lc[gc->off].off;
	uint64_t *c;
	/* upper bound on the operator count, then the CIGAR is built in place inside the mg_cigar_t */
	for (t = 0; t < n_item; ++t) {
		if (item[t].op >= 0) ++cap;
		else if (src->ord) cap += src->ncig[prob_base + item[t].val];
		else {
			const mga_wfa_res_t *r = &src->res[prob_base + item[t].val];
			if (r->status != MGA_WFA_OK) return -1;
			cap += r->n_cigar;
		}
	}
	gc->p = (mg_cigar_t*)malloc((size_t)cap * 8 + sizeof(mg_cigar_t));
	memset(gc->p, 0, sizeof(mg_cigar_t)); /* the operators are written below, no need to zero them */
	c = gc->p->cigar;
#define PUSH1(op_, len_) do { /* append_cigar1, galign.c:11-23 */ \
		if (n > 0 && (int32_t)(c[n - 1] & 0xf) == (op_)) c[n - 1] += (uint64_t)(len_) << 4; \
		else c[n++] = (uint64_t)(len_) << 4 | (uint64_t)(op_); \
	} while (0)
	for (t = 0; t < n_item; ++t) {
		if (item[t].op >= 0) PUSH1(item[t].op, item[t].val);
		else {
			const int64_t pj = prob_base + item[t].val;
			const uint32_t *cg;
			int32_t k, nc;
			if (src->ord) nc = src->ncig[pj], cg = src->ord + src->off[pj];
			else nc = src->res[pj].n_cigar, cg = src->pool + src->res[pj].cig_off;
			if (nc == 0) continue;
			PUSH1((int32_t)(cg[0] & 0xf), (int32_t)(cg[0] >> 4)); /* append_cigar, galign.c:25-37: only the first operator can merge */
			for (k = 1; k < nc; ++k) c[n++] = cg[k];
		}
	}
#undef PUSH1
	gc->p->ss = (int32_t)gt->a[off_a0].x + 1 - (int32_t)(gt->a[off_a0].y >> 32 & 0xff);
	gc->p->ee = (int32_t)gt->a[off_a0 + gc->n_anchor - 1].x + 1;
	gc->p->n_cigar = n;
	for (j = 0, l = 0; j < n; ++j) {
		int32_t op = (int32_t)(c[j] & 0xf), len = (int32_t)(c[j] >> 4);
		if (op == 7) gc->p->mlen += len, gc->p->blen += len;
		else gc->p->blen += len;
		if (op != 1) gc->p->aplen += len;
		if (op != 2) l += len;
	}
	memset(&gc->ds, 0, sizeof gc->ds);
	if (!(l == gc->qe - gc->qs && gc->p->aplen == gc->pe - gc->ps)) {
		fprintf(stderr, "[E::%s] CIGAR inconsistent with chain coordinates (galign.c:140): q %d vs %d, path %d vs %d\n", __func__, l, gc->qe - gc->qs, gc->p->aplen, gc->pe - gc->ps);
		return -2;
	}
	return 0;
}

/* ---- ds:Z ---- */
/* lower-case base of a (possibly ambiguous) sequence character, as the reference prints it */
#define NT_LC(ch) ("acgtn"[mga_nt4_table[(uint8_t)(ch)]])

static inline char *ds_put_int(char *w, int32_t x)
{
	char buf[12];
	int l = 0;
	if (x < 10) { *w++ = (char)('0' + x); return w; }
	do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
	while (l > 0) *w++ = buf[--l];
	return w;
}

static inline char *ds_put_indel(char *w, int64_t len, const char *seq, int64_t ll, int64_t lr) /* write_indel, galign.c:153-180 */
{
	int64_t i;
	if (ll + lr >= len) {
		*w++ = '[';
		for (i = 0; i < len; ++i) *w++ = NT_LC(seq[i]);
		*w++ = ']';
	} else {
		int64_t k = 0;
		if (ll > 0) {
			*w++ = '[';
			for (i = 0; i < ll; ++i) *w++ = NT_LC(seq[k + i]);
			*w++ = ']';
			k += ll;
		}
		for (i = 0; i < len - lr - ll; ++i) *w++ = NT_LC(seq[k + i]);
		k += len - lr - ll;
		if (lr > 0) {
			*w++ = '[';
			for (i = 0; i < lr; ++i) *w++ = NT_LC(seq[k + i]);
			*w++ = ']';
		}
	}
	return w;
}

/* mg_gchain_gen_ds (galign.c:182-293).  One sizing pass over the CIGAR (no base compares: an '=' run is one entry, an
 * X/M run of length l at most 1 + 2l entries and 3l + 11 characters, an indel one entry and l + 5 characters), then the
 * string is written with raw stores into a buffer that is handed to the chain as is. */
void mga_gen_ds(const gfa_edseq_t *es, const char *qseq, mg_gchains_t *gt)
{
	int32_t i;
	char *seq = 0;
	int64_t m_seq = 0;
	for (i = 0; i < gt->n_gc; ++i) {
		mg_gchain_t *gc = &gt->gc[i];
		int32_t j, n_off = 0, *off;
		int64_t x, y, l_seq = 0, cap_off = 0, cap_len = 1;
		char *ds, *w;
		const mg_cigar_t *p = gc->p;
		if (p == 0) continue;
		if (p->aplen + 1 > m_seq) { m_seq = p->aplen + 1 + (p->aplen >> 2); free(seq); seq = (char*)malloc((size_t)m_seq); }
		for (j = 0; j < gc->cnt; ++j) { /* the aligned stretch of the walk */
			uint32_t v = gt->lc[gc->off + j].v;
			int32_t st = j > 0 ? 0 : p->ss, en = j < gc->cnt - 1 ? es[v].len : p->ee;
			memcpy(&seq[l_seq], &es[v].seq[st], (size_t)(en - st));
			l_seq += en - st;
		}
		assert(l_seq == p->aplen);
		for (j = 0; j < p->n_cigar; ++j) {
			const int64_t op = p->cigar[j] & 0xf, len = (int64_t)(p->cigar[j] >> 4);
			if (op == 7) cap_off += 1, cap_len += 11;
			else if (op == 0 || op == 8) cap_off += 1 + 2 * len, cap_len += 3 * len + 11 * (len + 1);
			else cap_off += 1, cap_len += len + 5;
		}
		off = MGA_MALLOC(int32_t, cap_off > 0 ? cap_off : 1);
		ds = w = (char*)malloc((size_t)cap_len);
		for (j = 0, x = 0, y = gc->qs; j < p->n_cigar; ++j) {
			const int64_t op = p->cigar[j] & 0xf, len = (int64_t)(p->cigar[j] >> 4);
			if (op == 7) { /* the reference's base-by-base loop (galign.c:228-243) sees len equal codes: one ":len" entry */
				if (len > 0) { off[n_off++] = (int32_t)(w - ds); *w++ = ':'; w = ds_put_int(w, (int32_t)len); }
				x += len, y += len;
			} else if (op == 0 || op == 8) {
				int64_t zz;
				int32_t l = 0;
				for (zz = 0; zz < len; ++zz) {
					const uint8_t cx = mga_nt4_table[(uint8_t)seq[x + zz]], cy = mga_nt4_table[(uint8_t)qseq[y + zz]];
					if (cx != cy) {
						if (l > 0) { off[n_off++] = (int32_t)(w - ds); *w++ = ':'; w = ds_put_int(w, l); }
						off[n_off++] = (int32_t)(w - ds);
						*w++ = '*'; *w++ = "acgtn"[cx]; *w++ = "acgtn"[cy];
						l = 0;
					} else ++l;
				}
				if (l > 0) { off[n_off++] = (int32_t)(w - ds); *w++ = ':'; w = ds_put_int(w, l); }
				x += len, y += len;
			} else if (op == 1) { /* insertion: micro-homology on either side */
				int64_t zz, ll, lr;
				for (zz = 1; zz <= len; ++zz) if (y - zz < gc->qs || qseq[y + len - zz] != qseq[y - zz]) break;
				lr = zz - 1;
				for (zz = 0; zz < len; ++zz) if (y + len + zz >= gc->qe || qseq[y + len + zz] != qseq[y + zz]) break;
				ll = zz;
				off[n_off++] = (int32_t)(w - ds);
				*w++ = '+';
				w = ds_put_indel(w, len, &qseq[y], ll, lr);
				y += len;
			} else if (op == 2) {
				int64_t zz, ll, lr;
				for (zz = 1; zz <= len; ++zz) if (x - zz < 0 || seq[x + len - zz] != seq[x - zz]) break;
				lr = zz - 1;
				for (zz = 0; zz < len; ++zz) if (x + len + zz >= p->aplen || seq[x + zz] != seq[x + len + zz]) break;
				ll = zz;
				off[n_off++] = (int32_t)(w - ds);
				*w++ = '-';
				w = ds_put_indel(w, len, &seq[x], ll, lr);
				x += len;
			}
		}
		*w = 0;
		assert(w - ds < cap_len && n_off <= cap_off);
		gc->ds.len = (int32_t)(w - ds);
		gc->ds.ds = ds;       /* (the reference callocs len+1 bytes; the spare capacity here is never read) */
		gc->ds.n_off = n_off;
		gc->ds.off = off;
	}
	free(seq);
}
