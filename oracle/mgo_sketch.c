/*
 * ORACLE restatement of mg_sketch() (sketch.c:56-109) and hash64() (sketch.c:28-38).
 *
 * The reference is a ring-buffer state machine.  It is restated here in the "event timeline" form
 * the HIP kernel uses, so that the derivation itself is pinned against the reference:
 *
 *  events   every base that is NOT skipped by the symmetric-k-mer 'continue' (sketch.c:76) is one
 *           event t: ambiguous bases are events with info = (MAX,MAX) and reset the run length l
 *           (sketch.c:82); other events carry l = number of events since the last ambiguous base
 *           and, once l >= k, info.x = hash64(min(fwd,rev))<<8 | k, info.y = rid<<32|pos<<1|strand
 *           (sketch.c:77-81; kmer_span == k whenever l >= k).
 *  invariant  after event t the reference's `min` is the RIGHTMOST minimum of info.x over the last
 *           w events (virtual events before t=0 are MAX): '<=' at sketch.c:89 and '>=' at :95-98.
 *  emission (P = minimum before event t, N = minimum after it), in this order:
 *     E0  l(t)==w+k-1 && P!=MAX : events in (t-w, t-1] with x==x[P], other than P   sketch.c:84-88
 *     E1  x[t]<=x[P]            : P if l(t)>=w+k && P!=MAX                          sketch.c:89-91
 *     E2  else if P==t-w        : P if l(t)>=w+k-1; then, if l(t)>=w+k-1 && N!=MAX,
 *                                 events in (t-w, t] with x==x[N], other than N      sketch.c:92-104
 *     end the final minimum if != MAX                                                sketch.c:107-108
 */
#include <stdlib.h>
#include <string.h>
#include "mgo.h"

static const uint64_t MGO_MAX = ~(uint64_t)0;

static inline int nt4(char ch) /* seq_nt4_table (sketch.c:9-26): ACGT/acgt -> 0..3, U/u -> 3, else 4 */
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

uint64_t mgo_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

typedef struct { uint64_t x, y; int32_t l; } ev_t;

/* rightmost minimum of x over events (hi-w, hi], virtual events (index < 0) count as MAX */
static int64_t window_min(const ev_t *e, int64_t hi, int32_t w)
{
	int64_t j, lo = hi - w + 1, best = -1;
	uint64_t bx = MGO_MAX;
	if (lo < 0) { lo = 0; best = -1; } /* a virtual MAX event is older than every real one */
	for (j = lo; j <= hi; ++j)
		if (e[j].x <= bx) bx = e[j].x, best = j;
	return best; /* -1 only if hi < 0 */
}

int64_t mgo_sketch(const char *seq, int32_t len, int32_t w, int32_t k, uint32_t rid, mgo128_t *out, int64_t cap)
{
	const uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (k - 1);
	uint64_t fwd = 0, rev = 0;
	int64_t T = 0, t, n_out = 0, j;
	int32_t i, l = 0;
	ev_t *e = (ev_t*)malloc((size_t)(len > 0 ? len : 1) * sizeof(ev_t));

#define EMIT(ev) do { if (n_out < cap) out[n_out].x = (ev).x, out[n_out].y = (ev).y; ++n_out; } while (0)

	/* pass 1: the event timeline */
	for (i = 0; i < len; ++i) {
		int c = nt4(seq[i]);
		ev_t *p;
		if (c < 4) {
			fwd = (fwd << 2 | c) & mask;
			rev = rev >> 2 | (uint64_t)(3 ^ c) << shift1;
			if (fwd == rev) continue; /* strand unknown: the base leaves no event at all */
			++l;
			p = &e[T++];
			p->l = l, p->x = p->y = MGO_MAX;
			if (l >= k) {
				int z = fwd < rev ? 0 : 1;
				p->x = mgo_hash64(z ? rev : fwd, mask) << 8 | (uint64_t)k;
				p->y = (uint64_t)rid << 32 | (uint32_t)i << 1 | z;
			}
		} else {
			l = 0;
			p = &e[T++];
			p->l = 0, p->x = p->y = MGO_MAX;
		}
	}
	/* pass 2: emissions */
	for (t = 0; t < T; ++t) {
		int64_t P = window_min(e, t - 1, w); /* -1: virtual MAX */
		uint64_t px = P < 0 ? MGO_MAX : e[P].x;
		if (e[t].l == w + k - 1 && px != MGO_MAX) { /* E0 */
			for (j = t - w + 1 < 0 ? 0 : t - w + 1; j <= t - 1; ++j)
				if (e[j].x == px && j != P) EMIT(e[j]);
		}
		if (e[t].x <= px) { /* E1 */
			if (e[t].l >= w + k && px != MGO_MAX) EMIT(e[P]);
		} else if (P == t - w) { /* E2 (P >= 0 here because e[t].x > px implies px != MAX) */
			int64_t N;
			if (e[t].l >= w + k - 1) EMIT(e[P]);
			N = window_min(e, t, w);
			if (e[t].l >= w + k - 1 && e[N].x != MGO_MAX)
				for (j = t - w + 1 < 0 ? 0 : t - w + 1; j <= t; ++j)
					if (e[j].x == e[N].x && j != N) EMIT(e[j]);
		}
	}
	if (T > 0) {
		int64_t N = window_min(e, T - 1, w);
		if (e[N].x != MGO_MAX) EMIT(e[N]);
	}
	free(e);
	return n_out <= cap ? n_out : -n_out;
}
