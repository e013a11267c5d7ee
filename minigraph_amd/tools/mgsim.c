/*
 * mgsim -- deterministic synthetic workload generator for the mapping benchmark.
 *
 *   mgsim -p PREFIX [-G backbone_bp] [-c n_chr] [-H n_hap] [-n n_reads] [-l read_len]
 *         [-e err] [-s seed] [-S read_seed] [-R]
 *
 * writes
 *   PREFIX.lin.fa    backbone "chromosomes" as FASTA (config: linear reference, no graph)
 *   PREFIX.gfa       rGFA bubble graph: stems + per-bubble ref allele (rank 0) and one alt
 *                    allele per extra haplotype (rank h), SN/SO/SR tags, 0M links with SR
 *   PREFIX.reads.fa  reads sampled uniformly from the haplotype walks, 50% reverse-complemented,
 *                    ONT-like errors (total err = 40% sub / 30% ins / 30% del, i.i.d.)
 *
 * Everything derives from one xoshiro256** stream seeded by splitmix64(seed): identical bytes
 * on every machine, no dependence on libc rand() or numpy versions.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>

static uint64_t rs[4];
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix(uint64_t *x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static void rng_seed(uint64_t s) { int i; for (i = 0; i < 4; ++i) rs[i] = splitmix(&s); }
static inline uint64_t rng(void)
{
	uint64_t r = rotl(rs[1] * 5, 7) * 9, t = rs[1] << 17;
	rs[2] ^= rs[0]; rs[3] ^= rs[1]; rs[1] ^= rs[2]; rs[0] ^= rs[3]; rs[2] ^= t; rs[3] = rotl(rs[3], 45);
	return r;
}
static inline uint64_t rng_below(uint64_t n) { return (uint64_t)(((__uint128_t)rng() * n) >> 64); } /* tiny bias irrelevant here */
static inline double rng_unif(void) { return (rng() >> 11) * (1.0 / 9007199254740992.0); }

static void rand_seq(char *s, int64_t n)
{
	int64_t i = 0;
	while (i < n) {
		uint64_t r = rng();
		int k;
		for (k = 0; k < 32 && i < n; ++k, r >>= 2) s[i++] = "ACGT"[r & 3];
	}
}

static void write_fa_seq(FILE *fp, const char *s, int64_t n)
{
	int64_t i;
	for (i = 0; i < n; i += 80) {
		int64_t l = n - i < 80 ? n - i : 80;
		fwrite(s + i, 1, l, fp);
		fputc('\n', fp);
	}
}

typedef struct { char *s; int64_t n, m; } str_t;
static void str_app(str_t *t, const char *s, int64_t n)
{
	if (t->n + n + 1 > t->m) { t->m = (t->n + n + 1) * 3 / 2 + 1024; t->s = (char*)realloc(t->s, t->m); }
	memcpy(t->s + t->n, s, n); t->n += n;
}

static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

int main(int argc, char *argv[])
{
	int64_t G = 2000000, n_reads = 1000, read_len = 10000;
	int n_chr = 1, H = 3, c, h;
	double err = 0.10;
	uint64_t seed = 11, read_seed = 0;
	int reads_only = 0;
	const char *prefix = 0;
	char fn[4096];
	FILE *fgfa, *flin, *frd;
	str_t *hap; /* hap[h*n_chr + c] */
	int64_t seg_id = 0, tot_graph = 0, n_bub = 0;

	while ((c = getopt(argc, argv, "p:G:c:H:n:l:e:s:S:R")) >= 0) {
		if (c == 'p') prefix = optarg;
		else if (c == 'G') G = atoll(optarg);
		else if (c == 'c') n_chr = atoi(optarg);
		else if (c == 'H') H = atoi(optarg);
		else if (c == 'n') n_reads = atoll(optarg);
		else if (c == 'l') read_len = atoll(optarg);
		else if (c == 'e') err = atof(optarg);
		else if (c == 's') seed = strtoull(optarg, 0, 10);
		else if (c == 'S') read_seed = strtoull(optarg, 0, 10);
		else if (c == 'R') reads_only = 1; /* the graph of (-G -c -H -s) exists already: write PREFIX.reads.fa only (more reads / other reads, with -S, against one graph) */
	}
	if (prefix == 0 || H < 1 || n_chr < 1) {
		fprintf(stderr, "Usage: mgsim -p PREFIX [-G bp=2000000] [-c n_chr=1] [-H n_hap=3] [-n n_reads=1000] [-l read_len=10000] [-e err=0.1] [-s seed=11] [-S read_seed] [-R (reads only)]\n");
		return 1;
	}
	rng_seed(seed);
	snprintf(fn, sizeof fn, "%s.gfa", prefix);      fgfa = fopen(reads_only ? "/dev/null" : fn, "w");
	snprintf(fn, sizeof fn, "%s.lin.fa", prefix);   flin = fopen(reads_only ? "/dev/null" : fn, "w");
	snprintf(fn, sizeof fn, "%s.reads.fa", prefix); frd  = fopen(fn, "w");
	if (!fgfa || !flin || !frd) { perror("fopen"); return 1; }
	hap = (str_t*)calloc((size_t)H * n_chr, sizeof(str_t));

	for (c = 0; c < n_chr; ++c) {
		int64_t Lc = G / n_chr, pos = 0;
		char *bb = (char*)malloc(Lc + 1), *alt = (char*)malloc(3001);
		int64_t *hoff = (int64_t*)calloc(H, sizeof(int64_t));
		rand_seq(bb, Lc); bb[Lc] = 0;
		fprintf(flin, ">chr%d\n", c + 1);
		write_fa_seq(flin, bb, Lc);
		while (pos < Lc) {
			int64_t stem = 10000 + (int64_t)rng_below(20000), ref_len, stem_id, ref_id = -1;
			if (pos + stem > Lc || Lc - (pos + stem) < 4000) stem = Lc - pos; /* last stem takes the rest */
			stem_id = seg_id++;
			fprintf(fgfa, "S\ts%ld\t%.*s\tSN:Z:chr%d\tSO:i:%ld\tSR:i:0\n", (long)stem_id + 1, (int)stem, bb + pos, c + 1, (long)pos);
			tot_graph += stem;
			for (h = 0; h < H; ++h) str_app(&hap[h * n_chr + c], bb + pos, stem), hoff[h] += stem;
			pos += stem;
			if (pos >= Lc) break;
			/* bubble */
			ref_len = (int64_t)rng_below(3000);
			if (pos + ref_len > Lc - 2000) ref_len = 0;
			++n_bub;
			if (ref_len > 0) {
				ref_id = seg_id++;
				fprintf(fgfa, "S\ts%ld\t%.*s\tSN:Z:chr%d\tSO:i:%ld\tSR:i:0\n", (long)ref_id + 1, (int)ref_len, bb + pos, c + 1, (long)pos);
				tot_graph += ref_len;
			}
			{
				/* the stem after the bubble will get id = seg_id + (#alt segments); compute alt ids first */
				int64_t alt_id[64], next_stem;
				int64_t alt_len[64];
				int take[64];
				for (h = 1; h < H; ++h) {
					alt_len[h] = 50 + (int64_t)rng_below(2950);
					rand_seq(alt, alt_len[h]);
					alt_id[h] = seg_id++;
					take[h] = (rng() >> 63) & 1;
					fprintf(fgfa, "S\ts%ld\t%.*s\tSN:Z:h%dc%d\tSO:i:%ld\tSR:i:%d\n", (long)alt_id[h] + 1, (int)alt_len[h], alt, h, c + 1, (long)hoff[h], h);
					tot_graph += alt_len[h];
					if (take[h]) str_app(&hap[h * n_chr + c], alt, alt_len[h]), hoff[h] += alt_len[h];
					else str_app(&hap[h * n_chr + c], bb + pos, ref_len), hoff[h] += ref_len;
				}
				str_app(&hap[0 * n_chr + c], bb + pos, ref_len); hoff[0] += ref_len;
				next_stem = seg_id; /* id the next stem will receive */
				if (ref_id >= 0) {
					fprintf(fgfa, "L\ts%ld\t+\ts%ld\t+\t0M\tSR:i:0\n", (long)stem_id + 1, (long)ref_id + 1);
					fprintf(fgfa, "L\ts%ld\t+\ts%ld\t+\t0M\tSR:i:0\n", (long)ref_id + 1, (long)next_stem + 1);
				} else {
					fprintf(fgfa, "L\ts%ld\t+\ts%ld\t+\t0M\tSR:i:0\n", (long)stem_id + 1, (long)next_stem + 1);
				}
				for (h = 1; h < H; ++h) {
					fprintf(fgfa, "L\ts%ld\t+\ts%ld\t+\t0M\tSR:i:%d\n", (long)stem_id + 1, (long)alt_id[h] + 1, h);
					fprintf(fgfa, "L\ts%ld\t+\ts%ld\t+\t0M\tSR:i:%d\n", (long)alt_id[h] + 1, (long)next_stem + 1, h);
				}
			}
			pos += ref_len;
		}
		free(bb); free(alt); free(hoff);
	}
	fclose(fgfa); fclose(flin);

	/* reads (optionally from their own stream, so that ranks share one graph but draw different reads) */
	if (read_seed) rng_seed(read_seed);
	{
		int64_t i, tot_hap = 0, *cum = (int64_t*)calloc((size_t)H * n_chr + 1, sizeof(int64_t));
		char *rd = (char*)malloc(read_len * 2 + 64), *src = (char*)malloc(read_len * 2 + 64);
		double p_sub = err * 0.4, p_ins = err * 0.3, p_del = err * 0.3;
		for (i = 0; i < (int64_t)H * n_chr; ++i) { cum[i] = tot_hap; tot_hap += hap[i].n > read_len ? hap[i].n - read_len : 0; }
		cum[(int64_t)H * n_chr] = tot_hap;
		for (i = 0; i < n_reads; ++i) {
			int64_t x = (int64_t)rng_below(tot_hap), k, j, l = 0, sl;
			int rev = (rng() >> 63) & 1;
			const str_t *hp;
			for (k = 0; k < (int64_t)H * n_chr; ++k) if (x < cum[k + 1]) break;
			hp = &hap[k];
			x -= cum[k];
			sl = read_len + read_len / 5 + 32; /* source window; deletions consume extra template */
			if (x + sl > hp->n) sl = hp->n - x;
			if (!rev) memcpy(src, hp->s + x, sl);
			else for (j = 0; j < sl; ++j) src[j] = comp(hp->s[x + sl - 1 - j]);
			for (j = 0; j < sl && l < read_len; ) {
				double u = rng_unif();
				if (u < p_sub) { char b = "ACGT"[rng() >> 62]; while (b == src[j]) b = "ACGT"[rng() >> 62]; rd[l++] = b; ++j; }
				else if (u < p_sub + p_ins) { rd[l++] = "ACGT"[rng() >> 62]; }
				else if (u < p_sub + p_ins + p_del) { ++j; }
				else rd[l++] = src[j++];
			}
			fprintf(frd, ">r%ld\n", (long)i);
			fwrite(rd, 1, l, frd);
			fputc('\n', frd);
		}
		free(rd); free(src); free(cum);
	}
	fclose(frd);
	fprintf(stderr, "[mgsim] backbone=%ld bp, chr=%d, hap=%d, segments=%ld, bubbles=%ld, graph=%ld bp, reads=%ld x %ld\n",
			(long)G, n_chr, H, (long)seg_id, (long)n_bub, (long)tot_graph, (long)n_reads, (long)read_len);
	return 0;
}
