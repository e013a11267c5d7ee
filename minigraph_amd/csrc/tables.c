/* tables.c -- nucleotide tables: nt4 code (sketch.c:9-26) and IUPAC complement (gfa-base.c:509-526),
 * generated from their definitions rather than stored. */
#include <pthread.h>
#include "mga_host.h"

unsigned char mga_comp_table[256];
unsigned char mga_nt4_table[256];
static pthread_once_t g_tables_once = PTHREAD_ONCE_INIT;

static void tables_fill(void)
{
	static const char *pairs = "ATCGBVDHKMRY"; /* complement pairs; S, W, N and every other letter map to themselves; U -> A */
	int i;
	for (i = 0; i < 256; ++i) mga_comp_table[i] = (unsigned char)i, mga_nt4_table[i] = 4;
	for (i = 0; pairs[i]; i += 2) {
		unsigned char a = (unsigned char)pairs[i], b = (unsigned char)pairs[i+1];
		mga_comp_table[a] = b, mga_comp_table[b] = a;
		mga_comp_table[a + 32] = b + 32, mga_comp_table[b + 32] = a + 32;
	}
	mga_comp_table['U'] = 'A', mga_comp_table['u'] = 'a';
	mga_nt4_table['A'] = mga_nt4_table['a'] = 0;
	mga_nt4_table['C'] = mga_nt4_table['c'] = 1;
	mga_nt4_table['G'] = mga_nt4_table['g'] = 2;
	mga_nt4_table['T'] = mga_nt4_table['t'] = mga_nt4_table['U'] = mga_nt4_table['u'] = 3;
	mga_nt4_table[0] = 0, mga_nt4_table[1] = 1, mga_nt4_table[2] = 2, mga_nt4_table[3] = 3; /* sketch.c:10: codes map to themselves */
}

/* sketch.c:9: the reference exports its code table as data (miniwfa.c:648 reads it from other objects); same contents, ready at load time */
unsigned char seq_nt4_table[256];
__attribute__((constructor)) static void tables_export(void) { pthread_once(&g_tables_once, tables_fill); memcpy(seq_nt4_table, mga_nt4_table, 256); }

void mga_tables_init(void) { pthread_once(&g_tables_once, tables_fill); } /* callable from any thread, any number of times */
