/* mga_idxhash.h -- slot function of the flat minimizer table, shared by the host builder (index.c)
 * and the device probe (k_seed.hip).  The key is already an invertible mix of the k-mer (hash64,
 * sketch.c:28-38); one multiplicative scramble spreads it over the table. */
#ifndef MGA_IDXHASH_H
#define MGA_IDXHASH_H
#include <stdint.h>
#define MGA_IDX_EMPTY (~0ULL)
#define MGA_IDX_LIST  (1ULL << 63)   /* set in the stored key when the value is off<<32|n into the position array */
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t mga_idx_slot(uint64_t key, int bits)
{
	return ((key ^ key >> 31) * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
}
#endif
