// dev_common.h -- shared device-side helpers (included by every .hip file). gfx950, wave = 64.
#ifndef MGA_DEV_COMMON_H
#define MGA_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MGA_WAVE 64

#define MGA_HIP_CHECK(call) do { \
		hipError_t e_ = (call); \
		if (e_ != hipSuccess) { \
			mga_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
			return -1; \
		} \
	} while (0)

#ifdef __HIPCC__
__device__ __forceinline__ int mga_lane(void) { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t mga_lanemask_lt(void) { return (1ULL << (threadIdx.x & 63)) - 1ULL; }
__device__ __forceinline__ uint64_t mga_lanemask_le(void) { int l = threadIdx.x & 63; return l == 63 ? ~0ULL : (1ULL << (l + 1)) - 1ULL; }

// wave-level LDS/global visibility point for single-wave workgroups: LDS operations of one wave are
// issued in order, so a compiler-level fence plus the hardware's in-order DS queue is enough.
__device__ __forceinline__ void mga_wave_sync(void)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a WFA problem that gives up goes to the next rung's work list (mga_wfa_retry_t); beyond the list's room only the count moves on (the host then sweeps again)
#define mga_wfa_give_up(rt_, pi_) do { const int k_ = atomicAdd((rt_).cnt, 1); if (k_ < (rt_).cap) (rt_).list[k_] = (pi_); } while (0)

__device__ __forceinline__ int32_t mga_wave_bcast_i32(int32_t v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int64_t mga_wave_bcast_i64(int64_t v, int src) { return __shfl(v, src); }

__device__ __forceinline__ int32_t mga_wave_incl_scan_i32(int32_t v)
{
	const int lane = threadIdx.x & 63;
	for (int d = 1; d < 64; d <<= 1) { int32_t y = __shfl_up(v, d); if (lane >= d) v += y; }
	return v;
}
#endif

#endif
