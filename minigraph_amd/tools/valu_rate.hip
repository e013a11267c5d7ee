// valu_rate.hip -- measurement aid: how fast does ONE SIMD of gfx950 issue the integer vector instructions the WFA step is made of?
// Prints cycles per wave64 instruction for chains of independent v_max_i32 / v_add_u32 / v_cndmask / DPP moves, with 1..8 waves per SIMD.
// (The answer prices the "valu-issue" roofline of bench.py: DESIGN.md, Measurement notes.)   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template<int KIND>
__global__ void __launch_bounds__(64) k_rate(int iters, int32_t *out, long long *cyc)
{
	int32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const int32_t b = out[0];
	// KIND 5 / 6 / 7: the v_max_i32 stream with only lanes 0..31 / 0..15 / the even lanes active -- does a wave64 instruction whose EXEC half (quarter) is empty take fewer issue cycles?
	if (KIND == 5 && threadIdx.x >= 32) return;
	if (KIND == 6 && threadIdx.x >= 16) return;
	if (KIND == 7 && (threadIdx.x & 1)) return;
	int32_t sacc = 0;
	const long long t0 = clock64();
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) { // 8 independent chains x 8 = 64 instructions per trip
			if (KIND == 0) { a0 = max(a0, b); a1 = max(a1, b); a2 = max(a2, b); a3 = max(a3, b); a4 = max(a4, b); a5 = max(a5, b); a6 = max(a6, b); a7 = max(a7, b); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
			if (KIND == 1) { a0 += b; a1 += b; a2 += b; a3 += b; a4 += b; a5 += b; a6 += b; a7 += b; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
			if (KIND == 2) {
				a0 = __builtin_amdgcn_update_dpp(a0, a1, 0x138, 0xf, 0xf, false); a1 = __builtin_amdgcn_update_dpp(a1, a2, 0x138, 0xf, 0xf, false);
				a2 = __builtin_amdgcn_update_dpp(a2, a3, 0x138, 0xf, 0xf, false); a3 = __builtin_amdgcn_update_dpp(a3, a4, 0x138, 0xf, 0xf, false);
				a4 = __builtin_amdgcn_update_dpp(a4, a5, 0x111, 0xf, 0xf, false); a5 = __builtin_amdgcn_update_dpp(a5, a6, 0x111, 0xf, 0xf, false);
				a6 = __builtin_amdgcn_update_dpp(a6, a7, 0x111, 0xf, 0xf, false); a7 = __builtin_amdgcn_update_dpp(a7, a0, 0x111, 0xf, 0xf, false);
			}
			if (KIND == 5 || KIND == 6 || KIND == 7) { a0 = max(a0, b); a1 = max(a1, b); a2 = max(a2, b); a3 = max(a3, b); a4 = max(a4, b); a5 = max(a5, b); a6 = max(a6, b); a7 = max(a7, b); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
			if (KIND == 4) { // packed 16-bit maxima and sums (the packed WFA step)
				typedef short pk2 __attribute__((ext_vector_type(2)));
#define PKMAX(x_) x_ = __builtin_bit_cast(int32_t, __builtin_elementwise_max(__builtin_bit_cast(pk2, x_), __builtin_bit_cast(pk2, b)))
#define PKADD(x_) x_ = __builtin_bit_cast(int32_t, (pk2)(__builtin_bit_cast(pk2, x_) + __builtin_bit_cast(pk2, b)))
				PKMAX(a0); PKADD(a1); PKMAX(a2); PKADD(a3); PKMAX(a4); PKADD(a5); PKMAX(a6); PKADD(a7);
				asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
			}
			if (KIND == 8) { // v_alignbit_b32 (three operands) and v_and_or_b32
				a0 = __builtin_amdgcn_alignbit(a0, a1, 16); a1 = (a1 & b) | a2; a2 = __builtin_amdgcn_alignbit(a2, a3, 16); a3 = (a3 & b) | a4;
				a4 = __builtin_amdgcn_alignbit(a4, a5, 16); a5 = (a5 & b) | a6; a6 = __builtin_amdgcn_alignbit(a6, a7, 16); a7 = (a7 & b) | a0;
			}
			if (KIND == 9) { // the match-mask block of the windowed WFA kernels: SDWA byte compares into VCC, the carry shifted into a register (8 instructions)
				asm volatile("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
							 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:BYTE_2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
							 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:BYTE_1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
							 "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
							 : "+v"(a0) : "v"(a1), "v"(b) : "vcc");
			}
			if (KIND == 10) { // eight v_max_i32 issued with EXEC = 0 (what a skipped per-lane branch leaves in the instruction stream)
				uint64_t saved;
				asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0\n\t"
							 "v_max_i32 %1, %1, %9\n\tv_max_i32 %2, %2, %9\n\tv_max_i32 %3, %3, %9\n\tv_max_i32 %4, %4, %9\n\t"
							 "v_max_i32 %5, %5, %9\n\tv_max_i32 %6, %6, %9\n\tv_max_i32 %7, %7, %9\n\tv_max_i32 %8, %8, %9\n\t"
							 "s_mov_b64 exec, %0"
							 : "=&s"(saved), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
			}
			if (KIND == 11) { // v_readfirstlane_b32 (a vector instruction that reads one lane)
				int32_t s0, s1, s2, s3, s4, s5, s6, s7;
				asm volatile("v_readfirstlane_b32 %0, %8\n\tv_readfirstlane_b32 %1, %9\n\tv_readfirstlane_b32 %2, %10\n\tv_readfirstlane_b32 %3, %11\n\t"
							 "v_readfirstlane_b32 %4, %12\n\tv_readfirstlane_b32 %5, %13\n\tv_readfirstlane_b32 %6, %14\n\tv_readfirstlane_b32 %7, %15"
							 : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
				sacc += s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
			}
			if (KIND == 3) { // compare + select pairs
				a0 = a0 < b ? a1 : a0; a1 = a1 < b ? a2 : a1; a2 = a2 < b ? a3 : a2; a3 = a3 < b ? a4 : a3;
				asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
			}
		}
	}
	const long long t1 = clock64();
	out[1 + blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sacc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; // (lane 0 is active in every variant)
}

// A persistent kernel with a global work queue, as the WFA rungs are: 8 192 workgroups of which a fifth is resident (24 KB of LDS each), n_items items of ITEM_TRIPS x 64
// v_max_i32 drawn one at a time.  The instruction total is known exactly (items x (ITEM_TRIPS x 64 + a few)); how it spreads over the waves, CUs and XCDs is up to the dispatcher.
// Question: do the per-dispatch SQ counters report that total, or a scaled sample of part of the chip?
#define ITEM_TRIPS 100
__global__ void __launch_bounds__(64) k_queue(int n_items, int *counter, int32_t *out, unsigned long long *done)
{
	__shared__ int32_t pad[6144];
	int32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const int32_t b = out[0];
	pad[threadIdx.x] = b;
	unsigned long long mine = 0;
	for (;;) {
		int item = 0;
		if (threadIdx.x == 0) item = atomicAdd(counter, 1);
		item = __builtin_amdgcn_readfirstlane(item);
		if (item >= n_items) break;
		for (int i = 0; i < ITEM_TRIPS; ++i) {
#pragma unroll
			for (int u = 0; u < 8; ++u) { a0 = max(a0, b); a1 = max(a1, b); a2 = max(a2, b); a3 = max(a3, b); a4 = max(a4, b); a5 = max(a5, b); a6 = max(a6, b); a7 = max(a7, b); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
		}
		++mine;
	}
	out[1 + blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + pad[(threadIdx.x * 7) & 63];
	if (threadIdx.x == 0) atomicAdd(done, mine);
}

static void run_queue()
{
	int32_t *out; int *counter; unsigned long long *done;
	const int nb = 8192, n_items = 400000;
	hipMalloc(&out, 4 * (1 + 64 * nb)); hipMalloc(&counter, 4); hipMalloc(&done, 8); hipMemset(out, 0, 4);
	for (int rep = 0; rep < 2; ++rep) {
		hipMemset(counter, 0, 4); hipMemset(done, 0, 8);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_queue, dim3(nb), dim3(64), 0, 0, n_items, counter, out, done);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		unsigned long long h = 0; hipMemcpy(&h, done, 8, hipMemcpyDeviceToHost);
		printf("k_queue: %d workgroups, %d items of %d x 64 v_max_i32: %llu items done, %.3f ms -> %.1f G wave-instructions/s on the chip = %.0f M/s per SIMD (1024 SIMDs); expected VALU per wave (average over %d waves) = %.1f + a few per item\n",
			   nb, n_items, ITEM_TRIPS, h, ms, (double)h * ITEM_TRIPS * 64 / (ms * 1e6), (double)h * ITEM_TRIPS * 64 / (ms * 1e3) / 1024, nb, (double)h * ITEM_TRIPS * 64 / nb);
	}
	hipFree(out); hipFree(counter); hipFree(done);
}

template<int KIND> static void run(const char *name, int per_trip)
{
	int32_t *out; long long *cyc;
	const int iters = 20000;
	hipMalloc(&out, 4 * (1 + 64 * 8192)); hipMalloc(&cyc, 8 * 8192); hipMemset(out, 0, 4);
	for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: 256 CUs x 4 SIMDs x wps single-wave workgroups
		const int nb = 256 * 4 * wps;
		hipLaunchKernelGGL(k_rate<KIND>, dim3(nb), dim3(64), 0, 0, 1000, out, cyc); hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_rate<KIND>, dim3(nb), dim3(64), 0, 0, iters, out, cyc);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		long long h[8192]; hipMemcpy(h, cyc, 8 * nb, hipMemcpyDeviceToHost);
		double mean = 0; for (int i = 0; i < nb; ++i) mean += h[i]; mean /= nb;
		const double n_instr = (double)iters * per_trip;
		printf("%-28s %d waves/SIMD: %.2f clock64 ticks per instruction per wave (%.1f MHz tick), wall %.3f ms -> %.2f ns per instruction per SIMD = %.2f cycles at 2.4 GHz\n",
			   name, wps, mean / n_instr, mean / (ms * 1e3), ms, ms * 1e6 / (n_instr * wps), ms * 1e6 / (n_instr * wps) * 2.4);
	}
	hipFree(out); hipFree(cyc);
}

int main()
{
	run<0>("v_max_i32", 64);
	run<1>("v_add_u32", 64);
	run<2>("v_mov_b32 dpp", 64);
	run<3>("v_cmp + v_cndmask", 64);
	run<4>("v_pk_max_i16 / v_pk_add_i16", 64);
	run<8>("v_alignbit / v_and_or", 64);
	run<9>("v_cmp_sdwa + v_addc (mask)", 64);
	run<10>("v_max_i32 with EXEC = 0", 64);
	run<11>("v_readfirstlane_b32", 64);
	run<5>("v_max_i32, lanes 0..31", 64);
	run<6>("v_max_i32, lanes 0..15", 64);
	run<7>("v_max_i32, even lanes", 64);
	run_queue();
	return 0;
}
