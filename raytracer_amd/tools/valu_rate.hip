// Issue rates of the vector instructions the traversal kernel is made of, measured the way that kernel runs: blocks of
// 256 threads, 2 or 6 of them per CU (2 or 6 waves per SIMD), eight independent dependency chains per wave.
// Prints one line per instruction and block count: wave-instructions per nanosecond per SIMD.  The issue-time
// model of profiles/summarize.py divides the kernel's instruction counts per class by these rates.
//   hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHAIN8(OP, TAIL)                                                                                                \
	asm volatile(OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP  \
					" %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"                                      \
				 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                          \
				 : "v"(a), "v"(b)                                                                                          \
				 : "vcc", "s20", "s21")

enum Kind { FMA, MUL, ADD, SUB, OR, ADDU, CVTUB, MAX3, MIN, MINU, CMPVCC, CMPSGPR, CNDMASK, CNDMASK_E64, PERM, LSHLOR, ANDOR, MOV, BFE, LSHLADD, ADDC, FMAMIX, CVTF16, NKINDS };
const char* NAMES[NKINDS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_or_b32", "v_add_u32", "v_cvt_f32_ubyte1", "v_max3_f32", "v_min_f32",
							 "v_min_u32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr", "v_perm_b32", "v_lshl_or_b32",
							 "v_and_or_b32", "v_mov_b32", "v_bfe_u32", "v_lshl_add_u32", "v_addc_co_u32", "v_fma_mix_f32", "v_cvt_f32_f16"};

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (KIND == FMA) CHAIN8("v_fma_f32", ", %8, %9");
			else if (KIND == MUL) CHAIN8("v_mul_f32", ", %8");
			else if (KIND == ADD) CHAIN8("v_add_f32", ", %8");
			else if (KIND == SUB) CHAIN8("v_sub_f32", ", %8");
			else if (KIND == OR) CHAIN8("v_or_b32", ", %8");
			else if (KIND == ADDU) CHAIN8("v_add_u32", ", %8");
			else if (KIND == CVTUB) CHAIN8("v_cvt_f32_ubyte1", "");
			else if (KIND == MAX3) CHAIN8("v_max3_f32", ", %8, %9");
			else if (KIND == MIN) CHAIN8("v_min_f32", ", %8");
			else if (KIND == MINU) CHAIN8("v_min_u32", ", %8");
			else if (KIND == CMPVCC)
				asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
							 "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
							 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
			else if (KIND == CMPSGPR)
				asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[20:21], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[20:21], %3, %8\n"
							 "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[20:21], %7, %8\n"
							 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
			else if (KIND == CNDMASK) CHAIN8("v_cndmask_b32", ", %8, vcc");
			else if (KIND == CNDMASK_E64) CHAIN8("v_cndmask_b32", ", %8, s[20:21]");
			else if (KIND == PERM) CHAIN8("v_perm_b32", ", %8, %9");
			else if (KIND == LSHLOR) CHAIN8("v_lshl_or_b32", ", 1, %9");
			else if (KIND == ANDOR) CHAIN8("v_and_or_b32", ", %8, %9");
			else if (KIND == MOV)
				asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"
							 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
			else if (KIND == BFE) CHAIN8("v_bfe_u32", ", 3, 8");
			else if (KIND == LSHLADD) CHAIN8("v_lshl_add_u32", ", 1, %9");
			else if (KIND == FMAMIX) CHAIN8("v_fma_mix_f32", ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]");  // an f16 from the high half times an f32 plus an f32
			else if (KIND == CVTF16) CHAIN8("v_cvt_f32_f16", "");
			else if (KIND == ADDC)
				asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
							 "v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
							 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
void run(int blocksPerCU, int numCUs) {
	float* d = nullptr;
	const int blocks = numCUs * blocksPerCU;
	if (hipMalloc(&d, (size_t)blocks * 256 * 4) != hipSuccess) exit(1);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int iters = 6000;
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0.0f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double waveInstr = (double)blocks * 4 * iters * 64;
	printf("rate %-20s waves_per_simd %d  %.3f ms  per_simd_instr_per_ns %.4f\n", NAMES[KIND], blocksPerCU, ms, waveInstr / (ms * 1e6) / (numCUs * 4.0));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	(void)hipFree(d);
}

template <int KIND>
void run_all(int numCUs) {
	for (int b : {2, 6, 7}) run<KIND>(b, numCUs);
	if constexpr (KIND + 1 < NKINDS) run_all<KIND + 1>(numCUs);
}

int main() {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
	printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
	run_all<0>(prop.multiProcessorCount);
	return 0;
}
