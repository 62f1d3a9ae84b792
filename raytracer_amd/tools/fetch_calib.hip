// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the traversal kernels: one lane
// reads one 64-byte (or 128-byte) record at a random place, as four (eight) 16-byte loads.  The byte count the
// kernel asks for is known, so FETCH_SIZE / (records * record bytes) is the factor to apply; a streaming kernel
// (16 bytes per lane, coalesced) is there for comparison.  Run each kernel under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib
// and read profiles/r02_fetch_calib.txt for what came out.
//
//   fetch_calib [table MiB = 1024] [reads per thread = 8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
	do {                                                                                      \
		hipError_t e_ = (x);                                                                  \
		if (e_ != hipSuccess) {                                                               \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
			return 1;                                                                         \
		}                                                                                     \
	} while (0)

__device__ __forceinline__ unsigned mix(unsigned h, unsigned v) {
	h = (h + 0x7F4A7C15u) ^ (v * 0x9E3779B1u);
	h ^= h >> 16;
	h *= 0x85EBCA6Bu;
	h ^= h >> 13;
	h *= 0xC2B2AE35u;
	h ^= h >> 16;
	return h;
}

// Each thread reads `reads` records of QUADS float4 at independent random places (every lane of a wave somewhere else).
template <int QUADS>
__global__ __launch_bounds__(256) void k_random_records(const float4* __restrict__ table, unsigned nrec, int reads, float* __restrict__ out) {
	const unsigned t = blockIdx.x * 256 + threadIdx.x;
	float acc = 0.0f;
	for (int r = 0; r < reads; ++r) {
		const unsigned idx = mix(t, (unsigned)r) % nrec;
		const float4* p = table + (size_t)idx * QUADS;
#pragma unroll
		for (int q = 0; q < QUADS; ++q) {
			const float4 v = p[q];
			acc += v.x + v.y + v.z + v.w;
		}
	}
	out[t] = acc;
}

// Coalesced streaming read of n float4.
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ table, size_t n, float* __restrict__ out) {
	const size_t stride = (size_t)gridDim.x * 256;
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const float4 v = table[i];
		acc += v.x + v.y + v.z + v.w;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
	const size_t mib = argc > 1 ? (size_t)std::atoll(argv[1]) : 1024;
	const int reads = argc > 2 ? std::atoi(argv[2]) : 8;
	const size_t bytes = mib << 20;
	const unsigned threads = 1u << 21;  // 2 M threads
	float4* table = nullptr;
	float* out = nullptr;
	CHECK(hipMalloc((void**)&table, bytes));
	CHECK(hipMalloc((void**)&out, (size_t)threads * sizeof(float)));
	CHECK(hipMemset(table, 0, bytes));
	CHECK(hipDeviceSynchronize());
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	auto report = [&](const char* name, double askedBytes) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return;
		std::printf("%-22s table %5zu MiB  asked %.1f MB  %.3f ms  %.1f GB/s\n", name, mib, askedBytes / 1e6, ms, askedBytes / ms / 1e6);
	};
	for (int rep = 0; rep < 2; ++rep) {  // the second round of each is the one to read (warm instruction cache, table touched)
		CHECK(hipEventRecord(a));
		hipLaunchKernelGGL(k_random_records<4>, dim3(threads / 256), dim3(256), 0, 0, table, (unsigned)(bytes / 64), reads, out);
		CHECK(hipEventRecord(b));
		CHECK(hipDeviceSynchronize());
		report("k_random_records<4>", (double)threads * reads * 64.0);
		CHECK(hipEventRecord(a));
		hipLaunchKernelGGL(k_random_records<8>, dim3(threads / 256), dim3(256), 0, 0, table, (unsigned)(bytes / 128), reads, out);
		CHECK(hipEventRecord(b));
		CHECK(hipDeviceSynchronize());
		report("k_random_records<8>", (double)threads * reads * 128.0);
		CHECK(hipEventRecord(a));
		hipLaunchKernelGGL(k_stream, dim3(threads / 256 / 4), dim3(256), 0, 0, table, bytes / 16, out);
		CHECK(hipEventRecord(b));
		CHECK(hipDeviceSynchronize());
		report("k_stream", (double)bytes);
	}
	CHECK(hipFree(table));
	CHECK(hipFree(out));
	return 0;
}
