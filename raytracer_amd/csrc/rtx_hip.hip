// MI355X (gfx950, wave64) implementation of the render hot path of sky-5462/RayTracer
// behind the C ABI of include/rtx.h.
//
// What the reference does per pixel on the CPU (src/RayTracer.cpp:237-246: four camera rays,
// a branching recursive color(), accumulate) is done here as a wavefront pipeline:
//
//   raygen            Camera::getRay (src/Camera.cpp:40-49) for a batch of pixel blocks
//   per path level d: trace   BVH::hit + Triangle::hit + closest-hit select
//                             (src/BVH.cpp:162-192, src/Triangle.cpp:23-93, src/RayTracer.cpp:139-151)
//                     shade   miss/sky, normal, termination, emission, child-ray generation
//                             (src/RayTracer.cpp:154-214, src/Triangle.cpp:95-179), children appended
//                             to the next level's queue with wave64 ballot/popcount prefix sums
//   per level, deepest first: gather  the reference's own combine formulas
//                             (src/RayTracer.cpp:184,193-197,205-222) over each node's children
//   accumulate        accumulateImg(row,col) += (sum of 4 samples) * 0.25f (src/RayTracer.cpp:242-246)
//
// Because every node's children occupy consecutive queue slots, the gather pass adds them in
// the reference's order and applies the reference's scale factors afterwards (sum, then
// scale), so the accumulator is bit-identical to the recursive evaluation, not merely close.
//
// Arithmetic discipline (compile with -ffp-contract=off, IEEE divide/sqrt, no fast-math):
// every expression below is written in the evaluation order Eigen gives the reference
// (4-float dot = (x+z)+(y+w), w lanes are zero), fused multiply-adds appear only where the
// reference has explicit FMA intrinsics (src/Triangle.cpp:66,79), and transcendental values
// come from host-computed tables.
//
// Traversal visits exactly the candidates the reference visits (same slab predicate per
// node, src/BVH.cpp:14-41) minus nodes that provably cannot hold the winner: boxes wholly
// behind t = 0.001 or wholly beyond the best hit so far, judged on boxes inflated by a
// position-space safety margin.  Equal-t ties are resolved by the reference's depth-first
// (right child first) candidate order, which is precomputed per triangle.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_wide.h"

#define RTX_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
	g_err = msg;
	return code;
}

#define HIPCHK(expr)                                                                                   \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if (e_ != hipSuccess)                                                                          \
			return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
	} while (0)

// ------------------------------------------------------------------------------------------------
// Device-side scene description, passed to kernels by value.

constexpr int TILE = 64;            // image tile edge in pixels (sharding granule)
constexpr int BLK = 4;              // a wave covers a BLK x BLK pixel block x 4 sub-samples
constexpr int BLOCKS_PER_TILE = (TILE / BLK) * (TILE / BLK);
constexpr int SAMPLES_PER_BLOCK = BLK * BLK * 4;  // 64 = one wave
constexpr int TRACE_THREADS = 256;

enum NodeKind : int { KIND_LEAF = 0, KIND_METAL = 1, KIND_TRANSP = 2, KIND_DIEL = 3 };
constexpr int FLAG_TEX = 0x100;
constexpr int FLAG_TIR = 0x200;

constexpr int TRI_EMIT = 1, TRI_METAL = 2, TRI_TRANSP = 4;

struct DevScene {
	const float4* pairs;     // 4 x float4 per child pair: Lmin|codeL, Lmax|-, Rmin|codeR, Rmax|-
	const float4* wide;      // 4 x float4 per node of the 4-wide quantised collapse of the tree
	const float4* triHit;    // 4 x float4 per triangle in candidate-rank order: n, v0, v1, v2, index
	const float4* triShade;  // 4 x float4 per triangle in candidate-rank order: n0 r | n1 g | n2 b | rough ior tex flags
	const float4* triUv;     // 2 x float4 per triangle: uv0 uv1 | uv2 - -   (read for textured hits only)
	const uint8_t* texels;
	const int4* texDesc;     // byte offset, width, height
	const float* cosT;
	const float* sinT;
	const float* gammaT;
	float4 camO, camLUC, camRS, camDS;
	float4 bg;
	int width, height, maxDepth, nDiff, nSpec;
	int hasSky;
	float skyBrightness;
	int skyTex[6];
	uint32_t seedKey;
	float sceneAbsMax;
	int tilesX;
};

// ------------------------------------------------------------------------------------------------
// Small device helpers.  dot4 is Eigen's reduction order with the (zero) w lane kept as "+ 0.0f".

__device__ __forceinline__ float dot3w(float ax, float ay, float az, float bx, float by, float bz) {
	return (ax * bx + az * bz) + (ay * by + 0.0f);
}

__device__ __forceinline__ uint32_t rt_mix(uint32_t h, uint32_t v) {
	h = (h + 0x7F4A7C15u) ^ (v * 0x9E3779B1u);
	h ^= h >> 16;
	h *= 0x85EBCA6Bu;
	h ^= h >> 13;
	h *= 0xC2B2AE35u;
	h ^= h >> 16;
	return h;
}
__device__ __forceinline__ uint32_t rt_child_key(uint32_t key, uint32_t ordinal) { return rt_mix(key, 1u + ordinal); }
__device__ __forceinline__ uint32_t rt_draw(uint32_t key, uint32_t lobe, uint32_t k, uint32_t dim) {
	return rt_mix(key, 0x80000000u | (lobe << 20) | (k << 1) | dim);
}

__device__ __forceinline__ float clamp01(float u) {
	u = (u >= 0.0f) ? u : 0.0f;
	u = (u <= 1.0f) ? u : 1.0f;
	return u;
}

// Texture::sampleTexture (src/Texture.cpp:41-53) with clamped coordinates and the
// powf(c/255, 2.2) decode taken from the host-computed table.
__device__ __forceinline__ float3 tex_fetch(const DevScene& sc, int id, float u, float v) {
	int4 d = sc.texDesc[id];
	u = clamp01(u);
	v = clamp01(v);
	float uMax = (float)(d.y - 1);
	float vMax = (float)(d.z - 1);
	int sx = (int)roundf(u * uMax);
	int sy = (int)roundf(v * vMax);
	const uint8_t* c = sc.texels + (size_t)(uint32_t)d.x + ((size_t)sy * (size_t)d.y + (size_t)sx) * 3;
	return make_float3(sc.gammaT[c[0]], sc.gammaT[c[1]], sc.gammaT[c[2]]);
}

// Skybox::sampleBackground (src/Skybox.cpp:32-84).
__device__ float3 sky_fetch(const DevScene& sc, float dx, float dy, float dz) {
	for (int i = 0; i < 6; ++i) {
		float divisor, u, v;
		float px, py, pz;
		switch (i) {
		case 0:
			divisor = dz;
			px = dx / divisor; py = dy / divisor;
			u = (1.0f - px) * 0.5f;
			v = (1.0f + py) * 0.5f;
			break;
		case 1:
			divisor = -dz;
			px = dx / divisor; py = dy / divisor;
			u = (1.0f + px) * 0.5f;
			v = (1.0f + py) * 0.5f;
			break;
		case 2:
			divisor = -dx;
			pz = dz / divisor; py = dy / divisor;
			u = (1.0f - pz) * 0.5f;
			v = (1.0f + py) * 0.5f;
			break;
		case 3:
			divisor = dx;
			pz = dz / divisor; py = dy / divisor;
			u = (pz + 1.0f) * 0.5f;
			v = (py + 1.0f) * 0.5f;
			break;
		case 4:
			divisor = dy;
			px = dx / divisor; pz = dz / divisor;
			u = (px + 1.0f) * 0.5f;
			v = (pz + 1.0f) * 0.5f;
			break;
		default:
			divisor = -dy;
			px = dx / divisor; pz = dz / divisor;
			u = (px + 1.0f) * 0.5f;
			v = (pz - 1.0f) * 0.5f;
			break;
		}
		if (divisor < 0.0f || u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) continue;
		float3 c = tex_fetch(sc, sc.skyTex[i], u, v);
		return make_float3(c.x * sc.skyBrightness, c.y * sc.skyBrightness, c.z * sc.skyBrightness);
	}
	return make_float3(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Level-0 sample order.  Sample j of a batch that starts at pixel block g0:
//   block g = g0 + j/64 -> (tile slot, block in tile); lane = j%64 -> (pixel in block, sub-sample).
// A wave therefore holds the 4 sub-samples of a 4x4 pixel block, which keeps primary rays of
// one wave spatially adjacent.  Tiles are taken from the arithmetic list first + k*stride.

struct Batch {
	int g0;          // first pixel block (global over the scene's tile list)
	int nSamples;    // (number of blocks) * 64
	int tileFirst, tileStride;
};

__device__ __forceinline__ bool sample_to_pixel(const DevScene& sc, const Batch& b, int j, int& x, int& y, int& sub) {
	int g = b.g0 + (j >> 6);
	int lane = j & 63;
	int slot = g / BLOCKS_PER_TILE;
	int blk = g - slot * BLOCKS_PER_TILE;
	int tile = b.tileFirst + slot * b.tileStride;
	int ty = tile / sc.tilesX;
	int tx = tile - ty * sc.tilesX;
	int by = blk / (TILE / BLK);
	int bx = blk - by * (TILE / BLK);
	int pix = lane >> 2;
	sub = lane & 3;
	x = tx * TILE + bx * BLK + (pix & 3);
	y = ty * TILE + by * BLK + (pix >> 2);
	return x < sc.width && y < sc.height;
}

// Camera::getRay (src/Camera.cpp:37-49) plus the per-sample RNG key.
__global__ __launch_bounds__(256) void k_raygen(DevScene sc, Batch b, uint32_t frame, float4* __restrict__ rayO,
												float4* __restrict__ rayD) {
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= b.nSamples) return;
	int x, y, sub;
	bool valid = sample_to_pixel(sc, b, j, x, y, sub);
	if (!valid) {
		rayO[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		rayD[j] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);  // w = 1 marks a dead slot
		return;
	}
	const float xo = (sub & 1) ? 0.25f : -0.25f;
	const float yo = (sub & 2) ? 0.25f : -0.25f;
	float xTemp = (float)x + xo;
	float yTemp = (float)y + yo;
	float dx = (sc.camLUC.x + xTemp * sc.camRS.x) + yTemp * sc.camDS.x;
	float dy = (sc.camLUC.y + xTemp * sc.camRS.y) + yTemp * sc.camDS.y;
	float dz = (sc.camLUC.z + xTemp * sc.camRS.z) + yTemp * sc.camDS.z;
	uint32_t sample = ((uint32_t)y * (uint32_t)sc.width + (uint32_t)x) * 4u + (uint32_t)sub;
	uint32_t key = rt_mix(rt_mix(sc.seedKey, frame), sample);
	rayO[j] = make_float4(sc.camO.x, sc.camO.y, sc.camO.z, __uint_as_float(key));
	rayD[j] = make_float4(dx, dy, dz, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Traversal.

struct HitState {
	float t, alpha, beta;
	int rank;
};

// Triangle::hit (src/Triangle.cpp:23-93) for plane normal n and vertices v0, v1, v2, folded with the
// closest-hit rule of src/RayTracer.cpp:145 extended by the candidate-order tie break.
// Returns true when the triangle was accepted (any-hit callers stop there).
__device__ __forceinline__ bool tri_test_vals(float nx, float ny, float nz, float v0x, float v0y, float v0z, float v1x,
											  float v1y, float v1z, float v2x, float v2y, float v2z, int rank, float ox,
											  float oy, float oz, float dx, float dy, float dz, HitState& best) {
	float temp = dot3w(nx, ny, nz, v0x - ox, v0y - oy, v0z - oz);
	float t = temp / dot3w(nx, ny, nz, dx, dy, dz);
	if (!(t > 0.001f)) return false;
	if (t > best.t || (t == best.t && rank > best.rank)) return false;
	float bx = (ox + t * dx) - v2x;
	float by = (oy + t * dy) - v2y;
	float bz = (oz + t * dz) - v2z;
	// rows (a1_i, a2_i, b_i), i = x,y,z with a1 = v0 - v2, a2 = v1 - v2 (src/Triangle.cpp:31-32)
	float r00 = v0x - v2x, r01 = v1x - v2x, r02 = bx;
	float r10 = v0y - v2y, r11 = v1y - v2y, r12 = by;
	float r20 = v0z - v2z, r21 = v1z - v2z, r22 = bz;
	// first pivot: largest |column 0|, strict >, rows tried in order 0,1,2
	bool s1 = fabsf(r10) > fabsf(r00);
	float mx = s1 ? fabsf(r10) : fabsf(r00);
	bool s2 = fabsf(r20) > mx;
	float p0 = s2 ? r20 : (s1 ? r10 : r00);
	float p1 = s2 ? r21 : (s1 ? r11 : r01);
	float p2 = s2 ? r22 : (s1 ? r12 : r02);
	bool m1 = s1 && !s2;
	float a0 = m1 ? r00 : r10, a1 = m1 ? r01 : r11, a2 = m1 ? r02 : r12;  // new row 1
	float c0 = s2 ? r00 : r20, c1 = s2 ? r01 : r21, c2 = s2 ? r02 : r22;  // new row 2
	p1 = p1 / p0;
	p2 = p2 / p0;
	a1 = __fmaf_rn(-p1, a0, a1);
	a2 = __fmaf_rn(-p2, a0, a2);
	c1 = __fmaf_rn(-p1, c0, c1);
	c2 = __fmaf_rn(-p2, c0, c2);
	// second pivot
	bool s3 = fabsf(c1) > fabsf(a1);
	float e1 = s3 ? c1 : a1;
	float e2 = s3 ? c2 : a2;
	e2 = e2 / e1;
	float alpha = __fmaf_rn(-e2, p1, p2);
	float beta = e2;
	bool accept = (0.0f <= alpha) & (alpha <= 1.0f) & (0.0f <= beta) & (beta <= 1.0f) & (alpha + beta <= 1.0f);
	if (!accept) return false;
	best.t = t;
	best.alpha = alpha;
	best.beta = beta;
	best.rank = rank;
	return true;
}

// The triangle stored at `rank`: q0 = n.xyz v0.x | q1 = v0.yz v1.xy | q2 = v1.z v2.xyz | q3 = id - - -
__device__ __forceinline__ bool tri_test(const float4* __restrict__ triHit, int rank, float ox, float oy, float oz,
										 float dx, float dy, float dz, HitState& best) {
	const float4* T = triHit + (size_t)rank * 4;
	float4 q0 = T[0], q1 = T[1], q2 = T[2];
	return tri_test_vals(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, rank, ox, oy, oz, dx, dy,
						 dz, best);
}

// Per-ray constants of the slab test.
struct RaySlab {
	float ox, oy, oz, ix, iy, iz, mx, my, mz;
	bool nx, ny, nz;
};

// AABB::hit (src/BVH.cpp:14-41) for one box, plus the conservative entry/exit bounds used
// for culling.  The folds over lanes 1 and 2 ignore NaN exactly like the reference's
// "t0 > tmin ? t0 : tmin"; a NaN in lane 0 makes the reference's final comparison false,
// i.e. a hit, which the unordered test reproduces.
__device__ __forceinline__ bool box_test(const RaySlab& r, float4 bmin, float4 bmax, float& te, float& tx) {
	float t0x = (bmin.x - r.ox) * r.ix, t1x = (bmax.x - r.ox) * r.ix;
	float t0y = (bmin.y - r.oy) * r.iy, t1y = (bmax.y - r.oy) * r.iy;
	float t0z = (bmin.z - r.oz) * r.iz, t1z = (bmax.z - r.oz) * r.iz;
	float nX = r.nx ? t1x : t0x, fX = r.nx ? t0x : t1x;
	float nY = r.ny ? t1y : t0y, fY = r.ny ? t0y : t1y;
	float nZ = r.nz ? t1z : t0z, fZ = r.nz ? t0z : t1z;
	float tmin = fmaxf(fmaxf(nX, nY), nZ);
	float tmax = fminf(fminf(fX, fY), fZ);
	bool hit = !(tmax < tmin) || (nX != nX) || (fX != fX);
	te = fmaxf(fmaxf(nX - r.mx, nY - r.my), nZ - r.mz);
	tx = fminf(fminf(fX + r.mx, fY + r.my), fZ + r.mz);
	return hit;
}

// hits[j] = (rank or -1 / -2 for a dead slot, t, alpha, beta), all as raw bits of a float4.
// ANYHIT: only "some candidate is accepted" is needed (rays at the last recursion depth,
// src/RayTracer.cpp:154-171); x = 0 then stands for "hit".
// EXHAUSTIVE: no culling, i.e. the candidate set of BVH::hit itself.
//
// Persistent waves: rays of one queue differ in traversal length by orders of magnitude, so a wave
// that takes 64 rays and waits for the slowest runs mostly empty.  Here a wave keeps running and, once
// REFILL or more of its lanes are idle, hands them the next rays of its private chunk of the queue
// (CHUNK consecutive rays claimed with one atomic on *work, which the host zeroes before the launch).
// Every wave leaves the loop when the queue is exhausted and its lanes have drained.
constexpr int TRACE_CHUNK = 256;
constexpr int TRACE_REFILL = 16;

template <int STACK, bool ANYHIT, bool EXHAUSTIVE, bool STATS>
__global__ __launch_bounds__(TRACE_THREADS) void k_trace(DevScene sc, const float4* __restrict__ rayO,
														  const float4* __restrict__ rayD, float4* __restrict__ hits,
														  int n, unsigned long long* __restrict__ stats,
														  unsigned* __restrict__ work, int refill, int chunk, int triBatch,
														  const int* __restrict__ list, const unsigned* __restrict__ listCount,
														  const unsigned* __restrict__ nDev) {
	// the queue length may live on the device (written by the previous level's shade kernel); n is then the
	// capacity of the arrays
	if (nDev) n = (int)min(*nDev, (unsigned)n);
	// with a list: the rays to trace are list[0 .. *listCount), the ones the wide traversal set aside
	if (list) n = (int)*listCount;
	__shared__ int stk[STACK][TRACE_THREADS];
	const int tid = threadIdx.x;
	const unsigned lane = (unsigned)tid & 63u;
	const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;

	unsigned nodeTests = 0, triTests = 0;
	// wave-uniform: the chunk this wave is handing out
	int cnext = 0, cend = 0;
	bool more = true, firstChunk = true;
	// per lane: the ray in flight (cur < 0: none)
	int j = -1, cur = -1, sp = 0;
	RaySlab r;
	r.ox = r.oy = r.oz = r.ix = r.iy = r.iz = r.mx = r.my = r.mz = 0.0f;
	r.nx = r.ny = r.nz = false;
	float dx = 0.0f, dy = 0.0f, dz = 0.0f;
	HitState best;
	best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
	bool found = false;

	// Leaf children are not intersected where they are found: a lane parks them (at most two, the two
	// children of one pair) and waits.  Once triBatch lanes are parked, or nobody can traverse any more,
	// the wave runs the triangle code once for all of them.  The address path issues one vector-memory
	// instruction per ~22 clocks per CU however few lanes it serves, so four loads for one or two lanes
	// per step would cost as much as the box loads of the whole wave.
	bool busy = false;
	int pend0 = -1, pend1 = -1;
	for (;;) {
		const unsigned long long idle = __ballot(!busy);
		const int nidle = __popcll(idle);
		if (more && nidle >= refill) {
			if (cnext >= cend) {
				// A wave's first chunk is the one with its own number; later ones come from the shared cursor,
				// which therefore starts behind those.  A launch over a short queue then costs no atomics at
				// all (one address takes only ~90 of them per microsecond, and the grid has thousands of waves).
				unsigned base = 0;
				if (firstChunk) {
					base = (blockIdx.x * (TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6)) * (unsigned)chunk;
					firstChunk = false;
				}
				else {
					if (lane == 0) base = atomicAdd(work, (unsigned)chunk);
					base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + gridDim.x * (TRACE_THREADS / 64) * (unsigned)chunk;
				}
				if (base >= (unsigned)n) {
					more = false;
					cnext = cend = 0;
				}
				else {
					cnext = (int)base;
					cend = min((int)base + chunk, n);
				}
			}
			const int take = min(nidle, cend - cnext);
			if (take > 0) {
				const int slot = __popcll(idle & below);
				if (!busy && slot < take) {
					j = cnext + slot;
					if (list) j = list[j];
					float4 O = rayO[j];
					float4 D = rayD[j];
					if (D.w != 0.0f) {
						hits[j] = make_float4(__int_as_float(-2), FLT_MAX, 0.0f, 0.0f);
					}
					else {
						r.ox = O.x; r.oy = O.y; r.oz = O.z;
						dx = D.x; dy = D.y; dz = D.z;
						r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
						r.nx = (__float_as_uint(r.ix) >> 31) != 0;
						r.ny = (__float_as_uint(r.iy) >> 31) != 0;
						r.nz = (__float_as_uint(r.iz) >> 31) != 0;
						// Position-space slack for culling: 2^-14 of the largest coordinate magnitude in play.
						float scale = fmaxf(fmaxf(sc.sceneAbsMax, fabsf(r.ox)), fmaxf(fabsf(r.oy), fabsf(r.oz)));
						float eps = scale * 6.103515625e-05f;
						r.mx = eps * fabsf(r.ix); r.my = eps * fabsf(r.iy); r.mz = eps * fabsf(r.iz);
						best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
						found = false;
						sp = 0;
						cur = 0;
						pend0 = pend1 = -1;
						busy = true;
						// Rays no triangle can accept are answered without traversal.  A zero direction (the
						// reference produces one when a diffuse offset cancels the normal, src/Triangle.cpp:115)
						// makes n.d = +-0, so t is +-inf or NaN and the hit point NaN; a non-finite direction or
						// a NaN origin makes t NaN or zero.  Triangle::hit rejects all of these
						// (src/Triangle.cpp:28-29,84), yet BVH::hit would hand it a large part of the tree.
						if (!EXHAUSTIVE) {
							const bool zero = (dx == 0.0f) & (dy == 0.0f) & (dz == 0.0f);
							const bool finite = (fabsf(dx) <= FLT_MAX) & (fabsf(dy) <= FLT_MAX) & (fabsf(dz) <= FLT_MAX) &
												(r.ox == r.ox) & (r.oy == r.oy) & (r.oz == r.oz);
							if (zero | !finite) cur = -1;
						}
					}
				}
				cnext += take;
			}
		}
		const bool canWalk = busy && cur >= 0 && pend0 < 0;
		const unsigned long long mWalk = __ballot(canWalk);
		const unsigned long long mPark = __ballot(busy && pend0 >= 0);
		if (mWalk == 0ull && mPark == 0ull) {
			if (__ballot(busy) == 0ull) {
				if (!more) break;
				continue;
			}
		}
		else if (__popcll(mPark) >= triBatch || mWalk == 0ull) {
			// triangle phase; a second round serves the lanes that parked two leaves
			for (int rep = 0; rep < 2; ++rep) {
				if (busy && pend0 >= 0) {
					if (STATS) triTests++;
					found |= tri_test(sc.triHit, pend0, r.ox, r.oy, r.oz, dx, dy, dz, best);
					pend0 = pend1;
					pend1 = -1;
					if (ANYHIT && found) {
						pend0 = -1;
						cur = -1;
					}
				}
				if (__ballot(busy && pend0 >= 0) == 0ull) break;
			}
		}
		else if (canWalk) {
			const float4* P = sc.pairs + (size_t)cur * 4;
			float4 lmin = P[0], lmax = P[1], rmin = P[2], rmax = P[3];
			int codeL = __float_as_int(lmin.w);
			int codeR = __float_as_int(rmin.w);
			float teL, txL, teR, txR;
			bool hitL = box_test(r, lmin, lmax, teL, txL) && codeL != 0;
			bool hitR = box_test(r, rmin, rmax, teR, txR) && codeR != 0;
			if (STATS) nodeTests += (codeL != 0) + (codeR != 0);
			if (!EXHAUSTIVE) {
				hitL = hitL && !(txL < 0.001f) && !(teL > best.t);
				hitR = hitR && !(txR < 0.001f) && !(teR > best.t);
			}
			if (hitL && codeL < 0) pend0 = ~codeL;
			if (hitR && codeR < 0) {
				if (pend0 < 0) pend0 = ~codeR;
				else pend1 = ~codeR;
			}
			const bool intL = hitL && codeL > 0;
			const bool intR = hitR && codeR > 0;
			int next = -1;
			if (intL && intR) {
				bool lfirst = teL <= teR;
				next = lfirst ? codeL : codeR;
				if (sp < STACK) {  // the host rejects trees deeper than the stack
					stk[sp][tid] = lfirst ? codeR : codeL;
					sp++;
				}
			}
			else if (intL) next = codeL;
			else if (intR) next = codeR;
			else if (sp > 0) {
				sp--;
				next = stk[sp][tid];
			}
			cur = next;
		}
		if (busy && cur < 0 && pend0 < 0) {
			int id = found ? (ANYHIT ? 0 : best.rank) : -1;
			hits[j] = make_float4(__int_as_float(id), best.t, best.alpha, best.beta);
			busy = false;
		}
	}
	if (STATS) {
		// wave-level reduction, then one atomic per wave
		for (int off = 32; off > 0; off >>= 1) {
			nodeTests += __shfl_down(nodeTests, off, 64);
			triTests += __shfl_down(triTests, off, 64);
		}
		if (lane == 0) {
			atomicAdd(&stats[0], (unsigned long long)nodeTests);
			atomicAdd(&stats[1], (unsigned long long)triTests);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Wide traversal.
//
// For a ray whose three reciprocals 1/d are finite, every value the reference's slab test computes is
// finite, and (bound - o) * (1/d) is monotone in `bound`.  A child box of the reference's tree lies
// inside its parent's box exactly (both are min/max over vertex sets), so if the slab predicate
// (src/BVH.cpp:14-41) passes for a leaf's own box it passes for every ancestor: for such rays the
// candidate list of BVH::hit is exactly "the triangles whose own box passes".  The interior boxes only
// prune, and any conservative bound may stand in for them.  This kernel therefore walks a 4-wide tree of
// its own (rtx_wide.h) whose child boxes are quantised to 8 bits per coordinate inside the node's box,
// rounded outwards, and tested inflated by the same position-space slack as the culling bounds.  Leaves
// are tested with the exact predicate on their exact box (recomputed from the vertices) and the exact
// Triangle::hit, and ties go to the reference's candidate order, so the result equals the reference's.
//
// A ray parallel to an axis (1/d = +-inf there) is walked here too.  On that axis the reference computes
// +-inf, or NaN where the origin lies on a box plane.  Its folds over lanes 1 and 2 (y, z) skip a NaN, which
// is the same as the unconstrained (-inf, +inf) a parent's wider slab gives, so the argument above still
// holds.  A NaN in lane 0 (x) makes the reference accept the box whatever the other axes say, which a
// parent need not do: a ray that meets that case at a leaf is set aside in a list for the binary-tree kernel
// above, which evaluates every ancestor as the reference does; so are rays with a finite reciprocal beyond
// 2^100 (whose products overflow) and rays that need more stack than this kernel has.  The quantised
// tests are homogeneous in 1/d; they use +-2^100 in place of +-inf and so prune a parallel ray by
// "origin inside the slab, give or take the slack".
//
// Node, 4 x 16 bytes:  px py pz sx                          (origin; sx, sy, sz: quantisation steps, powers of two)
//                      lo.x[4] hi.x[4] lo.y[4] hi.y[4]      (one byte per child)
//                      lo.z[4] hi.z[4] code0 code1
//                      code2 code3 sy sz                       code > 0 node, < 0 ~rank, 0 absent
constexpr float WIDE_MAX_INV = 1.2676506e30f;  // 2^100

__device__ __forceinline__ float ubyte_f(unsigned v, int k) { return (float)((v >> (8 * k)) & 0xFFu); }

template <int STACK, bool ANYHIT, bool STATS>
__global__ __launch_bounds__(TRACE_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_trace_wide(DevScene sc, const float4* __restrict__ rayO,
															   const float4* __restrict__ rayD, float4* __restrict__ hits,
															   int n, unsigned long long* __restrict__ stats,
															   unsigned* __restrict__ work, int* __restrict__ unsafeList,
															   unsigned* __restrict__ unsafeCount, int refill, int chunk,
															   int triBatch, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	__shared__ int stk[STACK][TRACE_THREADS];
	const int tid = threadIdx.x;
	const unsigned lane = (unsigned)tid & 63u;
	const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
	unsigned boxTests = 0, triTests = 0, nodeVisits = 0, leafFetches = 0;
	unsigned walkIters = 0, leafIters = 0, walkLanes = 0, leafLanes = 0;  // wave-level, kept by lane 0
	unsigned overflows = 0;
	int cnext = 0, cend = 0;
	bool more = true, firstChunk = true;
	int j = -1, cur = -1, sp = 0;
	RaySlab r;
	r.ox = r.oy = r.oz = r.ix = r.iy = r.iz = r.mx = r.my = r.mz = 0.0f;
	r.nx = r.ny = r.nz = false;
	float dx = 0.0f, dy = 0.0f, dz = 0.0f;
	float wix = 0.0f, wiy = 0.0f, wiz = 0.0f;  // 1/d as the interior tests use it: +-inf replaced by +-2^100
	HitState best;
	best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
	bool found = false, busy = false;
	int pend0 = -1, pend1 = -1;

	for (;;) {
		const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
		const int nidle = __popcll(idle);
		if (more && nidle >= refill) {
			if (cnext >= cend) {
				// A wave's first chunk is the one with its own number; later ones come from the shared cursor,
				// which therefore starts behind those.  A launch over a short queue then costs no atomics at
				// all (one address takes only ~90 of them per microsecond, and the grid has thousands of waves).
				unsigned base = 0;
				if (firstChunk) {
					base = (blockIdx.x * (TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6)) * (unsigned)chunk;
					firstChunk = false;
				}
				else {
					if (lane == 0) base = atomicAdd(work, (unsigned)chunk);
					base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + gridDim.x * (TRACE_THREADS / 64) * (unsigned)chunk;
				}
				if (base >= (unsigned)n) {
					more = false;
					cnext = cend = 0;
				}
				else {
					cnext = (int)base;
					cend = min((int)base + chunk, n);
				}
			}
			const int take = min(nidle, cend - cnext);
			if (take > 0) {
				const int slot = __popcll(idle & below);
				if (!busy && slot < take) {
					j = cnext + slot;
					float4 O = rayO[j];
					float4 D = rayD[j];
					if (D.w != 0.0f) {
						hits[j] = make_float4(__int_as_float(-2), FLT_MAX, 0.0f, 0.0f);
					}
					else {
						r.ox = O.x; r.oy = O.y; r.oz = O.z;
						dx = D.x; dy = D.y; dz = D.z;
						r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
						const bool zero = (dx == 0.0f) & (dy == 0.0f) & (dz == 0.0f);
						const bool finite = (fabsf(dx) <= FLT_MAX) & (fabsf(dy) <= FLT_MAX) & (fabsf(dz) <= FLT_MAX) &
											(r.ox == r.ox) & (r.oy == r.oy) & (r.oz == r.oz);
						const float aix = fabsf(r.ix), aiy = fabsf(r.iy), aiz = fabsf(r.iz);
						const bool safe = ((aix <= WIDE_MAX_INV) | (aix > FLT_MAX)) & ((aiy <= WIDE_MAX_INV) | (aiy > FLT_MAX)) &
										  ((aiz <= WIDE_MAX_INV) | (aiz > FLT_MAX));
						if (zero | !finite) {
							// no triangle can accept such a ray (see k_trace)
							hits[j] = make_float4(__int_as_float(-1), FLT_MAX, 0.0f, 0.0f);
						}
						else if (!safe) {
							unsafeList[atomicAdd(unsafeCount, 1u)] = j;
						}
						else {
							r.nx = (__float_as_uint(r.ix) >> 31) != 0;
							r.ny = (__float_as_uint(r.iy) >> 31) != 0;
							r.nz = (__float_as_uint(r.iz) >> 31) != 0;
							// The interior tests are homogeneous in 1/d, so on an axis the ray is parallel to a large
							// finite reciprocal decides "origin inside the slab, give or take the slack" just as well.
							wix = aix > FLT_MAX ? copysignf(WIDE_MAX_INV, r.ix) : r.ix;
							wiy = aiy > FLT_MAX ? copysignf(WIDE_MAX_INV, r.iy) : r.iy;
							wiz = aiz > FLT_MAX ? copysignf(WIDE_MAX_INV, r.iz) : r.iz;
							float scale = fmaxf(fmaxf(sc.sceneAbsMax, fabsf(r.ox)), fmaxf(fabsf(r.oy), fabsf(r.oz)));
							float eps = scale * 6.103515625e-05f;
							r.mx = eps * fabsf(wix); r.my = eps * fabsf(wiy); r.mz = eps * fabsf(wiz);
							best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
							found = false;
							sp = 0;
							cur = 0;
							pend0 = pend1 = -1;
							busy = true;
						}
					}
				}
				cnext += take;
			}
		}
		// A lane walks while it has something to walk and a free leaf slot: a leaf it comes across is kept
		// for later (two slots) and the walk goes on to what lies behind it.  That is speculative (the leaf
		// may turn out to cut those nodes off) but keeps lanes out of the waiting room; the result does not
		// depend on the order of the tests.
		const bool canWalk = busy && pend1 < 0 && (cur >= 0 || sp > 0);
		const unsigned long long mWalk = __builtin_amdgcn_ballot_w64(canWalk);
		const unsigned long long mLeaf = __builtin_amdgcn_ballot_w64(busy && pend0 >= 0);
		if (mWalk == 0ull && mLeaf == 0ull) {
			if (!more) break;
			continue;
		}
		const bool leafPhase = __popcll(mLeaf) >= triBatch || mWalk == 0ull;
		if (STATS) {
			if (leafPhase) {
				leafIters++;
				leafLanes += __popcll(mLeaf);
			}
			else {
				walkIters++;
				walkLanes += __popcll(mWalk);
			}
		}
		if (leafPhase) {
			// exact box of the triangle, exact slab predicate, exact triangle test; one leaf per lane and pass
			if (busy && pend0 >= 0) {
				const float4* T = sc.triHit + (size_t)pend0 * 4;
				float4 q0 = T[0], q1 = T[1], q2 = T[2];
				if (STATS) leafFetches++;
				const float v0x = q0.w, v0y = q1.x, v0z = q1.y, v1x = q1.z, v1y = q1.w, v1z = q2.x, v2x = q2.y, v2y = q2.z, v2z = q2.w;
				// the reference's predicate on the leaf's own box (src/BVH.cpp:14-41); lane 0 is never NaN here and
				// fmaxf/fminf skip a NaN in lanes 1 and 2 as the reference's folds do
				const float lx = fminf(fminf(v0x, v1x), v2x), hx = fmaxf(fmaxf(v0x, v1x), v2x);
				const float ly = fminf(fminf(v0y, v1y), v2y), hy = fmaxf(fmaxf(v0y, v1y), v2y);
				const float lz = fminf(fminf(v0z, v1z), v2z), hz = fmaxf(fmaxf(v0z, v1z), v2z);
				const float t0x = (lx - r.ox) * r.ix, t1x = (hx - r.ox) * r.ix;
				const float t0y = (ly - r.oy) * r.iy, t1y = (hy - r.oy) * r.iy;
				const float t0z = (lz - r.oz) * r.iz, t1z = (hz - r.oz) * r.iz;
				const float tmin = fmaxf(fmaxf(r.nx ? t1x : t0x, r.ny ? t1y : t0y), r.nz ? t1z : t0z);
				const float tmax = fminf(fminf(r.nx ? t0x : t1x, r.ny ? t0y : t1y), r.nz ? t0z : t1z);
				const bool pass = !(tmax < tmin);
				if (STATS) boxTests++;
				if (t0x != t0x || t1x != t1x) {
					// The origin lies on an x plane of this box and the ray runs along it: the reference accepts
					// such a box whatever y and z say (NaN in lane 0), provided it gets there, which only its own
					// tree can tell.  The ray starts over in the binary-tree kernel.
					unsafeList[atomicAdd(unsafeCount, 1u)] = j;
					busy = false;
					cur = -1;
					sp = 0;
					pend1 = -1;
				}
				else if (pass) {
					if (STATS) triTests++;
					found |= tri_test_vals(q0.x, q0.y, q0.z, v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z, pend0, r.ox, r.oy,
										   r.oz, dx, dy, dz, best);
				}
				pend0 = pend1;
				pend1 = -1;
				if (ANYHIT && found) {
					pend0 = -1;
					cur = -1;
					sp = 0;
				}
			}
		}
		else if (canWalk) {
			int item = 0;  // what this lane looks at next: > 0 node, < 0 ~rank of a leaf, 0 nothing yet
			if (cur >= 0) {
				const float4* W = sc.wide + (size_t)cur * 4;
				float4 w0 = W[0], w1 = W[1], w2 = W[2], w3 = W[3];
				if (STATS) nodeVisits++;
				const float Sx = w0.w * wix, Sy = w3.z * wiy, Sz = w3.w * wiz;
				const float Ax = (w0.x - r.ox) * wix, Ay = (w0.y - r.oy) * wiy, Az = (w0.z - r.oz) * wiz;
				const float Anx = Ax - r.mx, Any = Ay - r.my, Anz = Az - r.mz;
				const float Afx = Ax + r.mx, Afy = Ay + r.my, Afz = Az + r.mz;
				const unsigned lox = __float_as_uint(w1.x), hix = __float_as_uint(w1.y), loy = __float_as_uint(w1.z),
							   hiy = __float_as_uint(w1.w), loz = __float_as_uint(w2.x), hiz = __float_as_uint(w2.y);
				const unsigned nxd = r.nx ? hix : lox, fxd = r.nx ? lox : hix;
				const unsigned nyd = r.ny ? hiy : loy, fyd = r.ny ? loy : hiy;
				const unsigned nzd = r.nz ? hiz : loz, fzd = r.nz ? loz : hiz;
				int cd[4] = {__float_as_int(w2.z), __float_as_int(w2.w), __float_as_int(w3.x), __float_as_int(w3.y)};
				// key: ordered bits of what the children are sorted by, all ones for a child that is not entered
				unsigned key[4];
				int nh = 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					float e = fmaxf(fmaxf(__fmaf_rn(ubyte_f(nxd, c), Sx, Anx), __fmaf_rn(ubyte_f(nyd, c), Sy, Any)),
									__fmaf_rn(ubyte_f(nzd, c), Sz, Anz));
					float x = fminf(fminf(__fmaf_rn(ubyte_f(fxd, c), Sx, Afx), __fmaf_rn(ubyte_f(fyd, c), Sy, Afy)),
									__fmaf_rn(ubyte_f(fzd, c), Sz, Afz));
					e = fmaxf(e, 0.001f);
					const bool h = (cd[c] != 0) & !(x < e) & !(e > best.t);
					// closest hit: nearest child first.  Any hit: the child the ray stays in longest first, where an
					// occluder is likeliest (any order gives the same answer; the key stays below the "not
					// entered" marker because the length is clamped to >= +0).
					key[c] = !h ? 0xFFFFFFFFu : ANYHIT ? 0x7FFFFFFFu - __float_as_uint(fmaxf(x - e, 0.0f)) : __float_as_uint(e);
					nh += h ? 1 : 0;
					if (STATS) boxTests += cd[c] != 0;
				}
				// sorting network on (key, code), nearest first
#define RTX_CSWAP(i, j)                                     \
	{                                                       \
		const bool sw = key[i] > key[j];                    \
		const unsigned klo = min(key[i], key[j]);           \
		const unsigned khi = max(key[i], key[j]);           \
		const int ci = sw ? cd[j] : cd[i];                  \
		const int cj = sw ? cd[i] : cd[j];                  \
		key[i] = klo; key[j] = khi; cd[i] = ci; cd[j] = cj; \
	}
				RTX_CSWAP(0, 1)
				RTX_CSWAP(2, 3)
				RTX_CSWAP(0, 2)
				RTX_CSWAP(1, 3)
				RTX_CSWAP(1, 2)
#undef RTX_CSWAP
				// the nearest is looked at next, the others wait with the farthest at the bottom
				if (sp + nh - 1 > STACK) {
					// no room: the ray starts over in the binary-tree kernel, whose stack covers its whole tree
					unsafeList[atomicAdd(unsafeCount, 1u)] = j;
					if (STATS) overflows++;
					busy = false;
					sp = 0;
					nh = 0;
					pend0 = -1;
				}
				if (nh > 3) stk[sp][tid] = cd[3];
				if (nh > 2) stk[sp + nh - 3][tid] = cd[2];
				if (nh > 1) stk[sp + nh - 2][tid] = cd[1];
				sp += max(nh - 1, 0);
				item = nh > 0 ? cd[0] : 0;
				cur = -1;
			}
			if (item == 0 && sp > 0) {
				sp--;
				item = stk[sp][tid];
			}
			if (item < 0) {
				// a leaf: keep it and look at what comes after it
				if (pend0 < 0) pend0 = ~item;
				else pend1 = ~item;
				item = 0;
				if (pend1 < 0 && sp > 0) {
					sp--;
					item = stk[sp][tid];
					if (item < 0) {
						pend1 = ~item;
						item = 0;
					}
				}
			}
			if (item > 0) cur = item;
		}
		if (busy && cur < 0 && pend0 < 0 && sp == 0) {
			int id = found ? (ANYHIT ? 0 : best.rank) : -1;
			hits[j] = make_float4(__int_as_float(id), best.t, best.alpha, best.beta);
			busy = false;
		}
	}
	if (STATS) {
		for (int off = 32; off > 0; off >>= 1) {
			boxTests += __shfl_down(boxTests, off, 64);
			triTests += __shfl_down(triTests, off, 64);
			nodeVisits += __shfl_down(nodeVisits, off, 64);
			leafFetches += __shfl_down(leafFetches, off, 64);
		}
		if (lane == 0) {
			atomicAdd(&stats[0], (unsigned long long)boxTests);
			atomicAdd(&stats[1], (unsigned long long)triTests);
			atomicAdd(&stats[4], (unsigned long long)nodeVisits);
			atomicAdd(&stats[5], (unsigned long long)leafFetches);
			atomicAdd(&stats[2], (unsigned long long)walkIters);
			atomicAdd(&stats[3], (unsigned long long)walkLanes);
			atomicAdd(&stats[6], (unsigned long long)leafIters);
			atomicAdd(&stats[7], (unsigned long long)leafLanes);
		}
		for (int off = 32; off > 0; off >>= 1) overflows += __shfl_down(overflows, off, 64);
		if (lane == 0) {
			atomicAdd(&stats[8], (unsigned long long)overflows);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Shade.  One thread per ray of level `level`.  Writes the path-node record of this ray:
//   A[j]  leaf: its radiance.   interior: triangle colour (replaced by the node's radiance in k_gather)
//   B[j]  (first child slot, kind | flags, scalar as float bits, -)   scalar = |n.d| (dielectric) or
//         the refracted proportion (transparent)
//   C[j]  texel of a textured dielectric
// and appends the node's child rays to the next level's queue.  cnt[0] receives the queue
// length, cnt[1] dead slots (refraction under total internal reflection), cnt[2] shaded hits,
// cnt[3] texel fetches.

__device__ __forceinline__ void scatter_offset(const DevScene& sc, uint32_t key, uint32_t lobe, uint32_t k, float& ox,
											   float& oy, float& oz) {
	uint32_t phi = rt_draw(key, lobe, k, 0) & 0x1FFu;
	uint32_t theta = (rt_draw(key, lobe, k, 1) & 0x1FFu) >> 1;
	float cp = sc.cosT[phi], sp = sc.sinT[phi], ct = sc.cosT[theta], st = sc.sinT[theta];
	// horizontal * sin(theta) + cos(theta) * UnitZ, lane by lane (src/Triangle.cpp:107-113)
	ox = cp * st + ct * 0.0f;
	oy = sp * st + ct * 0.0f;
	oz = 0.0f * st + ct * 1.0f;
}

constexpr int SHADE_THREADS = 1024;  // 16 waves share one queue-tail atomic

__global__ __launch_bounds__(SHADE_THREADS) void k_shade(DevScene sc, int level, const float4* __restrict__ rayO,
											   const float4* __restrict__ rayD, const float4* __restrict__ hits, int n,
											   float4* __restrict__ A, int4* __restrict__ B, float4* __restrict__ C,
											   float4* __restrict__ nRayO, float4* __restrict__ nRayD,
											   unsigned* __restrict__ cnt, unsigned nextCap, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	__shared__ unsigned s_tot[SHADE_THREADS / 64];
	__shared__ unsigned s_base[SHADE_THREADS / 64];
	__shared__ unsigned s_stat[3][SHADE_THREADS / 64];
	// The grid does not grow with the queue: a block walks over it in steps of the grid's width (every thread
	// of a block makes the same number of rounds, so the barriers below are uniform).
	for (long long base = (long long)blockIdx.x * SHADE_THREADS; base < n; base += (long long)gridDim.x * SHADE_THREADS) {
		const int j = (int)base + (int)threadIdx.x;
		const int S = sc.nSpec, Dn = sc.nDiff;
		int kind = KIND_LEAF;
		int flags = 0;
		int nchild = 0;
		float3 val = make_float3(0.0f, 0.0f, 0.0f);
		float scalar = 0.0f;
		float3 texel = make_float3(0.0f, 0.0f, 0.0f);
		// state needed after the allocation
		float hx = 0, hy = 0, hz = 0, nx = 0, ny = 0, nz = 0, dx = 0, dy = 0, dz = 0, rough = 0, ior = 1;
		uint32_t key = 0;
		unsigned dead = 0, shaded = 0, fetches = 0;
		const bool active = j < n;
		if (active) {
			float4 O = rayO[j];
			float4 D = rayD[j];
			float4 H = hits[j];
			int rank = __float_as_int(H.x);
			key = __float_as_uint(O.w);
			dx = D.x; dy = D.y; dz = D.z;
			if (rank == -2) {
				dead = 1;
			}
			else if (rank == -1) {
				if (sc.hasSky) {
					val = sky_fetch(sc, dx, dy, dz);
					fetches = 1;
				}
				else
					val = make_float3(sc.bg.x, sc.bg.y, sc.bg.z);
			}
			else if (level < sc.maxDepth) {
				// (at level == maxDepth every hit returns zero, src/RayTracer.cpp:170; the any-hit
				//  traversal reports such hits with rank 0, which is never dereferenced)
				// one aligned 64-byte record serves every hit; the texture coordinates live apart
				const float4* T = sc.triShade + (size_t)rank * 4;
				float4 s0 = T[0], s1 = T[1], s2 = T[2], s3 = T[3];
				const float t = H.y, alpha = H.z, beta = H.w;
				const int triFlags = __float_as_int(s3.w);
				const int texId = __float_as_int(s3.z);
				hx = O.x + t * dx;
				hy = O.y + t * dy;
				hz = O.z + t * dz;
				float g = 1.0f - (alpha + beta);
				nx = (alpha * s0.x + beta * s1.x) + g * s2.x;
				ny = (alpha * s0.y + beta * s1.y) + g * s2.y;
				nz = (alpha * s0.z + beta * s1.z) + g * s2.z;
				float z = dot3w(nx, ny, nz, nx, ny, nz);
				if (z > 0.0f) {
					float len = sqrtf(z);
					nx = nx / len;
					ny = ny / len;
					nz = nz / len;
				}
				float cosine = dot3w(nx, ny, nz, dx, dy, dz);
				const bool transp = (triFlags & TRI_TRANSP) != 0;
				if (!transp && cosine >= 0.0f) {
					// back side: zero
				}
				else if (triFlags & TRI_EMIT) {
					val = make_float3(s0.w, s1.w, s2.w);
				}
				else {
					shaded = 1;
					val = make_float3(s0.w, s1.w, s2.w);  // colour, consumed by k_gather
					rough = s3.x;
					ior = s3.y;
					if (triFlags & TRI_METAL) {
						kind = KIND_METAL;
						nchild = S;
					}
					else if (transp) {
						kind = KIND_TRANSP;
						nchild = S + 1;
					}
					else {
						kind = KIND_DIEL;
						nchild = S + Dn;
						scalar = fabsf(cosine);
						if (texId >= 0) {
							flags |= FLAG_TEX;
							const float4 u01 = sc.triUv[(size_t)rank * 2], u2 = sc.triUv[(size_t)rank * 2 + 1];
							float u = (alpha * u01.x + beta * u01.z) + g * u2.x;
							float v = (alpha * u01.y + beta * u01.w) + g * u2.y;
							texel = tex_fetch(sc, texId, u, v);
							fetches = 1;
						}
					}
				}
			}
		}

		// ---- queue slots for the children.  Inside a wave, per-kind ballots + popcounts give each lane its
		// offset; the 16 wave totals of the block are scanned in LDS and the block claims its range of the
		// next level's queue with a single atomic (one word takes only ~90 atomics per microsecond, so one
		// atomic per wave would bound the kernel).
		const unsigned lane = threadIdx.x & 63;
		const unsigned wave = threadIdx.x >> 6;
		const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
		unsigned long long bm = __ballot(kind == KIND_METAL);
		unsigned long long bt = __ballot(kind == KIND_TRANSP);
		unsigned long long bd = __ballot(kind == KIND_DIEL);
		unsigned offset = (unsigned)__popcll(bm & lt) * (unsigned)S + (unsigned)__popcll(bt & lt) * (unsigned)(S + 1) +
						  (unsigned)__popcll(bd & lt) * (unsigned)(S + Dn);
		unsigned total = (unsigned)__popcll(bm) * (unsigned)S + (unsigned)__popcll(bt) * (unsigned)(S + 1) +
						 (unsigned)__popcll(bd) * (unsigned)(S + Dn);
		unsigned long long bdead = __ballot(dead != 0), bsh = __ballot(shaded != 0), bf = __ballot(fetches != 0);
		if (lane == 0) {
			s_tot[wave] = total;
			s_stat[0][wave] = (unsigned)__popcll(bdead);
			s_stat[1][wave] = (unsigned)__popcll(bsh);
			s_stat[2][wave] = (unsigned)__popcll(bf);
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned run = 0, d = 0, sh = 0, f = 0;
			for (int w = 0; w < SHADE_THREADS / 64; ++w) {
				s_base[w] = run;
				run += s_tot[w];
				d += s_stat[0][w];
				sh += s_stat[1][w];
				f += s_stat[2][w];
			}
			unsigned base = run ? atomicAdd(&cnt[0], run) : 0u;
			for (int w = 0; w < SHADE_THREADS / 64; ++w) s_base[w] += base;
			if (d) atomicAdd(&cnt[1], d);
			if (sh) atomicAdd(&cnt[2], sh);
			if (f) atomicAdd(&cnt[3], f);
		}
		__syncthreads();
		const unsigned childBase = s_base[wave] + offset;

		if (nchild) {
			if (childBase + (unsigned)nchild > nextCap) {
				// cannot happen when the host sized the queue for the worst case; drop the subtree
				kind = KIND_LEAF;
				val = make_float3(0.0f, 0.0f, 0.0f);
				nchild = 0;
				atomicOr(&cnt[4], 1u);
			}
		}
		if (nchild) {
			// Triangle::specular (src/Triangle.cpp:121-146)
			float np = dot3w(dx, dy, dz, nx, ny, nz);
			float k2 = 2.0f * np;
			float rx = dx - k2 * nx, ry = dy - k2 * ny, rz = dz - k2 * nz;
			for (int i = 0; i < S; ++i) {
				float ox, oy, oz;
				scatter_offset(sc, key, 0, (uint32_t)i, ox, oy, oz);
				float cx = rx + ox * rough, cy = ry + oy * rough, cz = rz + oz * rough;
				nRayO[childBase + i] = make_float4(hx, hy, hz, __uint_as_float(rt_child_key(key, (uint32_t)i)));
				nRayD[childBase + i] = make_float4(cx, cy, cz, 0.0f);
			}
			if (kind == KIND_DIEL) {
				// Triangle::diffuse (src/Triangle.cpp:95-119)
				bool keep = dot3w(dx, dy, dz, nx, ny, nz) < 0.0f;
				float tx = keep ? nx : -nx, ty = keep ? ny : -ny, tz = keep ? nz : -nz;
				for (int i = 0; i < Dn; ++i) {
					float ox, oy, oz;
					scatter_offset(sc, key, 1, (uint32_t)i, ox, oy, oz);
					nRayO[childBase + S + i] = make_float4(hx, hy, hz, __uint_as_float(rt_child_key(key, (uint32_t)(S + i))));
					nRayD[childBase + S + i] = make_float4(tx + ox, ty + oy, tz + oz, 0.0f);
				}
			}
			else if (kind == KIND_TRANSP) {
				// Triangle::refract (src/Triangle.cpp:149-179)
				float dot = dot3w(dx, dy, dz, nx, ny, nz);
				float indexRatio = dot > 0.0f ? ior : (1.0f / ior);
				float tx = dot > 0.0f ? -nx : nx, ty = dot > 0.0f ? -ny : ny, tz = dot > 0.0f ? -nz : nz;
				float cosineTheta = fabsf(dot);
				float squareSine1 = 1.0f - cosineTheta * cosineTheta;
				float squareSine2 = (indexRatio * indexRatio) * squareSine1;
				float4 ro = make_float4(hx, hy, hz, __uint_as_float(rt_child_key(key, (uint32_t)S)));
				if (squareSine2 > 1.0f) {
					// total internal reflection: proportion 0, child not traced
					flags |= FLAG_TIR;
					scalar = 0.0f;
					nRayO[childBase + S] = ro;
					nRayD[childBase + S] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
				}
				else {
					float cosine2 = sqrtf(1.0f - squareSine2);
					float ex = indexRatio * (dx + cosineTheta * tx) - tx * cosine2;
					float ey = indexRatio * (dy + cosineTheta * ty) - ty * cosine2;
					float ez = indexRatio * (dz + cosineTheta * tz) - tz * cosine2;
					float temp = (1.0f - indexRatio) / (1.0f + indexRatio);
					float r0 = temp * temp;
					float squareTemp = (1.0f - cosineTheta) * (1.0f - cosineTheta);
					float pow5 = squareTemp * squareTemp * (1.0f - cosineTheta);
					float reflectProportion = r0 + (1.0f - r0) * pow5;
					scalar = 1.0f - reflectProportion;
					nRayO[childBase + S] = ro;
					nRayD[childBase + S] = make_float4(ex, ey, ez, 0.0f);
				}
			}
		}
		if (active) {
			A[j] = make_float4(val.x, val.y, val.z, 0.0f);
			B[j] = make_int4((int)childBase, kind | flags, __float_as_int(scalar), 0);
			if (flags & FLAG_TEX) C[j] = make_float4(texel.x, texel.y, texel.z, 0.0f);
		}
		__syncthreads();  // the LDS words are reused by the next round
	}
}

// The rays of the last recursion level end their paths whatever they hit: a hit returns zero
// (src/RayTracer.cpp:170), a miss the sky or the background (:148-152).  Only A is written (B and C of the
// last level are never read).  Blocks stride over the queue so that each adds its counts once.
__global__ __launch_bounds__(256) void k_shade_last(DevScene sc, const float4* __restrict__ rayD, const float4* __restrict__ hits,
													int n, float4* __restrict__ A, unsigned* __restrict__ cnt, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	unsigned dead = 0, fetches = 0;
	for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
		const int rank = __float_as_int(hits[j].x);
		float3 val = make_float3(0.0f, 0.0f, 0.0f);
		if (rank == -2) dead++;
		else if (rank == -1) {
			if (sc.hasSky) {
				const float4 D = rayD[j];
				val = sky_fetch(sc, D.x, D.y, D.z);
				fetches++;
			}
			else
				val = make_float3(sc.bg.x, sc.bg.y, sc.bg.z);
		}
		A[j] = make_float4(val.x, val.y, val.z, 0.0f);
	}
	for (int off = 32; off > 0; off >>= 1) {
		dead += __shfl_down(dead, off, 64);
		fetches += __shfl_down(fetches, off, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		if (dead) atomicAdd(&cnt[1], dead);
		if (fetches) atomicAdd(&cnt[3], fetches);
	}
}

// Combine step of RayTracer::color for the nodes of one level, children read from the next
// level (src/RayTracer.cpp:177-224).  Sums start from zero and run in child order, scale
// factors are applied after the sum, exactly as the reference does.
__global__ __launch_bounds__(256) void k_gather(int S, int Dn, int n, float4* __restrict__ A,
												const int4* __restrict__ B, const float4* __restrict__ C,
												const float4* __restrict__ childA, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
		int4 b = B[j];
		const int kind = b.y & 0xFF;
		if (kind == KIND_LEAF) continue;
		const float4 col = A[j];
		const float4* ch = childA + (unsigned)b.x;
		float sx = 0.0f, sy = 0.0f, sz = 0.0f;
		for (int i = 0; i < S; ++i) {
			float4 c = ch[i];
			sx = sx + c.x;
			sy = sy + c.y;
			sz = sz + c.z;
		}
		const float fS = (float)S;
		float ox, oy, oz;
		if (kind == KIND_METAL) {
			ox = (sx * col.x) / fS;
			oy = (sy * col.y) / fS;
			oz = (sz * col.z) / fS;
		}
		else if (kind == KIND_TRANSP) {
			sx = sx / fS;
			sy = sy / fS;
			sz = sz / fS;
			const float rp = __int_as_float(b.z);
			float4 rc = (b.y & FLAG_TIR) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : ch[S];
			const float q = 1.0f - rp;
			ox = rp * rc.x + q * sx;
			oy = rp * rc.y + q * sy;
			oz = rp * rc.z + q * sz;
		}
		else {
			const float w = 0.04f / fS;
			sx = sx * w;
			sy = sy * w;
			sz = sz * w;
			float dx = 0.0f, dy = 0.0f, dz = 0.0f;
			for (int i = 0; i < Dn; ++i) {
				float4 c = ch[S + i];
				dx = dx + c.x;
				dy = dy + c.y;
				dz = dz + c.z;
			}
			const float fD = (float)Dn;
			dx = (dx * col.x) / fD;
			dy = (dy * col.y) / fD;
			dz = (dz * col.z) / fD;
			const float ad = __int_as_float(b.z);
			ox = sx + dx * ad;
			oy = sy + dy * ad;
			oz = sz + dz * ad;
			if (b.y & FLAG_TEX) {
				float4 tx = C[j];
				ox = ox * tx.x;
				oy = oy * tx.y;
				oz = oz * tx.z;
			}
		}
		A[j] = make_float4(ox, oy, oz, 0.0f);
	}
}

// accumulateImg(row,col) += (c0+c1+c2+c3) * 0.25f (src/RayTracer.cpp:242-246).  One thread per pixel.
// True when some level's child queue was too small for this batch (slot 4 of the counter blocks): the batch's
// result is then incomplete, is not stored, and the host renders the batch again with worst-case queues.
__device__ __forceinline__ bool batch_overflowed(const unsigned* __restrict__ counts, int maxDepth) {
	unsigned f = 0;
	for (int lv = 1; lv <= maxDepth + 1; ++lv) f |= counts[lv * 8 + 4];
	return f != 0;
}

__global__ __launch_bounds__(256) void k_accumulate(DevScene sc, Batch b, const float4* __restrict__ A,
													float4* __restrict__ accum, const unsigned* __restrict__ counts) {
	if (batch_overflowed(counts, sc.maxDepth)) return;
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p * 4 >= b.nSamples) return;
	int x, y, sub;
	if (!sample_to_pixel(sc, b, p * 4, x, y, sub)) return;
	float tx = 0.0f, ty = 0.0f, tz = 0.0f;
	for (int k = 0; k < 4; ++k) {
		float4 c = A[p * 4 + k];
		tx = tx + c.x;
		ty = ty + c.y;
		tz = tz + c.z;
	}
	float4* dst = accum + ((size_t)y * sc.width + x);
	float4 a = *dst;
	a.x = a.x + tx * 0.25f;
	a.y = a.y + ty * 0.25f;
	a.z = a.z + tz * 0.25f;
	a.w = a.w + 0.0f * 0.25f;
	*dst = a;
}

// out8 = clamp(lroundf(powf(acc / frame, 1 / 2.2f) * 255)) (src/RayTracer.cpp:248-257) as a table search: the
// bytes of the step function were tabulated with the host's own libm, `bits` are the float bit patterns at
// which it steps (ascending, bits[0] = 0).  One thread per four pixels = three packed words of output.
constexpr int TONE_MAX = 1024;
__global__ __launch_bounds__(256) void k_tonemap(const float4* __restrict__ accum, int npix, float frame,
												 const unsigned* __restrict__ bits, const unsigned char* __restrict__ values, int n,
												 unsigned negative, unsigned nan, unsigned char* __restrict__ rgb8) {
	__shared__ unsigned sBits[TONE_MAX];
	__shared__ unsigned char sVal[TONE_MAX];
	for (int i = threadIdx.x; i < n; i += 256) {
		sBits[i] = bits[i];
		sVal[i] = values[i];
	}
	__syncthreads();
	const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
	if (p0 >= npix) return;
	unsigned char out[12];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		float4 a = p0 + k < npix ? accum[p0 + k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		const float ch[3] = {a.x / frame, a.y / frame, a.z / frame};
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const unsigned b = __float_as_uint(ch[c]);
			unsigned v;
			if (ch[c] != ch[c]) v = nan;
			else if (b >> 31) v = negative;
			else {
				int lo = 0, hi = n;  // last i with sBits[i] <= b
				while (hi - lo > 1) {
					const int mid = (lo + hi) >> 1;
					if (sBits[mid] <= b) lo = mid;
					else hi = mid;
				}
				v = sVal[lo];
			}
			out[k * 3 + c] = (unsigned char)v;
		}
	}
	if (p0 + 4 <= npix) {
		unsigned w[3];
#pragma unroll
		for (int i = 0; i < 3; ++i)
			w[i] = (unsigned)out[i * 4] | ((unsigned)out[i * 4 + 1] << 8) | ((unsigned)out[i * 4 + 2] << 16) | ((unsigned)out[i * 4 + 3] << 24);
		unsigned* dst = (unsigned*)(rgb8 + (size_t)p0 * 3);  // p0 is a multiple of 4: 12-byte stride, 4-byte aligned
		dst[0] = w[0];
		dst[1] = w[1];
		dst[2] = w[2];
	}
	else {
		for (int i = 0; i < (npix - p0) * 3; ++i) rgb8[(size_t)p0 * 3 + i] = out[i];
	}
}

// Per-sample radiance export (rtx_sample_radiance): out[(y*W+x)*4+sub] = A[j].
__global__ __launch_bounds__(256) void k_export_samples(DevScene sc, Batch b, const float4* __restrict__ A,
														float4* __restrict__ out, const unsigned* __restrict__ counts) {
	if (batch_overflowed(counts, sc.maxDepth)) return;
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= b.nSamples) return;
	int x, y, sub;
	if (!sample_to_pixel(sc, b, j, x, y, sub)) return;
	out[((size_t)y * sc.width + x) * 4 + sub] = A[j];
}

// Primary-hit export: rank -> reference triangle index.
__global__ __launch_bounds__(256) void k_export_hits(DevScene sc, Batch b, const float4* __restrict__ hits,
													 int* __restrict__ ids, float* __restrict__ tab) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= b.nSamples) return;
	int x, y, sub;
	if (!sample_to_pixel(sc, b, j, x, y, sub)) return;
	float4 h = hits[j];
	int rank = __float_as_int(h.x);
	size_t o = ((size_t)y * sc.width + x) * 4 + sub;
	ids[o] = rank >= 0 ? __float_as_int(sc.triHit[(size_t)rank * 4 + 3].x) : -1;
	if (tab) {
		tab[o * 3 + 0] = h.y;
		tab[o * 3 + 1] = h.z;
		tab[o * 3 + 2] = h.w;
	}
}

__global__ __launch_bounds__(256) void k_export_hits_linear(DevScene sc, int n, const float4* __restrict__ hits,
															int* __restrict__ ids, float* __restrict__ tab) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	float4 h = hits[j];
	int rank = __float_as_int(h.x);
	ids[j] = rank >= 0 ? __float_as_int(sc.triHit[(size_t)rank * 4 + 3].x) : -1;
	if (tab) {
		tab[(size_t)j * 3 + 0] = h.y;
		tab[(size_t)j * 3 + 1] = h.z;
		tab[(size_t)j * 3 + 2] = h.w;
	}
}

// ------------------------------------------------------------------------------------------------
// Host side of the library.

struct HostTri {  // include/RayTracer/Triangle.h:6-34
	float v[3][4];
	float n[3][4];
	float pn[4];
	float color[4];
	float uv[3][2];
	float rough, ior;
	int32_t tex;
	uint8_t emit, metal, transp, pad0;
	uint8_t pad1[8];
};
static_assert(sizeof(HostTri) == RTX_TRIANGLE_BYTES, "triangle record");

struct HostNode {  // include/RayTracer/BVH.h:35-43
	float bmin[4], bmax[4];
	int32_t left, right, tri, pad;
};
static_assert(sizeof(HostNode) == RTX_NODE_BYTES, "node record");

struct TexInfo {
	size_t offset;
	int w, h;
};

template <typename T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	hipError_t alloc(size_t count) {
		release();
		n = count;
		if (!count) return hipSuccess;
		return hipMalloc((void**)&p, count * sizeof(T));
	}
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		n = 0;
	}
};

struct EventPair {
	hipEvent_t a, b;
	int kind;   // 0 trace, 1 shade/gather/accumulate, 2 whole call
	int level;  // path level of a trace launch
};

}  // namespace

struct rtx_scene {
	int device = 0;
	int numCUs = 256;
	int traceRefill = TRACE_REFILL, traceChunk = TRACE_CHUNK, traceBlocksPerCU = 8, traceTriBatch = 24;  // tunables (RTX_TRACE_* env)
	hipStream_t stream = nullptr;
	hipStream_t ownStream = nullptr;
	// parameters
	bool haveParams = false, haveCamera = false, haveTris = false, haveBvh = false, haveTables = false;
	int width = 0, height = 0, maxDepth = 1, nDiff = 1, nSpec = 1;
	float bg[3] = {0, 0, 0};
	uint64_t seed = 0;
	float cam[16] = {0};
	bool hasSky = false;
	float skyBrightness = 1.0f;
	int skyTex[6] = {-1, -1, -1, -1, -1, -1};
	int tileFirst = 0, tileStride = 1, tileCount = -1;
	uint64_t arenaLimit = 0;
	bool instrumented = false;
	// host copies
	std::vector<HostTri> tris;
	std::vector<uint8_t> texels;
	std::vector<TexInfo> texInfo;
	std::vector<int> rankOf;  // triangle index -> candidate rank
	float sceneAbsMax = 0.0f;
	int treeDepth = 0;
	int stackNeed = 32;  // deepest interior node: the traversal pushes at most once per interior level
	int wideStackNeed = 52;  // wide traversal: entries waiting when every child of every node on a path is hit
	int wideStackCap = 24;   // RTX_WIDE_STACK: 16 or 24
	// largest number of rays per primary sample seen at each level so far (0: nothing seen yet)
	double popRatio[17] = {0};
	bool haveHistory = false;
	bool worstCaseQueues = false;  // RTX_WORST_CASE_QUEUES=1: never size queues from history
	bool wideOk = false;     // the tree satisfies what k_trace_wide relies on
	bool forceBinary = false;  // RTX_TRACE_BINARY=1: binary-tree kernel only (for comparisons)
	uint32_t numWide = 0;
	uint32_t numPairs = 0;
	// device
	DevBuf<float4> dPairs, dWide, dTriHit, dTriShade, dTriUv;
	DevBuf<unsigned> dToneBits;
	DevBuf<unsigned char> dToneVal, dRgb8;
	int toneN = 0;
	unsigned toneNeg = 0, toneNan = 0;
	DevBuf<uint8_t> dTexels;
	DevBuf<int4> dTexDesc;
	DevBuf<float> dTables;  // cos[512] sin[512] gamma[256]
	DevBuf<float4> dAccumOwn;
	float4* dAccum = nullptr;  // own or external
	bool texDirty = true;
	DevBuf<uint8_t> arena;
	DevBuf<unsigned> dCounts;           // 8 per level
	DevBuf<unsigned long long> dStats;  // node tests, tri tests
	unsigned* hCounts = nullptr;        // pinned
	unsigned long long* hStats = nullptr;
	std::vector<EventPair> events;
	rtx_stats stats;

	~rtx_scene() {
		(void)hipSetDevice(device);
		dPairs.release(); dWide.release(); dTriHit.release(); dTriShade.release(); dTriUv.release(); dToneBits.release(); dToneVal.release(); dRgb8.release(); dTexels.release(); dTexDesc.release();
		dTables.release(); dAccumOwn.release(); arena.release(); dCounts.release(); dStats.release();
		if (hCounts) (void)hipHostFree(hCounts);
		if (hStats) (void)hipHostFree(hStats);
		for (auto& e : events) {
			(void)hipEventDestroy(e.a);
			(void)hipEventDestroy(e.b);
		}
		if (ownStream) (void)hipStreamDestroy(ownStream);
	}
};

namespace {

constexpr int MAX_DEPTH = 15;
constexpr int CNT_STRIDE = 8;   // per level: 0 queue length, 1 dead, 2 shaded, 3 texel fetches, 4 overflow flag, 5-7 trace cursors

int num_tiles(const rtx_scene* s) {
	int tx = (s->width + TILE - 1) / TILE, ty = (s->height + TILE - 1) / TILE;
	return tx * ty;
}

// Number of tiles in this scene's share: first, first+stride, ... below the tile total.
int tile_slots(const rtx_scene* s) {
	int total = num_tiles(s);
	if (s->tileFirst >= total) return 0;
	int avail = (total - s->tileFirst + s->tileStride - 1) / s->tileStride;
	if (s->tileCount >= 0 && s->tileCount < avail) avail = s->tileCount;
	return avail;
}

DevScene make_dev_scene(const rtx_scene* s) {
	DevScene d;
	std::memset(&d, 0, sizeof(d));
	d.pairs = s->dPairs.p;
	d.wide = s->dWide.p;
	d.triHit = s->dTriHit.p;
	d.triShade = s->dTriShade.p;
	d.triUv = s->dTriUv.p;
	d.texels = s->dTexels.p;
	d.texDesc = s->dTexDesc.p;
	d.cosT = s->dTables.p;
	d.sinT = s->dTables.p + 512;
	d.gammaT = s->dTables.p + 1024;
	d.camO = make_float4(s->cam[0], s->cam[1], s->cam[2], s->cam[3]);
	d.camLUC = make_float4(s->cam[4], s->cam[5], s->cam[6], s->cam[7]);
	d.camRS = make_float4(s->cam[8], s->cam[9], s->cam[10], s->cam[11]);
	d.camDS = make_float4(s->cam[12], s->cam[13], s->cam[14], s->cam[15]);
	d.bg = make_float4(s->bg[0], s->bg[1], s->bg[2], 0.0f);
	d.width = s->width;
	d.height = s->height;
	d.maxDepth = s->maxDepth;
	d.nDiff = s->nDiff;
	d.nSpec = s->nSpec;
	d.hasSky = s->hasSky ? 1 : 0;
	d.skyBrightness = s->skyBrightness;
	for (int i = 0; i < 6; ++i) d.skyTex[i] = s->skyTex[i];
	// same folding of the 64-bit seed as the CPU side of the RNG specification
	auto mix = [](uint32_t h, uint32_t v) {
		h = (h + 0x7F4A7C15u) ^ (v * 0x9E3779B1u);
		h ^= h >> 16;
		h *= 0x85EBCA6Bu;
		h ^= h >> 13;
		h *= 0xC2B2AE35u;
		h ^= h >> 16;
		return h;
	};
	d.seedKey = mix((uint32_t)s->seed, (uint32_t)(s->seed >> 32));
	d.sceneAbsMax = s->sceneAbsMax;
	d.tilesX = (s->width + TILE - 1) / TILE;
	return d;
}

int upload_textures(rtx_scene* s) {
	if (!s->texDirty) return RTX_OK;
	HIPCHK(s->dTexels.alloc(std::max<size_t>(s->texels.size(), 16)));
	if (!s->texels.empty())
		HIPCHK(hipMemcpy(s->dTexels.p, s->texels.data(), s->texels.size(), hipMemcpyHostToDevice));
	std::vector<int4> desc(std::max<size_t>(s->texInfo.size(), 1));
	for (size_t i = 0; i < s->texInfo.size(); ++i) {
		if (s->texInfo[i].offset > 0x7FFFFFFFull) return fail(RTX_ERR_ARG, "texture data beyond 2 GiB");
		desc[i] = make_int4((int)s->texInfo[i].offset, s->texInfo[i].w, s->texInfo[i].h, 0);
	}
	HIPCHK(s->dTexDesc.alloc(desc.size()));
	HIPCHK(hipMemcpy(s->dTexDesc.p, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
	s->texDirty = false;
	return RTX_OK;
}

int ensure_ready(rtx_scene* s, bool needCamera) {
	if (!s->haveParams) return fail(RTX_ERR_STATE, "rtx_scene_set_params has not been called");
	if (needCamera && !s->haveCamera) return fail(RTX_ERR_STATE, "rtx_scene_set_camera has not been called");
	if (!s->haveTris) return fail(RTX_ERR_STATE, "rtx_scene_set_triangles has not been called");
	if (!s->haveBvh) return fail(RTX_ERR_STATE, "rtx_scene_set_bvh has not been called");
	if (!s->haveTables) return fail(RTX_ERR_STATE, "rtx_scene_set_tables has not been called");
	HIPCHK(hipSetDevice(s->device));
	const int ntex = (int)s->texInfo.size();
	for (const HostTri& t : s->tris)
		if (t.tex >= ntex) return fail(RTX_ERR_ARG, "a triangle refers to a texture id that was never added");
	if (s->hasSky)
		for (int i = 0; i < 6; ++i)
			if (s->skyTex[i] < 0 || s->skyTex[i] >= ntex) return fail(RTX_ERR_ARG, "skybox texture id out of range");
	int rc = upload_textures(s);
	if (rc) return rc;
	if (!s->dAccum) {
		HIPCHK(s->dAccumOwn.alloc((size_t)s->width * s->height));
		HIPCHK(hipMemset(s->dAccumOwn.p, 0, (size_t)s->width * s->height * sizeof(float4)));
		s->dAccum = s->dAccumOwn.p;
	}
	if (!s->stream) {
		if (!s->ownStream) HIPCHK(hipStreamCreateWithFlags(&s->ownStream, hipStreamNonBlocking));
		s->stream = s->ownStream;
	}
	if (!s->dCounts.p) {
		HIPCHK(s->dCounts.alloc((MAX_DEPTH + 2) * CNT_STRIDE));
		HIPCHK(s->dStats.alloc(12));
		HIPCHK(hipHostMalloc((void**)&s->hCounts, (MAX_DEPTH + 2) * CNT_STRIDE * sizeof(unsigned), hipHostMallocDefault));
		HIPCHK(hipHostMalloc((void**)&s->hStats, 12 * sizeof(unsigned long long), hipHostMallocDefault));
	}
	return RTX_OK;
}

// Queue bytes.  Persistent per path node: A (16) + B (16) + C (16); transient, two ping-pong sets sized
// for the widest level: origin (16) + direction (16) + hit (16).
constexpr size_t NODE_BYTES = 48, RAY_BYTES = 48;

struct EventScope {
	rtx_scene* s;
	size_t idx;
};

int begin_event(rtx_scene* s, int kind, size_t& idx, int level = 0) {
	EventPair e;
	HIPCHK(hipEventCreate(&e.a));
	HIPCHK(hipEventCreate(&e.b));
	e.kind = kind;
	e.level = level;
	s->events.push_back(e);
	idx = s->events.size() - 1;
	HIPCHK(hipEventRecord(e.a, s->stream));
	return RTX_OK;
}
int end_event(rtx_scene* s, size_t idx) {
	HIPCHK(hipEventRecord(s->events[idx].b, s->stream));
	return RTX_OK;
}
// After a stream sync: fold event durations into the stats and recycle the events.
int harvest_events(rtx_scene* s) {
	for (auto& e : s->events) {
		float ms = 0.0f;
		HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
		if (e.kind == 0) {
			s->stats.trace_ms += ms;
			s->stats.trace_ms_per_level[e.level & 15] += ms;
			s->stats.trace_launches++;
		}
		else if (e.kind == 1)
			s->stats.shade_ms += ms;
		else
			s->stats.total_ms += ms;
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	s->events.clear();
	return RTX_OK;
}

// Slots of a level's counter block (CNT_STRIDE unsigned, zeroed at the start of a pixel batch).
constexpr int WORK_SLOT = 5;     // queue cursor of the wide traversal
constexpr int UNSAFE_SLOT = 6;   // number of rays set aside for the binary-tree kernel
constexpr int WORK2_SLOT = 7;    // queue cursor of the binary-tree kernel

// The binary-tree kernel (every ancestor box evaluated as the reference does).  With a list it
// traces list[0 .. *listCount); the count lives on the device, so the grid is a fixed small one.
template <bool ANYHIT, bool EXH>
int launch_exact_t(rtx_scene* s, const DevScene& d, const float4* ro, const float4* rd, float4* hits, int n, unsigned* work,
				   const int* list, const unsigned* listCount, const unsigned* nDev) {
	const int wanted = list ? s->numCUs : nDev ? s->numCUs * s->traceBlocksPerCU : (n + TRACE_THREADS - 1) / TRACE_THREADS;
	dim3 grid(std::max(1, std::min(wanted, s->numCUs * s->traceBlocksPerCU))), block(TRACE_THREADS);
	const int need = s->stackNeed;
	const bool st = s->instrumented;
	unsigned long long* stats = s->dStats.p;
#define RTX_LAUNCH(STK, ST)                                                                                              \
	hipLaunchKernelGGL((k_trace<STK, ANYHIT, EXH, ST>), grid, block, 0, s->stream, d, ro, rd, hits, n, stats, work,      \
					   s->traceRefill, s->traceChunk, s->traceTriBatch, list, listCount, nDev)
	if (need <= 16) {
		if (st) RTX_LAUNCH(16, true); else RTX_LAUNCH(16, false);
	}
	else if (need <= 20) {
		if (st) RTX_LAUNCH(20, true); else RTX_LAUNCH(20, false);
	}
	else if (need <= 24) {
		if (st) RTX_LAUNCH(24, true); else RTX_LAUNCH(24, false);
	}
	else {
		if (st) RTX_LAUNCH(32, true); else RTX_LAUNCH(32, false);
	}
#undef RTX_LAUNCH
	HIPCHK(hipGetLastError());
	return RTX_OK;
}

template <bool ANYHIT>
int launch_wide_t(rtx_scene* s, const DevScene& d, const float4* ro, const float4* rd, float4* hits, int n, unsigned* work,
				  int* unsafeList, unsigned* unsafeCount, const unsigned* nDev) {
	const int wanted = nDev ? s->numCUs * s->traceBlocksPerCU : (n + TRACE_THREADS - 1) / TRACE_THREADS;
	dim3 grid(std::min(wanted, s->numCUs * s->traceBlocksPerCU)), block(TRACE_THREADS);
	// The stack is no deeper than wideStackCap entries however deep the tree: a ray that would need more
	// moves to the binary-tree kernel.  Fewer entries, more resident waves.
	const int need = std::min(s->wideStackNeed, s->wideStackCap);
	const bool st = s->instrumented;
	unsigned long long* stats = s->dStats.p;
#define RTX_LAUNCH(STK, ST)                                                                                              \
	hipLaunchKernelGGL((k_trace_wide<STK, ANYHIT, ST>), grid, block, 0, s->stream, d, ro, rd, hits, n, stats, work,      \
					   unsafeList, unsafeCount, s->traceRefill, s->traceChunk, s->traceTriBatch, nDev)
	if (need <= 16) {
		if (st) RTX_LAUNCH(16, true); else RTX_LAUNCH(16, false);
	}
	else {
		if (st) RTX_LAUNCH(24, true); else RTX_LAUNCH(24, false);
	}
#undef RTX_LAUNCH
	HIPCHK(hipGetLastError());
	return RTX_OK;
}

// One queue of rays -> hit records.  `cnt` is the level's zeroed counter block, `unsafeList` room for n ints.
// Production: wide traversal, then the binary-tree kernel over the rays the wide one set aside.
// Exhaustive: the binary-tree kernel without culling over the whole queue.
int launch_trace(rtx_scene* s, const DevScene& d, bool anyhit, bool exhaustive, const float4* ro, const float4* rd,
				 float4* hits, int n, unsigned* cnt, int* unsafeList, int level = 0, const unsigned* nDev = nullptr) {
	if (n <= 0 && !nDev) return RTX_OK;
	size_t ev;
	int rc = begin_event(s, 0, ev, level);
	if (rc) return rc;
	if (exhaustive)
		rc = launch_exact_t<false, true>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, nullptr, nullptr, nDev);
	else if (!s->wideOk || s->forceBinary)
		rc = anyhit ? launch_exact_t<true, false>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, nullptr, nullptr, nDev)
					: launch_exact_t<false, false>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, nullptr, nullptr, nDev);
	else if (anyhit) {
		rc = launch_wide_t<true>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, unsafeList, cnt + UNSAFE_SLOT, nDev);
		if (rc == RTX_OK)
			rc = launch_exact_t<true, false>(s, d, ro, rd, hits, n, cnt + WORK2_SLOT, unsafeList, cnt + UNSAFE_SLOT, nullptr);
	}
	else {
		rc = launch_wide_t<false>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, unsafeList, cnt + UNSAFE_SLOT, nDev);
		if (rc == RTX_OK)
			rc = launch_exact_t<false, false>(s, d, ro, rd, hits, n, cnt + WORK2_SLOT, unsafeList, cnt + UNSAFE_SLOT, nullptr);
	}
	if (rc) return rc;
	return end_event(s, ev);
}

int ensure_arena(rtx_scene* s, size_t want) {
	if (s->arena.n >= want) return RTX_OK;
	s->arena.release();
	hipError_t e = s->arena.alloc(want);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return fail(RTX_ERR_MEMORY, "cannot allocate the work arena (" + std::to_string(want >> 20) + " MiB)");
	}
	return RTX_OK;
}

struct Bump {
	uint8_t* base;
	size_t cap, lo, hi;  // persistent grows from lo, transient sets are carved at fixed offsets from hi
	void* take(size_t bytes) {
		bytes = (bytes + 255) & ~(size_t)255;
		if (lo + bytes > hi) return nullptr;
		void* p = base + lo;
		lo += bytes;
		return p;
	}
};

// Rays per primary sample each level's queue is sized for.  Worst case: every node spawns its maximum at every
// level.  With history: 1.3 times the most seen so far plus a little, capped by the worst case; a batch that
// outgrows that is detected on the device, left out of the image and rendered again with worst-case queues.
void level_ratios(const rtx_scene* s, bool useHistory, int levels, double* r) {
	const double bmax = (double)std::max(s->nSpec + s->nDiff, s->nSpec + 1);
	double worst = 1.0;
	for (int lv = 0; lv <= levels; ++lv) {
		r[lv] = (useHistory && lv > 0) ? std::min(worst, s->popRatio[lv] * 1.3 + 1.0 / 64.0) : worst;
		worst *= bmax;
	}
}

struct RenderTarget {
	int mode;      // 0 = accumulate, 1 = export per-sample radiance to dOut, 2 = primary hits to dIds/dTab
	int hitMode;   // mode 2: 0 culled / 1 exhaustive
	float4* dOut;
	int* dIds;
	float* dTab;
};

// Renders the 4x4-pixel blocks [gBegin, gEnd) of this scene's tile share in as few pixel batches as the arena allows.
int render_range(rtx_scene* s, const DevScene& d, uint32_t frame, const RenderTarget& tg, long long gBegin, long long gEnd,
				 bool useHistory) {
	const int mode = tg.mode;
	const int levels = (mode == 2) ? 0 : s->maxDepth;
	double ratio[MAX_DEPTH + 2];
	level_ratios(s, useHistory, levels, ratio);
	double sumRatio = 0.0, maxRatio = 0.0;
	for (int lv = 0; lv <= levels; ++lv) {
		sumRatio += ratio[lv];
		maxRatio = std::max(maxRatio, ratio[lv]);
	}
	// per sample: A, B, C of every level; two ping-pong ray sets and the set-aside list sized for the widest level
	const double perSample = (mode == 2) ? (double)(NODE_BYTES + 2 * RAY_BYTES + 4) : sumRatio * NODE_BYTES + maxRatio * (2.0 * RAY_BYTES + 4.0);
	const double fixedBytes = (double)(MAX_DEPTH + 8) * 4096 + (double)(levels + 1) * 1024.0 * NODE_BYTES + 1024.0 * (2.0 * RAY_BYTES + 4.0) + 65536.0;
	const long long totalBlocks = gEnd - gBegin;

	size_t freeB = 0, totalB = 0;
	HIPCHK(hipMemGetInfo(&freeB, &totalB));
	// Default budget: 60 % of what is free, at most 64 GiB.  Each batch costs one synchronisation plus, per
	// level, the latency of its longest ray, so bigger batches are better; allocating and mapping hundreds of GB
	// costs seconds, so the arena is not grown beyond that.
	size_t budget = s->arenaLimit ? (size_t)s->arenaLimit
								  : std::min<size_t>((size_t)((double)(freeB + s->arena.n) * 0.6), (size_t)64 << 30);
	const double needAll = perSample * (double)totalBlocks * SAMPLES_PER_BLOCK + fixedBytes;
	size_t want = (size_t)std::min((double)budget, needAll);
	// an arena that is at least nearly as big as wanted is used as it is rather than reallocated
	if (s->arena.n && (double)s->arena.n >= 0.9 * (double)want) want = std::max(want, s->arena.n);
	if (s->arenaLimit && want > (size_t)s->arenaLimit) want = (size_t)s->arenaLimit;
	long long blocksPerBatch = (long long)(((double)want - fixedBytes) / (perSample * SAMPLES_PER_BLOCK));
	if (blocksPerBatch < 1) {
		blocksPerBatch = 1;
		want = (size_t)(perSample * SAMPLES_PER_BLOCK + fixedBytes);
	}
	if (blocksPerBatch > totalBlocks) blocksPerBatch = totalBlocks;
	// keep ray indices inside int32
	const long long maxBlocksInt = (long long)(0x7FFFF000ll / SAMPLES_PER_BLOCK);
	if ((double)blocksPerBatch * maxRatio > (double)maxBlocksInt) blocksPerBatch = std::max<long long>(1, (long long)((double)maxBlocksInt / maxRatio));
	int rc = ensure_arena(s, want);
	if (rc) return rc;

	for (long long g0 = gBegin; g0 < gEnd; g0 += blocksPerBatch) {
		const long long nb = std::min(blocksPerBatch, gEnd - g0);
		Batch b;
		b.g0 = (int)g0;
		b.nSamples = (int)(nb * SAMPLES_PER_BLOCK);
		b.tileFirst = s->tileFirst;
		b.tileStride = s->tileStride;
		s->stats.chunks++;

		// capacities of the levels of this batch
		size_t cap[MAX_DEPTH + 2];
		size_t widestCap = (size_t)b.nSamples;
		{
			size_t worst = (size_t)b.nSamples;
			const size_t bmax = (size_t)std::max(s->nSpec + s->nDiff, s->nSpec + 1);
			for (int lv = 0; lv <= levels; ++lv) {
				cap[lv] = lv == 0 ? worst : std::min(worst, (size_t)std::ceil((double)b.nSamples * ratio[lv]) + 1024);
				widestCap = std::max(widestCap, cap[lv]);
				worst *= bmax;
			}
			cap[levels + 1] = 0;
		}

		// arena carve-up: transient ray sets at the top, node arrays bump-allocated from the bottom
		const size_t setBytes = ((widestCap * 16 + 255) & ~(size_t)255);
		const size_t listBytes = ((widestCap * 4 + 255) & ~(size_t)255);
		Bump bump{s->arena.p, s->arena.n, 0, s->arena.n};
		if (6 * setBytes + listBytes > bump.cap) return fail(RTX_ERR_MEMORY, "work arena too small for one pixel block");
		bump.hi = bump.cap - 6 * setBytes - listBytes;
		int* unsafeList = (int*)(bump.base + bump.hi + 6 * setBytes);
		float4* rayO[2] = {(float4*)(bump.base + bump.hi), (float4*)(bump.base + bump.hi + 3 * setBytes)};
		float4* rayD[2] = {(float4*)(bump.base + bump.hi + setBytes), (float4*)(bump.base + bump.hi + 4 * setBytes)};
		float4* hit[2] = {(float4*)(bump.base + bump.hi + 2 * setBytes), (float4*)(bump.base + bump.hi + 5 * setBytes)};

		HIPCHK(hipMemsetAsync(s->dCounts.p, 0, (MAX_DEPTH + 2) * CNT_STRIDE * sizeof(unsigned), s->stream));
		size_t ev;
		rc = begin_event(s, 1, ev);
		if (rc) return rc;
		hipLaunchKernelGGL(k_raygen, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, frame, rayO[0], rayD[0]);
		HIPCHK(hipGetLastError());
		rc = end_event(s, ev);
		if (rc) return rc;

		if (mode == 2) {
			rc = launch_trace(s, d, false, tg.hitMode == 1, rayO[0], rayD[0], hit[0], b.nSamples, s->dCounts.p, unsafeList);
			if (rc) return rc;
			hipLaunchKernelGGL(k_export_hits, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, hit[0], tg.dIds, tg.dTab);
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(s->stream));
			continue;
		}

		// Every level is enqueued at once: a level's queue length stays on the device (slot 0 of its counter
		// block, written by the shade kernel of the level above), arrays are carved at the level's capacity
		// and grids are fixed.  The host reads the counters once, after the batch.
		float4* A[MAX_DEPTH + 1];
		int4* B[MAX_DEPTH + 1];
		float4* C[MAX_DEPTH + 1];
		for (int lv = 0; lv <= s->maxDepth; ++lv) {
			const size_t n = cap[lv];
			A[lv] = (float4*)bump.take(n * 16);
			B[lv] = (int4*)bump.take(n * 16);
			C[lv] = (float4*)bump.take(n * 16);
			if (!A[lv] || !B[lv] || !C[lv]) return fail(RTX_ERR_MEMORY, "work arena exhausted (internal sizing error)");
			const int cur = lv & 1, nxt = cur ^ 1;
			const unsigned* nDev = lv ? s->dCounts.p + (size_t)lv * CNT_STRIDE : nullptr;
			rc = launch_trace(s, d, lv == s->maxDepth, false, rayO[cur], rayD[cur], hit[cur], (int)n,
							  s->dCounts.p + (size_t)lv * CNT_STRIDE, unsafeList, lv, nDev);
			if (rc) return rc;
			unsigned* cnt = s->dCounts.p + (size_t)(lv + 1) * CNT_STRIDE;
			rc = begin_event(s, 1, ev);
			if (rc) return rc;
			if (lv == s->maxDepth)
				hipLaunchKernelGGL(k_shade_last, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)s->numCUs * 8)), dim3(256), 0, s->stream,
								   d, rayD[cur], hit[cur], (int)n, A[lv], cnt, nDev);
			else
				hipLaunchKernelGGL(k_shade, dim3((unsigned)std::min<size_t>((n + SHADE_THREADS - 1) / SHADE_THREADS, (size_t)s->numCUs * 4)), dim3(SHADE_THREADS), 0, s->stream, d, lv,
								   rayO[cur], rayD[cur], hit[cur], (int)n, A[lv], B[lv], C[lv], rayO[nxt], rayD[nxt], cnt, (unsigned)cap[lv + 1],
								   nDev);
			HIPCHK(hipGetLastError());
			rc = end_event(s, ev);
			if (rc) return rc;
		}
		rc = begin_event(s, 1, ev);
		if (rc) return rc;
		for (int lv = s->maxDepth - 1; lv >= 0; --lv) {
			const unsigned* nDev = lv ? s->dCounts.p + (size_t)lv * CNT_STRIDE : nullptr;
			hipLaunchKernelGGL(k_gather, dim3((unsigned)std::min<size_t>((cap[lv] + 255) / 256, (size_t)s->numCUs * 16)), dim3(256), 0, s->stream,
							   s->nSpec, s->nDiff, (int)cap[lv], A[lv], B[lv], C[lv], A[lv + 1], nDev);
			HIPCHK(hipGetLastError());
		}
		const int npix = b.nSamples / 4;
		if (mode == 0)
			hipLaunchKernelGGL(k_accumulate, dim3((npix + 255) / 256), dim3(256), 0, s->stream, d, b, A[0], s->dAccum, s->dCounts.p);
		else
			hipLaunchKernelGGL(k_export_samples, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, A[0], tg.dOut, s->dCounts.p);
		HIPCHK(hipGetLastError());
		rc = end_event(s, ev);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(s->hCounts, s->dCounts.p, (MAX_DEPTH + 2) * CNT_STRIDE * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
		HIPCHK(hipStreamSynchronize(s->stream));
		rc = harvest_events(s);
		if (rc) return rc;

		bool overflow = false;
		for (int lv = 0; lv <= s->maxDepth; ++lv) overflow = overflow || s->hCounts[(size_t)(lv + 1) * CNT_STRIDE + 4] != 0;
		// what the levels asked for, whether or not it fitted (the cursor runs on past the capacity)
		for (int lv = 1; lv <= s->maxDepth; ++lv)
			s->popRatio[lv] = std::max(s->popRatio[lv], (double)s->hCounts[(size_t)lv * CNT_STRIDE] / (double)b.nSamples);
		if (overflow) {
			if (!useHistory) return fail(RTX_ERR_MEMORY, "child queue overflow (internal sizing error)");
			// nothing of this batch was stored: once more, with queues that cannot overflow
			s->stats.chunks--;
			s->stats.queue_retries++;
			rc = render_range(s, d, frame, tg, g0, g0 + nb, false);
			if (rc) return rc;
			continue;
		}
		s->haveHistory = true;
		uint64_t n = (uint64_t)b.nSamples;
		for (int lv = 0; lv <= s->maxDepth; ++lv) {
			const unsigned* hc = s->hCounts + (size_t)(lv + 1) * CNT_STRIDE;
			s->stats.rays_per_level[lv] += n - hc[1];
			s->stats.rays += n - hc[1];
			s->stats.shaded_hits += hc[2];
			s->stats.texture_fetches += hc[3];
			n = hc[0];
		}
	}
	HIPCHK(hipStreamSynchronize(s->stream));
	return harvest_events(s);
}

int render_batches(rtx_scene* s, uint32_t frame, int mode, int hitMode, float4* dOut, int* dIds, float* dTab) {
	const DevScene d = make_dev_scene(s);
	const long long totalBlocks = (long long)tile_slots(s) * BLOCKS_PER_TILE;
	if (totalBlocks == 0) return RTX_OK;
	const RenderTarget tg{mode, hitMode, dOut, dIds, dTab};
	return render_range(s, d, frame, tg, 0, totalBlocks, s->haveHistory && !s->worstCaseQueues);
}

int finish_stats(rtx_scene* s) {
	if (s->instrumented) {
		HIPCHK(hipMemcpy(s->hStats, s->dStats.p, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		s->stats.node_tests = s->hStats[0];
		s->stats.tri_tests = s->hStats[1];
		if (std::getenv("RTX_TRACE_DEBUG"))
			std::fprintf(stderr, "[rtx] wide: walk iterations %llu with %.1f lanes, leaf phases %llu with %.1f lanes, %llu stack overflows\n", s->hStats[2],
						 (double)s->hStats[3] / (double)std::max(1ull, s->hStats[2]), s->hStats[6],
						 (double)s->hStats[7] / (double)std::max(1ull, s->hStats[6]), s->hStats[8]);
		s->stats.wide_visits = s->hStats[4];
		s->stats.leaf_fetches = s->hStats[5];
	}
	return RTX_OK;
}

int begin_call(rtx_scene* s) {
	std::memset(&s->stats, 0, sizeof(s->stats));
	if (s->instrumented) HIPCHK(hipMemsetAsync(s->dStats.p, 0, 12 * sizeof(unsigned long long), s->stream));
	return RTX_OK;
}

}  // namespace

// ================================================================================================
// C ABI

RTX_API const char* rtx_last_error(void) { return g_err.c_str(); }

RTX_API int rtx_device_count(void) {
	int n = 0;
	HIPCHK(hipGetDeviceCount(&n));
	return n;
}

RTX_API int rtx_scene_create(rtx_scene** out, int device) {
	if (!out) return fail(RTX_ERR_ARG, "out is null");
	int n = 0;
	HIPCHK(hipGetDeviceCount(&n));
	if (device < 0 || device >= n) return fail(RTX_ERR_ARG, "no such HIP device");
	HIPCHK(hipSetDevice(device));
	rtx_scene* s = new (std::nothrow) rtx_scene();
	if (!s) return fail(RTX_ERR_MEMORY, "out of host memory");
	s->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->numCUs = prop.multiProcessorCount;
	if (const char* e = std::getenv("RTX_TRACE_REFILL")) s->traceRefill = std::max(1, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_TRACE_CHUNK")) s->traceChunk = std::max(64, std::atoi(e));
	if (const char* e = std::getenv("RTX_TRACE_TRI_BATCH")) s->traceTriBatch = std::max(1, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_TRACE_BLOCKS_PER_CU")) s->traceBlocksPerCU = std::max(1, std::atoi(e));
	if (const char* e = std::getenv("RTX_TRACE_BINARY")) s->forceBinary = std::atoi(e) != 0;
	if (const char* e = std::getenv("RTX_WORST_CASE_QUEUES")) s->worstCaseQueues = std::atoi(e) != 0;
	if (const char* e = std::getenv("RTX_WIDE_STACK")) s->wideStackCap = std::max(16, std::min(24, std::atoi(e)));
	std::memset(&s->stats, 0, sizeof(s->stats));
	*out = s;
	return RTX_OK;
}

RTX_API int rtx_scene_destroy(rtx_scene* s) {
	delete s;
	return RTX_OK;
}

RTX_API int rtx_scene_set_stream(rtx_scene* s, void* hip_stream) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	s->stream = (hipStream_t)hip_stream;
	return RTX_OK;
}

RTX_API int rtx_scene_set_params(rtx_scene* s, int width, int height, int max_depth, int n_diffuse, int n_specular,
								 const float background_rgb[3], uint64_t seed) {
	if (!s || !background_rgb) return fail(RTX_ERR_ARG, "null argument");
	if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
		return fail(RTX_ERR_ARG, "image size out of range");
	if (max_depth < 1 || max_depth > MAX_DEPTH) return fail(RTX_ERR_ARG, "max_depth must be in 1..15");
	if (n_diffuse < 1 || n_diffuse > 64 || n_specular < 1 || n_specular > 64)
		return fail(RTX_ERR_ARG, "ray numbers must be in 1..64");
	if (s->haveParams && (width != s->width || height != s->height)) {
		HIPCHK(hipSetDevice(s->device));
		s->dAccumOwn.release();
		if (s->dAccum && s->dAccum == s->dAccumOwn.p) s->dAccum = nullptr;
		s->dAccum = nullptr;
	}
	s->width = width;
	s->height = height;
	s->maxDepth = max_depth;
	s->nDiff = n_diffuse;
	s->nSpec = n_specular;
	for (int i = 0; i < 3; ++i) s->bg[i] = background_rgb[i];
	s->seed = seed;
	s->haveParams = true;
	s->haveHistory = false;
	std::fill(s->popRatio, s->popRatio + 17, 0.0);
	return RTX_OK;
}

RTX_API int rtx_scene_set_camera(rtx_scene* s, const float origin[4], const float left_up_corner[4],
								 const float right_step[4], const float down_step[4]) {
	if (!s || !origin || !left_up_corner || !right_step || !down_step) return fail(RTX_ERR_ARG, "null argument");
	std::memcpy(s->cam + 0, origin, 16);
	std::memcpy(s->cam + 4, left_up_corner, 16);
	std::memcpy(s->cam + 8, right_step, 16);
	std::memcpy(s->cam + 12, down_step, 16);
	s->haveCamera = true;
	return RTX_OK;
}

RTX_API int rtx_scene_set_triangles(rtx_scene* s, const void* triangles, uint32_t n) {
	if (!s || (!triangles && n)) return fail(RTX_ERR_ARG, "null argument");
	if (n > (1u << 30)) return fail(RTX_ERR_ARG, "too many triangles");
	s->tris.resize(n);
	if (n) std::memcpy((void*)s->tris.data(), triangles, (size_t)n * sizeof(HostTri));
	float amax = 0.0f;
	for (const HostTri& t : s->tris) {
		for (int k = 0; k < 3; ++k) {
			if (!(t.v[k][3] == 0.0f) || !(t.n[k][3] == 0.0f))
				return fail(RTX_ERR_ARG, "triangle w lanes must be zero, as the reference constructs them");
			for (int c = 0; c < 3; ++c) {
				float a = std::fabs(t.v[k][c]);
				if (!(a <= FLT_MAX)) return fail(RTX_ERR_ARG, "non-finite vertex position");
				amax = std::max(amax, a);
			}
		}
		if (!(t.pn[3] == 0.0f)) return fail(RTX_ERR_ARG, "triangle w lanes must be zero, as the reference constructs them");
		if (t.tex < -1) return fail(RTX_ERR_ARG, "texture index below -1");
	}
	s->sceneAbsMax = amax;
	s->haveTris = true;
	s->haveBvh = false;
	s->haveHistory = false;
	std::fill(s->popRatio, s->popRatio + 17, 0.0);
	return RTX_OK;
}

RTX_API int rtx_scene_set_bvh(rtx_scene* s, const void* linear_nodes, uint32_t n) {
	if (!s || !linear_nodes) return fail(RTX_ERR_ARG, "null argument");
	if (!s->haveTris) return fail(RTX_ERR_STATE, "rtx_scene_set_triangles must come first");
	if (n == 0) return fail(RTX_ERR_ARG, "a tree has at least its root");
	const HostNode* N = (const HostNode*)linear_nodes;
	const uint32_t ntri = (uint32_t)s->tris.size();

	// Pass 1: the reference's own visiting order with every box hit (src/BVH.cpp:170-189: left
	// pushed first, right popped first) numbers the leaves; that number is the tie-break rank.
	std::vector<int> rankOf(ntri, -1);
	std::vector<int> depthOf(n, 0);
	int maxInteriorDepth = 0;
	std::vector<uint8_t> seen(n, 0);
	std::vector<int> st;
	st.push_back(0);
	int nextRank = 0, maxDepth = 0;
	size_t visited = 0;
	while (!st.empty()) {
		int i = st.back();
		st.pop_back();
		if (seen[i]) return fail(RTX_ERR_ARG, "node reachable twice: not a tree");
		seen[i] = 1;
		++visited;
		const HostNode& nd = N[i];
		maxDepth = std::max(maxDepth, depthOf[i]);
		if (nd.tri >= 0) {
			if ((uint32_t)nd.tri >= ntri) return fail(RTX_ERR_ARG, "leaf refers to a triangle out of range");
			if (rankOf[nd.tri] != -1) return fail(RTX_ERR_ARG, "triangle referenced by two leaves");
			rankOf[nd.tri] = nextRank++;
			continue;
		}
		maxInteriorDepth = std::max(maxInteriorDepth, depthOf[i]);
		for (int c : {nd.left, nd.right}) {
			if (c > 0) {
				if ((uint32_t)c >= n) return fail(RTX_ERR_ARG, "child index out of range");
				depthOf[c] = depthOf[i] + 1;
				st.push_back(c);
			}
		}
	}
	if ((uint32_t)nextRank != ntri) return fail(RTX_ERR_ARG, "tree does not hold every triangle exactly once");
	if (maxDepth > 32) return fail(RTX_ERR_ARG, "tree deeper than 32 levels");

	// Pass 2: pair records.  Pair 0 holds the root alone; every interior node with at least one
	// child owns one pair, numbered breadth-first.
	std::vector<int> pairOf(n, 0);
	std::vector<int> order;  // interior nodes owning a pair, in pair order (pair index = position + 1)
	order.reserve(n / 2 + 1);
	{
		std::vector<int> q;
		q.push_back(0);
		for (size_t h = 0; h < q.size(); ++h) {
			int i = q[h];
			const HostNode& nd = N[i];
			if (nd.tri >= 0) continue;
			if (nd.left > 0 || nd.right > 0) {
				order.push_back(i);
				pairOf[i] = (int)order.size();
				if (nd.left > 0) q.push_back(nd.left);
				if (nd.right > 0) q.push_back(nd.right);
			}
		}
	}
	auto code_of = [&](int c) -> int {
		if (c <= 0) return 0;
		if (N[c].tri >= 0) return ~rankOf[N[c].tri];
		return pairOf[c];  // 0 when the interior node has no children: nothing below it
	};
	const size_t npairs = order.size() + 1;
	std::vector<float4> pairs(npairs * 4);
	auto put = [&](size_t p, int side, int node, int code) {
		const HostNode& nd = N[node];
		pairs[p * 4 + side * 2 + 0] = make_float4(nd.bmin[0], nd.bmin[1], nd.bmin[2], 0.0f);
		pairs[p * 4 + side * 2 + 1] = make_float4(nd.bmax[0], nd.bmax[1], nd.bmax[2], 0.0f);
		std::memcpy(&pairs[p * 4 + side * 2 + 0].w, &code, 4);
	};
	for (auto& q : pairs) q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	{
		int rootCode = N[0].tri >= 0 ? ~rankOf[N[0].tri] : pairOf[0];
		put(0, 0, 0, rootCode);
	}
	for (size_t k = 0; k < order.size(); ++k) {
		const HostNode& nd = N[order[k]];
		if (nd.left > 0) put(k + 1, 0, nd.left, code_of(nd.left));
		if (nd.right > 0) put(k + 1, 1, nd.right, code_of(nd.right));
	}

	// Pass 3: the 4-wide collapse walked by k_trace_wide.  The wide node of a binary interior node X holds
	// X's grandchildren, and X's children where those are leaves.  It may only be used when boxes nest
	// exactly and leaf boxes are the triangles' own bounds, which is what the reference's builder
	// produces (src/BVH.cpp:60-160); a tree that breaks either rule is walked by the binary kernel alone.
	bool wideOk = true;
	for (uint32_t i = 0; i < n && wideOk; ++i) {
		const HostNode& nd = N[i];
		if (!seen[i]) continue;
		if (nd.tri >= 0) {
			const HostTri& t = s->tris[nd.tri];
			for (int c = 0; c < 3; ++c) {
				float lo = std::min(std::min(t.v[0][c], t.v[1][c]), t.v[2][c]);
				float hi = std::max(std::max(t.v[0][c], t.v[1][c]), t.v[2][c]);
				if (!(nd.bmin[c] == lo) || !(nd.bmax[c] == hi)) wideOk = false;
			}
			continue;
		}
		for (int ch : {nd.left, nd.right}) {
			if (ch <= 0) continue;
			for (int c = 0; c < 3; ++c)
				if (!(N[ch].bmin[c] >= nd.bmin[c]) || !(N[ch].bmax[c] <= nd.bmax[c])) wideOk = false;
		}
	}
	std::vector<float> wide;
	int wideStackNeed = 1;
	const char* treeKind = "none";
	if (wideOk) {
		// own surface-area tree over the triangles' boxes; the reference's tree if that one cannot be used
		bool built = false;
		const char* want = std::getenv("RTX_WIDE_TREE");
		if (!(want && std::strcmp(want, "reference") == 0)) {
			std::vector<float> box((size_t)ntri * 6);
			for (uint32_t i = 0; i < ntri; ++i) {
				const HostTri& t = s->tris[i];
				for (int c = 0; c < 3; ++c) {
					box[(size_t)i * 6 + c] = std::min(std::min(t.v[0][c], t.v[1][c]), t.v[2][c]);
					box[(size_t)i * 6 + 3 + c] = std::max(std::max(t.v[0][c], t.v[1][c]), t.v[2][c]);
				}
			}
			std::vector<rtxwide::BinNode> own;
			const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
			rtxwide::build_sah(box, (int)ntri, own, threads);
			built = rtxwide::collapse_wide(own, rankOf, wide, wideStackNeed);
			treeKind = "surface-area";
		}
		if (!built) {
			std::vector<rtxwide::BinNode> ref(n);
			for (uint32_t i = 0; i < n; ++i) {
				std::memcpy(ref[i].bmin, N[i].bmin, 12);
				std::memcpy(ref[i].bmax, N[i].bmax, 12);
				ref[i].left = N[i].left > 0 ? N[i].left : -1;
				ref[i].right = N[i].right > 0 ? N[i].right : -1;
				ref[i].tri = N[i].tri >= 0 ? N[i].tri : -1;
			}
			built = rtxwide::collapse_wide(ref, rankOf, wide, wideStackNeed);
			treeKind = "reference";
		}
		wideOk = built;
	}

	if (std::getenv("RTX_TRACE_DEBUG"))
		std::fprintf(stderr, "[rtx] wide tree (%s): ok %d, %zu nodes, stack need %d (binary: %zu pairs, stack need %d)\n", treeKind,
					 (int)wideOk, wide.size() / 16, wideStackNeed, npairs, maxInteriorDepth + 1);

	// Triangle records in rank order.
	std::vector<float4> th((size_t)std::max<uint32_t>(ntri, 1) * 4), ts((size_t)std::max<uint32_t>(ntri, 1) * 4),
		tu((size_t)std::max<uint32_t>(ntri, 1) * 2);
	for (uint32_t i = 0; i < ntri; ++i) {
		const HostTri& t = s->tris[i];
		const size_t r = (size_t)rankOf[i];
		const float* v0 = t.v[0];
		const float* v1 = t.v[1];
		const float* v2 = t.v[2];
		float a1[3], a2[3];
		for (int c = 0; c < 3; ++c) {
			a1[c] = v0[c] - v2[c];  // src/Triangle.cpp:31
			a2[c] = v1[c] - v2[c];  // src/Triangle.cpp:32
		}
		float idf;
		int32_t id = (int32_t)i;
		std::memcpy(&idf, &id, 4);
		th[r * 4 + 0] = make_float4(t.pn[0], t.pn[1], t.pn[2], v0[0]);
		th[r * 4 + 1] = make_float4(v0[1], v0[2], v1[0], v1[1]);
		th[r * 4 + 2] = make_float4(v1[2], v2[0], v2[1], v2[2]);
		th[r * 4 + 3] = make_float4(idf, 0.0f, 0.0f, 0.0f);
		int32_t flags = (t.emit ? TRI_EMIT : 0) | (t.metal ? TRI_METAL : 0) | (t.transp ? TRI_TRANSP : 0);
		float texf, flagf;
		std::memcpy(&texf, &t.tex, 4);
		std::memcpy(&flagf, &flags, 4);
		ts[r * 4 + 0] = make_float4(t.n[0][0], t.n[0][1], t.n[0][2], t.color[0]);
		ts[r * 4 + 1] = make_float4(t.n[1][0], t.n[1][1], t.n[1][2], t.color[1]);
		ts[r * 4 + 2] = make_float4(t.n[2][0], t.n[2][1], t.n[2][2], t.color[2]);
		ts[r * 4 + 3] = make_float4(t.rough, t.ior, texf, flagf);
		tu[r * 2 + 0] = make_float4(t.uv[0][0], t.uv[0][1], t.uv[1][0], t.uv[1][1]);
		tu[r * 2 + 1] = make_float4(t.uv[2][0], t.uv[2][1], 0.0f, 0.0f);
	}

	HIPCHK(hipSetDevice(s->device));
	HIPCHK(s->dPairs.alloc(pairs.size()));
	HIPCHK(hipMemcpy(s->dPairs.p, pairs.data(), pairs.size() * sizeof(float4), hipMemcpyHostToDevice));
	if (wideOk) {
		HIPCHK(s->dWide.alloc(wide.size() / 4));
		HIPCHK(hipMemcpy(s->dWide.p, wide.data(), wide.size() * sizeof(float), hipMemcpyHostToDevice));
	}
	s->wideOk = wideOk;
	s->wideStackNeed = wideStackNeed;
	s->numWide = (uint32_t)(wide.size() / 16);
	HIPCHK(s->dTriHit.alloc(th.size()));
	HIPCHK(hipMemcpy(s->dTriHit.p, th.data(), th.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(s->dTriShade.alloc(ts.size()));
	HIPCHK(hipMemcpy(s->dTriShade.p, ts.data(), ts.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(s->dTriUv.alloc(tu.size()));
	HIPCHK(hipMemcpy(s->dTriUv.p, tu.data(), tu.size() * sizeof(float4), hipMemcpyHostToDevice));
	s->rankOf.swap(rankOf);
	s->numPairs = (uint32_t)npairs;
	s->treeDepth = maxDepth + 1;  // + the root-only pair
	s->stackNeed = maxInteriorDepth + 1;
	s->haveBvh = true;
	return RTX_OK;
}

RTX_API int rtx_scene_add_texture(rtx_scene* s, const uint8_t* rgb, int width, int height) {
	if (!s || !rgb) return fail(RTX_ERR_ARG, "null argument");
	if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28)) return fail(RTX_ERR_ARG, "texture size out of range");
	TexInfo ti;
	ti.offset = (s->texels.size() + 15) & ~(size_t)15;
	ti.w = width;
	ti.h = height;
	s->texels.resize(ti.offset + (size_t)width * height * 3);
	std::memcpy(s->texels.data() + ti.offset, rgb, (size_t)width * height * 3);
	s->texInfo.push_back(ti);
	s->texDirty = true;
	return (int)s->texInfo.size() - 1;
}

RTX_API int rtx_scene_set_skybox(rtx_scene* s, float brightness, const int texture_ids[6]) {
	if (!s || !texture_ids) return fail(RTX_ERR_ARG, "null argument");
	s->hasSky = true;
	s->skyBrightness = brightness;
	for (int i = 0; i < 6; ++i) s->skyTex[i] = texture_ids[i];
	return RTX_OK;
}

RTX_API int rtx_scene_set_tables(rtx_scene* s, const float cos512[512], const float sin512[512], const float gamma256[256]) {
	if (!s || !cos512 || !sin512 || !gamma256) return fail(RTX_ERR_ARG, "null argument");
	HIPCHK(hipSetDevice(s->device));
	std::vector<float> t(1280);
	std::memcpy(t.data(), cos512, 2048);
	std::memcpy(t.data() + 512, sin512, 2048);
	std::memcpy(t.data() + 1024, gamma256, 1024);
	HIPCHK(s->dTables.alloc(1280));
	HIPCHK(hipMemcpy(s->dTables.p, t.data(), 1280 * sizeof(float), hipMemcpyHostToDevice));
	s->haveTables = true;
	return RTX_OK;
}

RTX_API int rtx_scene_set_tiles(rtx_scene* s, int first, int stride, int count) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	if (first < 0 || stride < 1) return fail(RTX_ERR_ARG, "first >= 0 and stride >= 1 required");
	s->tileFirst = first;
	s->tileStride = stride;
	s->tileCount = count;
	return RTX_OK;
}

RTX_API int rtx_scene_set_arena_bytes(rtx_scene* s, uint64_t bytes) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	s->arenaLimit = bytes;
	return RTX_OK;
}

RTX_API int rtx_scene_set_accum_device(rtx_scene* s, void* device_rgba) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	s->dAccum = device_rgba ? (float4*)device_rgba : s->dAccumOwn.p;
	return RTX_OK;
}

RTX_API int rtx_accum_reset(rtx_scene* s) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	int rc = ensure_ready(s, false);
	if (rc) return rc;
	HIPCHK(hipMemsetAsync(s->dAccum, 0, (size_t)s->width * s->height * sizeof(float4), s->stream));
	HIPCHK(hipStreamSynchronize(s->stream));
	return RTX_OK;
}

RTX_API int rtx_render_frames(rtx_scene* s, uint32_t first_frame, uint32_t n_frames) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	size_t evTotal;
	rc = begin_event(s, 2, evTotal);
	if (rc) return rc;
	// the total-time event pair must survive the per-batch harvests: park it
	EventPair total = s->events.back();
	s->events.pop_back();
	for (uint32_t f = 0; f < n_frames; ++f) {
		rc = render_batches(s, first_frame + f, 0, 0, nullptr, nullptr, nullptr);
		if (rc) {
			(void)hipEventDestroy(total.a);
			(void)hipEventDestroy(total.b);
			return rc;
		}
	}
	s->events.push_back(total);
	rc = end_event(s, s->events.size() - 1);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(s->stream));
	rc = harvest_events(s);
	if (rc) return rc;
	return finish_stats(s);
}

RTX_API int rtx_accum_read(rtx_scene* s, float* host_rgba) {
	if (!s || !host_rgba) return fail(RTX_ERR_ARG, "null argument");
	int rc = ensure_ready(s, false);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(host_rgba, s->dAccum, (size_t)s->width * s->height * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
	HIPCHK(hipStreamSynchronize(s->stream));
	return RTX_OK;
}

RTX_API int rtx_accum_write(rtx_scene* s, const float* host_rgba) {
	if (!s || !host_rgba) return fail(RTX_ERR_ARG, "null argument");
	int rc = ensure_ready(s, false);
	if (rc) return rc;
	HIPCHK(hipMemcpyAsync(s->dAccum, host_rgba, (size_t)s->width * s->height * sizeof(float4), hipMemcpyHostToDevice, s->stream));
	HIPCHK(hipStreamSynchronize(s->stream));
	return RTX_OK;
}

RTX_API int rtx_scene_set_tonemap_table(rtx_scene* s, const uint32_t* bits, const uint8_t* values, int n, uint8_t negative, uint8_t nan) {
	if (!s || !bits || !values) return fail(RTX_ERR_ARG, "null argument");
	if (n < 1 || n > TONE_MAX) return fail(RTX_ERR_ARG, "tone map table size out of range");
	if (bits[0] != 0) return fail(RTX_ERR_ARG, "tone map table must start at +0");
	for (int i = 1; i < n; ++i)
		if (bits[i] <= bits[i - 1] || bits[i] > 0x7F800000u) return fail(RTX_ERR_ARG, "tone map breakpoints must ascend over the non-negative floats");
	HIPCHK(hipSetDevice(s->device));
	HIPCHK(s->dToneBits.alloc((size_t)n));
	HIPCHK(s->dToneVal.alloc((size_t)n));
	HIPCHK(hipMemcpy(s->dToneBits.p, bits, (size_t)n * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(s->dToneVal.p, values, (size_t)n, hipMemcpyHostToDevice));
	s->toneN = n;
	s->toneNeg = negative;
	s->toneNan = nan;
	return RTX_OK;
}

RTX_API int rtx_tonemap_read(rtx_scene* s, int frame, uint8_t* host_rgb8) {
	if (!s || !host_rgb8) return fail(RTX_ERR_ARG, "null argument");
	if (frame < 1) return fail(RTX_ERR_ARG, "frame must be >= 1");
	if (!s->toneN) return fail(RTX_ERR_STATE, "rtx_scene_set_tonemap_table has not been called");
	int rc = ensure_ready(s, false);
	if (rc) return rc;
	const size_t npix = (size_t)s->width * s->height;
	if (!s->dRgb8.p || s->dRgb8.n < npix * 3 + 16) HIPCHK(s->dRgb8.alloc(npix * 3 + 16));
	hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((npix + 1023) / 1024)), dim3(256), 0, s->stream, s->dAccum, (int)npix, (float)frame,
					   s->dToneBits.p, s->dToneVal.p, s->toneN, s->toneNeg, s->toneNan, s->dRgb8.p);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(host_rgb8, s->dRgb8.p, npix * 3, hipMemcpyDeviceToHost, s->stream));
	HIPCHK(hipStreamSynchronize(s->stream));
	return RTX_OK;
}

RTX_API int rtx_primary_hits(rtx_scene* s, int mode, int32_t* tri_ids, float* t_alpha_beta) {
	if (!s || !tri_ids) return fail(RTX_ERR_ARG, "null argument");
	if (mode != 0 && mode != 1) return fail(RTX_ERR_ARG, "mode must be 0 or 1");
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	const size_t ns = (size_t)s->width * s->height * 4;
	DevBuf<int> dIds;
	DevBuf<float> dTab;
	HIPCHK(dIds.alloc(ns));
	hipError_t e = hipMemsetAsync(dIds.p, 0xFF, ns * sizeof(int), s->stream);
	if (e == hipSuccess && t_alpha_beta) {
		e = dTab.alloc(ns * 3);
		if (e == hipSuccess) e = hipMemsetAsync(dTab.p, 0, ns * 3 * sizeof(float), s->stream);
	}
	if (e != hipSuccess) {
		dIds.release();
		dTab.release();
		return fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	}
	rc = render_batches(s, 0, 2, mode, nullptr, dIds.p, t_alpha_beta ? dTab.p : nullptr);
	if (rc == RTX_OK) {
		e = hipMemcpy(tri_ids, dIds.p, ns * sizeof(int), hipMemcpyDeviceToHost);
		if (e == hipSuccess && t_alpha_beta) e = hipMemcpy(t_alpha_beta, dTab.p, ns * 3 * sizeof(float), hipMemcpyDeviceToHost);
		if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	}
	dIds.release();
	dTab.release();
	if (rc) return rc;
	// primary rays actually traced: pixels of this scene's tiles that lie inside the image
	uint64_t px = 0;
	const int tilesX = (s->width + TILE - 1) / TILE;
	for (int k = 0, n = tile_slots(s); k < n; ++k) {
		const int tile = s->tileFirst + k * s->tileStride;
		const int tx = tile % tilesX, ty = tile / tilesX;
		px += (uint64_t)std::min(TILE, s->width - tx * TILE) * (uint64_t)std::min(TILE, s->height - ty * TILE);
	}
	s->stats.rays = s->stats.rays_per_level[0] = px * 4;
	return finish_stats(s);
}

RTX_API int rtx_trace_rays(rtx_scene* s, int mode, uint32_t n, const float* origins, const float* directions,
						   int32_t* tri_ids, float* t_alpha_beta) {
	if (!s || !origins || !directions || !tri_ids) return fail(RTX_ERR_ARG, "null argument");
	if (mode != 0 && mode != 1) return fail(RTX_ERR_ARG, "mode must be 0 or 1");
	if (n > (1u << 28)) return fail(RTX_ERR_ARG, "too many rays in one call");
	int rc = ensure_ready(s, false);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	if (n == 0) return RTX_OK;
	std::vector<float4> ho(n), hd(n);
	for (uint32_t i = 0; i < n; ++i) {
		ho[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.0f);
		hd[i] = make_float4(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2], 0.0f);
	}
	DevBuf<float4> dO, dD, dH;
	DevBuf<int> dIds, dList;
	DevBuf<float> dTab;
	hipError_t e = dO.alloc(n);
	if (e == hipSuccess) e = dList.alloc(n);
	if (e == hipSuccess) e = dD.alloc(n);
	if (e == hipSuccess) e = dH.alloc(n);
	if (e == hipSuccess) e = dIds.alloc(n);
	if (e == hipSuccess) e = dTab.alloc((size_t)n * 3);
	if (e == hipSuccess) e = hipMemcpy(dO.p, ho.data(), (size_t)n * 16, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(dD.p, hd.data(), (size_t)n * 16, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		const DevScene d = make_dev_scene(s);
		e = hipMemsetAsync(s->dCounts.p, 0, CNT_STRIDE * sizeof(unsigned), s->stream);
		if (e == hipSuccess) rc = launch_trace(s, d, false, mode == 1, dO.p, dD.p, dH.p, (int)n, s->dCounts.p, dList.p);
		if (rc == RTX_OK) {
			hipLaunchKernelGGL(k_export_hits_linear, dim3((n + 255) / 256), dim3(256), 0, s->stream, d, (int)n, dH.p, dIds.p, dTab.p);
			e = hipGetLastError();
			if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
			if (e == hipSuccess) e = hipMemcpy(tri_ids, dIds.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
			if (e == hipSuccess && t_alpha_beta) e = hipMemcpy(t_alpha_beta, dTab.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost);
		}
	}
	dO.release(); dD.release(); dH.release(); dIds.release(); dTab.release(); dList.release();
	if (e != hipSuccess) return fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	if (rc) return rc;
	rc = harvest_events(s);
	if (rc) return rc;
	s->stats.rays = n;
	s->stats.rays_per_level[0] = n;
	return finish_stats(s);
}

RTX_API int rtx_sample_radiance(rtx_scene* s, uint32_t frame, float* host_out) {
	if (!s || !host_out) return fail(RTX_ERR_ARG, "null argument");
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	const size_t ns = (size_t)s->width * s->height * 4;
	DevBuf<float4> dOut;
	HIPCHK(dOut.alloc(ns));
	hipError_t e = hipMemcpy(dOut.p, host_out, ns * sizeof(float4), hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		dOut.release();
		return fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	}
	rc = render_batches(s, frame, 1, 0, dOut.p, nullptr, nullptr);
	if (rc == RTX_OK) {
		e = hipMemcpy(host_out, dOut.p, ns * sizeof(float4), hipMemcpyDeviceToHost);
		if (e != hipSuccess) rc = fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	}
	dOut.release();
	if (rc) return rc;
	return finish_stats(s);
}

RTX_API int rtx_scene_set_instrumented(rtx_scene* s, int enabled) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	s->instrumented = enabled != 0;
	return RTX_OK;
}

RTX_API int rtx_get_stats(rtx_scene* s, rtx_stats* out) {
	if (!s || !out) return fail(RTX_ERR_ARG, "null argument");
	*out = s->stats;
	return RTX_OK;
}
