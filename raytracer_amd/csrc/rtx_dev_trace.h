// Traversal kernels of rtx_hip.hip: Triangle::hit and AABB::hit restated, the binary-tree kernel over the
// reference's own tree and the 4-wide kernel over the tree of rtx_wide.h (included by rtx_hip.hip).
#pragma once

#include "rtx_dev_common.h"

namespace {

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

// The same test for a record of the wide tree's leaves, which carries the part of the elimination that does not depend
// on the ray.  Columns 0 and 1 of the rows (a1_i, a2_i, b_i) are the triangle's own, so both pivot choices, the
// quotient p1 / p0 of the first pivot row and the second pivot e1 are known when the record is written (host side of
// rtx_scene_set_bvh, same operations in the same order: IEEE division, one fused multiply-add); what remains per ray
// is column 2: b of the first pivot's row over p0, one fused multiply-add into the second pivot's row, its division
// by e1, and the back-substitution -- the values src/Triangle.cpp:61-82 computes, bit for bit.
// bits: axis (0 x, 1 y, 2 z) of the first pivot's row in bits 0-1, of the second pivot's row in bits 2-3.
__device__ __forceinline__ bool tri_test_pre(float nx, float ny, float nz, float v0x, float v0y, float v0z, float v2x, float v2y,
											 float v2z, int rank, int bits, float p1q, float e1, float ox, float oy, float oz,
											 float dx, float dy, float dz, HitState& best) {
	float temp = dot3w(nx, ny, nz, v0x - ox, v0y - oy, v0z - oz);
	float t = temp / dot3w(nx, ny, nz, dx, dy, dz);
	if (!(t > 0.001f)) return false;
	if (t > best.t || (t == best.t && rank > best.rank)) return false;
	const float bx = (ox + t * dx) - v2x;
	const float by = (oy + t * dy) - v2y;
	const float bz = (oz + t * dz) - v2z;
	const float cx = v0x - v2x, cy = v0y - v2y, cz = v0z - v2z;  // column 0: a1 = v0 - v2 (src/Triangle.cpp:31)
	const int ax0 = bits & 3, ax1 = (bits >> 2) & 3;
	const float p0 = ax0 == 2 ? cz : (ax0 == 1 ? cy : cx);
	const float bp = ax0 == 2 ? bz : (ax0 == 1 ? by : bx);
	const float e0 = ax1 == 2 ? cz : (ax1 == 1 ? cy : cx);
	const float be = ax1 == 2 ? bz : (ax1 == 1 ? by : bx);
	const float p2 = bp / p0;
	float e2 = __fmaf_rn(-p2, e0, be);
	e2 = e2 / e1;
	const float alpha = __fmaf_rn(-e2, p1q, p2);
	const float beta = e2;
	const bool accept = (0.0f <= alpha) & (alpha <= 1.0f) & (0.0f <= beta) & (beta <= 1.0f) & (alpha + beta <= 1.0f);
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

// hits[j] = (rank, or -1 miss / -2 dead slot / -3 zero-valued ray that was not traced; t; alpha; beta), all as raw
// bits of a float4.
// ANYHIT: only "some candidate is accepted" is needed (rays at the last recursion depth,
// src/RayTracer.cpp:154-171); the answer is then one int per ray, the code alone with 0 for "hit", and the
// buffer is read as an int array.
// EXHAUSTIVE: no culling, i.e. the candidate set of BVH::hit itself.
//
// Persistent waves: rays of one queue differ in traversal length by orders of magnitude, so a wave
// that takes 64 rays and waits for the slowest runs mostly empty.  Here a wave keeps running and, once
// REFILL or more of its lanes are idle, hands them the next rays of its private chunk of the queue
// (CHUNK consecutive rays claimed with one atomic on *work, which the host zeroes before the launch).
// Every wave leaves the loop when the queue is exhausted and its lanes have drained.
constexpr int TRACE_CHUNK = 256;
constexpr int TRACE_REFILL = 16;

template <bool ANYHIT>
__device__ __forceinline__ void put_hit(float4* __restrict__ hits, int j, int code, float t, float alpha, float beta) {
	if (ANYHIT) reinterpret_cast<int*>(hits)[j] = code;
	else hits[j] = make_float4(__int_as_float(code), t, alpha, beta);
}

template <int STACK, bool ANYHIT, bool EXHAUSTIVE, bool STATS>
__global__ __launch_bounds__(TRACE_THREADS) void k_trace(DevScene sc, const float4* __restrict__ rayO,
														  const float4* __restrict__ rayD, float4* __restrict__ hits,
														  int n, unsigned long long* __restrict__ stats,
														  unsigned* __restrict__ work, int refill, int chunk, int triBatch,
														  const int* __restrict__ list, const unsigned* __restrict__ listCount,
														  const unsigned* __restrict__ nDev) {
	// the queue length may live on the device (written by the previous level's shade kernel); n is then the
	// capacity of the arrays
	const int capN = n;  // capacity of the ray arrays: what an entry of the list may index
	if (nDev) n = (int)min(*nDev, (unsigned)n);
	// with a list: the rays to trace are list[0 .. *listCount), the ones the wide traversal set aside
	if (list) n = (int)min(*listCount, (unsigned)capN);
	// A short queue is cut into smaller chunks, so that every wave of the grid has about three to take: with
	// chunks of the full size a queue of a million rays would occupy two waves in three and the launch would
	// last as long as one wave needs for its whole chunk.
	chunk = min(chunk, max(64, (int)((unsigned)n / (gridDim.x * (TRACE_THREADS / 64) * 3u) + 63u) & ~63));
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
					const bool inRange = (unsigned)j < (unsigned)capN;
					float4 O = make_float4(0.0f, 0.0f, 0.0f, 0.0f), D = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					if (inRange) {
						O = rayO[j];
						D = rayD[j];
					}
					if (!inRange) {
						atomicOr(sc.fault, FAULT_LIST_ENTRY);  // a list entry that is no ray of this queue: skipped, reported to the host
					}
					else if (D.w != 0.0f) {
						// not to be traced: 1 = no ray at all (refraction under total internal reflection),
						// 2 = a ray of the last level whose value is zero whether it hits or not
						put_hit<ANYHIT>(hits, j, D.w == 2.0f ? -3 : -2, FLT_MAX, 0.0f, 0.0f);
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
			put_hit<ANYHIT>(hits, j, id, best.t, best.alpha, best.beta);
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
// parent need not do: a ray that meets that case at a leaf whose y and z slabs it misses (see wide_leaf_test) is
// set aside in a list for the binary-tree kernel above, which evaluates every ancestor as the reference does; so
// are rays with a finite reciprocal beyond 2^100 (whose products overflow) and rays that need more stack than
// this kernel has.  The quantised
// tests are homogeneous in 1/d; they use +-2^100 in place of +-inf and so prune a parallel ray by
// "origin inside the slab, give or take the slack".
//
// Node, 4 x 16 bytes:  px py pz sx                          (origin; sx, sy, sz: quantisation steps, powers of two)
//                      lo.x[4] hi.x[4] lo.y[4] hi.y[4]      (one byte per child)
//                      lo.z[4] hi.z[4] code0 code1
//                      code2 code3 sy sz                       code > 0 node, < 0 ~rank, 0 absent
constexpr float WIDE_MAX_INV = 1.2676506e30f;  // 2^100

__device__ __forceinline__ float ubyte_f(unsigned v, int k) { return (float)((v >> (8 * k)) & 0xFFu); }

// One triangle of a leaf of the wide tree: the reference's slab predicate on the triangle's own box (src/BVH.cpp:14-41;
// fmaxf/fminf skip a NaN in lanes 1 and 2 as the reference's folds do), then Triangle::hit.  T: the triangle's record
// in sc.triWide.  Returns 1 when the triangle was accepted (best updated), 0 when not, 2 when only the reference's
// own tree can tell.
// NaN in lane 0 (the origin lies on an x plane of the box and the ray runs along it): the reference accepts such a
// box whatever y and z say, provided it gets there.  The folds here skip that NaN too, which leaves x unconstrained;
// an ancestor's box contains this one, so x does not constrain it either (the origin's x lies within its range: its
// lane 0 is +-inf on both sides, or NaN and then it accepts outright), and its y and z slabs are wider.  If y and z
// pass here, every ancestor therefore passes, the reference gets to this box, and accepts it.  If they fail here the
// reference still accepts the box once it gets there, which the ancestors' y and z decide: 2.
template <bool STATS>
__device__ __forceinline__ int wide_leaf_test(const float4* __restrict__ T, const RaySlab& r, float dx, float dy, float dz, HitState& best,
											  unsigned& boxTests, unsigned& triTests) {
	const float4 q0 = T[0], q1 = T[1], q2 = T[2];
	// fourth quad: index | rank << 4 | pivot rows | p1 / p0 | second pivot (see tri_test_pre)
	const int rankBits = __float_as_int(T[3].y);
	const float p1q = T[3].z, e1 = T[3].w;
	const int rank = rankBits >> 4;
	// all of the record is asked for at once: left to itself the compiler moves the loads of the plane normal and of the
	// rank behind the box test, where they cost further memory round trips in every leaf phase (2 to 5 % of a launch)
	asm volatile("" : : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(rankBits), "v"(p1q), "v"(e1));
	const float v0x = q0.w, v0y = q1.x, v0z = q1.y, v1x = q1.z, v1y = q1.w, v1z = q2.x, v2x = q2.y, v2y = q2.z, v2z = q2.w;
	const float lx = fminf(fminf(v0x, v1x), v2x), hx = fmaxf(fmaxf(v0x, v1x), v2x);
	const float ly = fminf(fminf(v0y, v1y), v2y), hy = fmaxf(fmaxf(v0y, v1y), v2y);
	const float lz = fminf(fminf(v0z, v1z), v2z), hz = fmaxf(fmaxf(v0z, v1z), v2z);
	const float t0x = (lx - r.ox) * r.ix, t1x = (hx - r.ox) * r.ix;
	const float t0y = (ly - r.oy) * r.iy, t1y = (hy - r.oy) * r.iy;
	const float t0z = (lz - r.oz) * r.iz, t1z = (hz - r.oz) * r.iz;
	const float tmin = fmaxf(fmaxf(r.nx ? t1x : t0x, r.ny ? t1y : t0y), r.nz ? t1z : t0z);
	const float tmax = fminf(fminf(r.nx ? t0x : t1x, r.ny ? t0y : t1y), r.nz ? t0z : t1z);
	if (STATS) boxTests++;
	if ((t0x != t0x || t1x != t1x) && tmax < tmin) return 2;
	if (tmax < tmin) return 0;
	if (STATS) triTests++;
	return tri_test_pre(q0.x, q0.y, q0.z, v0x, v0y, v0z, v2x, v2y, v2z, rank, rankBits & 15, p1q, e1, r.ox, r.oy, r.oz, dx, dy, dz, best) ? 1 : 0;
}

template <int STACK, bool ANYHIT, bool STATS>
__global__ __launch_bounds__(TRACE_THREADS) __attribute__((amdgpu_waves_per_eu(STACK <= 16 ? 7 : 6, STACK <= 16 ? 7 : 6))) void k_trace_wide(DevScene sc, const float4* __restrict__ rayO,
															   const float4* __restrict__ rayD, float4* __restrict__ hits,
															   int n, unsigned long long* __restrict__ stats,
															   unsigned* __restrict__ work, int* __restrict__ unsafeList,
															   unsigned* __restrict__ unsafeCount, int refill, int chunk,
															   int triBatch, const unsigned* __restrict__ nDev,
															   int* __restrict__ longList, unsigned* __restrict__ longCount, int longVisits,
															   int smallQueue, int tailVisits, int tailLanes, int retire) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	// A queue of a few thousand rays cannot fill the chip with one ray per lane; its launch would last as long as
	// its longest ray walked by a single lane.  All its rays go to k_trace_long, a wave each, at once.
	if (n <= smallQueue) longVisits = 0;
	// A short queue is cut into smaller chunks, so that every wave of the grid has about three to take: with
	// chunks of the full size a queue of a million rays would occupy two waves in three and the launch would
	// last as long as one wave needs for its whole chunk.
	chunk = min(chunk, max(64, (int)((unsigned)n / (gridDim.x * (TRACE_THREADS / 64) * 3u) + 63u) & ~63));
	// The per-lane stack lives in LDS, one row of the array per entry, and grows downwards; a lane's stack pointer is
	// kept as the LDS address of its top entry (`top`), so that a push or a pop is one LDS instruction with a constant
	// offset and no address arithmetic of its own.  An empty stack points one row past the array.
	// A node's four children are all written, sorted, to the four rows below `top` whether they are entered or not --
	// the ones that are not sort to the far end, below what `top` then moves down to -- and the nearest is popped
	// straight back: four stores and a load that cost no vector-ALU instruction, where choosing what to store did.
	// The four rows of room that takes come out of the stack's depth: a ray whose stack reaches into the lowest four
	// rows when it comes to a node does not fit.
	__shared__ int stk[STACK + 1][TRACE_THREADS];
	__shared__ unsigned char sPool[TRACE_THREADS / 64][128];  // leaf phase: lanes that want help | lanes that help, by rank
	const int tid = threadIdx.x;
	typedef __attribute__((address_space(3))) int lds_int;
	constexpr int ROW = TRACE_THREADS;                          // ints from one stack entry of a lane to the next
	lds_int* const stkEmpty = (lds_int*)&stk[0][tid] + (STACK + 1) * ROW;  // `top` of an empty stack (never dereferenced)
	lds_int* const stkFloor = (lds_int*)&stk[4][tid];           // a node is visited only while `top` is at or above this row
	lds_int* top = stkEmpty;
	const unsigned lane = (unsigned)tid & 63u;
	const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
	unsigned boxTests = 0, triTests = 0, nodeVisits = 0, leafFetches = 0;
	unsigned walkIters = 0, leafIters = 0, walkLanes = 0, leafLanes = 0;  // wave-level, kept by lane 0
	unsigned walkIdle = 0, walkParked = 0, walkLeafWait = 0, leafIdle = 0, leafWalkers = 0, refills = 0, refillLanes = 0;  // likewise
	unsigned overflows = 0;
	unsigned alonePasses = 0, aloneLanes = 0, helpedLanes = 0;  // likewise
	unsigned myVisits = 0, maxVisits = 0, over100 = 0, over300 = 0;
	int cnext = 0, cend = 0;
	bool more = true, firstChunk = true, lastChunk = false;
	int j = -1, cur = -1;
	RaySlab r;
	r.ox = r.oy = r.oz = r.ix = r.iy = r.iz = r.mx = r.my = r.mz = 0.0f;
	r.nx = r.ny = r.nz = false;
	float dx = 0.0f, dy = 0.0f, dz = 0.0f;
	float wix = 0.0f, wiy = 0.0f, wiz = 0.0f;  // 1/d as the interior tests use it: +-inf replaced by +-2^100
	HitState best;
	best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
	bool found = false, busy = false;
	int pend0 = -1, pend1 = -1;
	int visitsLeft = 0;  // nodes this lane's ray may still visit here before it moves to k_trace_long

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
				else if (!lastChunk) {
					if (lane == 0) base = atomicAdd(work, (unsigned)chunk);
					base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + gridDim.x * (TRACE_THREADS / 64) * (unsigned)chunk;
				}
				if (lastChunk || base >= (unsigned)n) {
					more = false;
					cnext = cend = 0;
					// The queue is exhausted: from here on the launch lasts as long as this wave's slowest ray.  Rays
					// that still have a long way to go move to the wave-per-ray kernel sooner than they otherwise would.
					visitsLeft = min(visitsLeft, tailVisits);
				}
				else {
					cnext = (int)base;
					cend = min((int)base + chunk, n);
					// Left to themselves all waves of the grid find the queue exhausted at about the same time, and the launch
					// ends with every one of them running its last rays out on ever fewer lanes, side by side.  Instead the
					// waves leave one after the other as the queue runs low: this chunk is a wave's last one when what remains
					// behind it is less than `retire` rays for each wave that ranks below it (the rank is a hash of the wave's
					// number, which spreads the leavers over the SIMDs).  Those that stay find the vector ALUs less contended.
					if (retire > 0) {
						const unsigned g = blockIdx.x * (TRACE_THREADS / 64) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
						const unsigned long long rank = (g * 2654435761u) >> 16;  // 0 .. 65535
						const unsigned long long room = (rank * gridDim.x * (TRACE_THREADS / 64) * (unsigned long long)retire) >> 16;
						lastChunk = (unsigned long long)(n - cend) < room;
					}
				}
			}
			const int take = min(nidle, cend - cnext);
			if (take > 0) {
				if (STATS) {
					refills++;
					refillLanes += take;
				}
				const int slot = __popcll(idle & below);
				if (!busy && slot < take) {
					j = cnext + slot;
					float4 O = rayO[j];
					float4 D = rayD[j];
					// the origin is asked for together with the direction, not behind the test of the direction's marker
					asm volatile("" : : "v"(O.x), "v"(O.y), "v"(O.z));
					if (D.w != 0.0f) {
						// not to be traced: 1 = no ray at all (refraction under total internal reflection),
						// 2 = a ray of the last level whose value is zero whether it hits or not
						put_hit<ANYHIT>(hits, j, D.w == 2.0f ? -3 : -2, FLT_MAX, 0.0f, 0.0f);
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
							put_hit<ANYHIT>(hits, j, -1, FLT_MAX, 0.0f, 0.0f);
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
							top = stkEmpty;
							cur = 0;
							pend0 = pend1 = -1;
							busy = true;
							visitsLeft = longVisits;
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
		// A wave that has found the queue exhausted and is down to a few rays hands those to the wave-per-ray kernel: they
		// are its longest, and one lane each would walk them one node after the other while the launch waits.
		if (!more && 64 - nidle <= tailLanes) visitsLeft = 0;
		const bool canWalk = busy && pend1 < 0 && (cur >= 0 || top != stkEmpty);
		const unsigned long long mWalk = __builtin_amdgcn_ballot_w64(canWalk);
		const unsigned long long mLeaf = __builtin_amdgcn_ballot_w64(busy && pend0 >= 0);
		if (mWalk == 0ull && mLeaf == 0ull) {
			if (!more) break;
			continue;
		}
		// The leaves are tested once triBatch of 64 lanes hold one -- of the lanes that have a ray: a wave that is running
		// empty (the end of the queue, a short queue) would otherwise come to its leaves only when none of its lanes can
		// walk any more, with the lanes that hold two leaves waiting all that time.
		const bool leafPhase = __popcll(mLeaf) >= max(1, ((64 - nidle) * triBatch) >> 6) || mWalk == 0ull;
		if (STATS) {
			if (leafPhase) {
				leafIters++;
				leafLanes += __popcll(mLeaf);
				leafIdle += __popcll(__builtin_amdgcn_ballot_w64(!busy));
				leafWalkers += __popcll(mWalk & ~mLeaf);
			}
			else {
				walkIters++;
				walkLanes += __popcll(mWalk);
				walkIdle += __popcll(__builtin_amdgcn_ballot_w64(!busy));
				walkParked += __popcll(__builtin_amdgcn_ballot_w64(busy && pend1 >= 0));
				walkLeafWait += __popcll(__builtin_amdgcn_ballot_w64(busy && pend1 < 0 && cur < 0 && top == stkEmpty));
			}
		}
		if (leafPhase) {
			// A leaf is one triangle or two that share a box (value = first record * 2 + 1 if there is a second): each is
			// tested on its own, exact box, exact slab predicate, exact triangle test; one leaf per lane and pass.
#ifdef RTX_BOUNDS_CHECK
			if (busy && pend0 >= 0 && (unsigned)((pend0 >> 1) + (pend0 & 1)) >= (unsigned)sc.nTris) {
				printf("[rtx] k_trace_wide: leaf record %d (+%d) of %d (ray %d, sp %d, cur %d, pend1 %d)\n", pend0 >> 1, pend0 & 1, sc.nTris, j, (int)(stkEmpty - top) / ROW, cur, pend1);
				atomicOr(sc.fault, FAULT_LEAF);
				pend0 = pend1 = -1; cur = -1; top = stkEmpty;
			}
#endif
			// The second triangle of a leaf is tested at the same time as the first, by a lane that has no leaf of its own
			// (one without a ray, or a walker waiting for this phase): leaves that want help and lanes that can give it
			// are matched by their ranks in the two ballots, through a small table in LDS; the helper pulls the ray and
			// the best hit so far from its partner (ds_bpermute), tests record + 1 against them, and the partner pulls the
			// outcome back and keeps the better of the two results -- smallest t, lower rank on ties, as if it had tested
			// the triangles one after the other.  Seconds that found no helper are tested by their own lane afterwards.
			const bool todo = busy && pend0 >= 0;
			const bool second = todo && (pend0 & 1) != 0;
			const unsigned long long mTodo = __builtin_amdgcn_ballot_w64(todo), mSecond = __builtin_amdgcn_ballot_w64(second);
			const int rankSecond = __popcll(mSecond & below), rankFree = __popcll(~mTodo & below);
			const int matched = min(__popcll(mSecond), __popcll(~mTodo));
			const bool helped = second && rankSecond < matched;
			const bool helper = !todo && rankFree < matched;
			unsigned char* pool = sPool[tid >> 6];
			if (helped) pool[rankSecond] = (unsigned char)lane;
			if (helper) pool[64 + rankFree] = (unsigned char)lane;
			__builtin_amdgcn_wave_barrier();
			const int src = helper ? (int)pool[rankFree] : (int)lane;           // whose ray and leaf this lane works on
			const int back = helped ? (int)pool[64 + rankSecond] : (int)lane;   // who works on this lane's second triangle
			__builtin_amdgcn_wave_barrier();
			RaySlab q;
			q.ox = __shfl(r.ox, src, 64); q.oy = __shfl(r.oy, src, 64); q.oz = __shfl(r.oz, src, 64);
			q.ix = __shfl(r.ix, src, 64); q.iy = __shfl(r.iy, src, 64); q.iz = __shfl(r.iz, src, 64);
			q.mx = q.my = q.mz = 0.0f;
			q.nx = (__float_as_uint(q.ix) >> 31) != 0;
			q.ny = (__float_as_uint(q.iy) >> 31) != 0;
			q.nz = (__float_as_uint(q.iz) >> 31) != 0;
			const float qdx = __shfl(dx, src, 64), qdy = __shfl(dy, src, 64), qdz = __shfl(dz, src, 64);
			HitState hs = best;
			hs.t = __shfl(best.t, src, 64);
			hs.rank = __shfl(best.rank, src, 64);
			const int leaf = __shfl(pend0, src, 64);
			int res = 0;
			if (todo || helper) {
				if (STATS) leafFetches++;
				res = wide_leaf_test<STATS>(sc.triWide + (size_t)((leaf >> 1) + (helper ? 1 : 0)) * 4, q, qdx, qdy, qdz, hs, boxTests, triTests);
			}
			// what the helper found, seen from the lane it helped
			const int hres = __shfl(res, back, 64);
			const float ht = __shfl(hs.t, back, 64), ha = __shfl(hs.alpha, back, 64), hb = __shfl(hs.beta, back, 64);
			const int hrank = __shfl(hs.rank, back, 64);
			bool aside = false;
			if (todo) {
				if (res == 1) {
					best = hs;
					found = true;
				}
				aside = res == 2;
				if (helped) {
					aside = aside || hres == 2;
					if (hres == 1 && (ht < best.t || (ht == best.t && hrank < best.rank))) {
						best.t = ht; best.alpha = ha; best.beta = hb; best.rank = hrank;
						found = true;
					}
				}
			}
			// seconds without a helper
			const bool alone = todo && second && !helped && !aside && !(ANYHIT && found);
			if (__builtin_amdgcn_ballot_w64(alone) != 0ull) {
				if (STATS) {
					alonePasses++;
					aloneLanes += __popcll(__builtin_amdgcn_ballot_w64(alone));
					helpedLanes += __popcll(__builtin_amdgcn_ballot_w64(helped));
				}
				if (alone) {
					if (STATS) leafFetches++;
					const int res2 = wide_leaf_test<STATS>(sc.triWide + (size_t)((pend0 >> 1) + 1) * 4, r, dx, dy, dz, best, boxTests, triTests);
					if (res2 == 1) found = true;
					aside = res2 == 2;
				}
			}
			if (aside) {
				// the ray starts over in the binary-tree kernel
				unsafeList[atomicAdd(unsafeCount, 1u)] = j;
				busy = false;
				cur = -1;
				top = stkEmpty;
				pend0 = pend1 = -1;
			}
			if (busy && pend0 >= 0) {
				pend0 = pend1;
				pend1 = -1;
				if (ANYHIT && found) {
					pend0 = -1;
					cur = -1;
					top = stkEmpty;
				}
			}
		}
		else if (canWalk) {
			int item = 0;  // what this lane looks at next: > 0 node, < 0 ~rank of a leaf, 0 nothing yet
#ifdef RTX_BOUNDS_CHECK
			if (cur >= sc.nWide) {
				printf("[rtx] k_trace_wide: node %d of %d (ray %d, sp %d, pend %d %d)\n", cur, sc.nWide, j, (int)(stkEmpty - top) / ROW, pend0, pend1);
				atomicOr(sc.fault, FAULT_NODE);
				cur = -1; top = stkEmpty;
			}
			if (top > stkEmpty || top < (lds_int*)&stk[0][tid] || (int)(stkEmpty - top) % ROW != 0) {
				printf("[rtx] k_trace_wide: stack pointer %d (ray %d)\n", (int)(stkEmpty - top), j);
				atomicOr(sc.fault, FAULT_STACK);
				top = stkEmpty;
			}
#endif
			// A launch lasts as long as its longest ray, and one lane walks one node after the other.  A ray
			// that has used up its allowance starts over in k_trace_long, where a whole wave works on it.  (Rare: the
			// wave asks once whether any of its lanes is there, so that the step of the others does not run through the
			// hand-over's branches.)
			if (cur >= 0) --visitsLeft;
			if (__builtin_amdgcn_ballot_w64(visitsLeft < 0) != 0ull) {
				if (cur >= 0 && visitsLeft < 0) {
					longList[atomicAdd(longCount, 1u)] = j;
					busy = false;
					cur = -1;
					top = stkEmpty;
					pend0 = -1;
				}
			}
			if (cur >= 0) {
				const float4* W = sc.wide + (size_t)cur * 4;
				float4 w0 = W[0], w1 = W[1], w2 = W[2], w3 = W[3];
				if (STATS) { nodeVisits++; myVisits++; }
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
				// key: ordered bits of what the children are sorted by (never zero), zero for a child that is not entered
				unsigned key[4];
				int nh = 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					float e = fmaxf(fmaxf(__fmaf_rn(ubyte_f(nxd, c), Sx, Anx), __fmaf_rn(ubyte_f(nyd, c), Sy, Any)),
									__fmaf_rn(ubyte_f(nzd, c), Sz, Anz));
					float x = fminf(fminf(__fmaf_rn(ubyte_f(fxd, c), Sx, Afx), __fmaf_rn(ubyte_f(fyd, c), Sy, Afy)),
									__fmaf_rn(ubyte_f(fzd, c), Sz, Afz));
					e = fmaxf(e, 0.001f);
					// an absent child is an empty interval on every axis (lo 255, hi 0), which the slack can reopen only in a
					// node a few slacks wide; its code is 0, and an item 0 is skipped wherever items are taken up
					const bool h = !(x < e) & !(e > best.t);
					// closest hit: nearest child first.  Any hit: the child the ray stays in longest first, where an
					// occluder is likeliest (any order gives the same answer; the key stays below the "not
					// entered" marker because the length is clamped to >= +0).
					key[c] = !h ? 0u : ANYHIT ? 0x7FFFFFFFu - __float_as_uint(fmaxf(x - e, 0.0f)) : __float_as_uint(e);
					nh += h ? 1 : 0;
					if (STATS) boxTests += cd[c] != 0;
				}
				// sorting network on (key, code): the children that are not entered first, then the entered ones, nearest first
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
				if (__builtin_amdgcn_ballot_w64(top < stkFloor) != 0ull) {
					if (top < stkFloor) {
						// no room: the ray starts over in the wave-per-ray kernel, whose stack is a thousand entries deep (in
						// the binary-tree kernel, whose stack covers its whole tree, where that kernel is not run)
						if (smallQueue >= 0) longList[atomicAdd(longCount, 1u)] = j;
						else unsafeList[atomicAdd(unsafeCount, 1u)] = j;
						if (STATS) overflows++;
						busy = false;
						top = stkEmpty;
						nh = 0;
						pend0 = -1;
					}
				}
				// all four go below the top, the farthest next to it; the top then moves down over the nh entered ones
				lds_int* low = top - 4 * ROW;
				asm("" : "+v"(low));  // (one subtraction and four constant offsets; seen through, it becomes four additions)
				low[0] = cd[0];
				low[ROW] = cd[1];
				low[2 * ROW] = cd[2];
				low[3 * ROW] = cd[3];
				top -= nh * ROW;
				cur = -1;
			}
			if (top != stkEmpty) {
				item = *top;
				top += ROW;
			}
			if (item < 0) {
				// a leaf: keep it and look at what comes after it
				if (pend0 < 0) pend0 = ~item;
				else pend1 = ~item;
				item = 0;
				if (pend1 < 0 && top != stkEmpty) {
					item = *top;
					top += ROW;
					if (item < 0) {
						pend1 = ~item;
						item = 0;
					}
				}
			}
			if (item > 0) cur = item;
		}
		if (busy && cur < 0 && pend0 < 0 && top == stkEmpty) {
			int id = found ? (ANYHIT ? 0 : best.rank) : -1;
			put_hit<ANYHIT>(hits, j, id, best.t, best.alpha, best.beta);
			busy = false;
			if (STATS) { maxVisits = max(maxVisits, myVisits); over100 += myVisits > 100; over300 += myVisits > 300; myVisits = 0; }
		}
	}
	if (STATS) {
		atomicMax(&stats[9], (unsigned long long)maxVisits);
		atomicAdd(&stats[10], (unsigned long long)over100);
		atomicAdd(&stats[11], (unsigned long long)over300);
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
			atomicAdd(&stats[12], (unsigned long long)walkIdle);
			atomicAdd(&stats[13], (unsigned long long)walkParked);
			atomicAdd(&stats[14], (unsigned long long)walkLeafWait);
			atomicAdd(&stats[15], (unsigned long long)leafIdle);
			atomicAdd(&stats[16], (unsigned long long)leafWalkers);
			atomicAdd(&stats[17], (unsigned long long)refills);
			atomicAdd(&stats[18], (unsigned long long)refillLanes);
			atomicAdd(&stats[19], (unsigned long long)alonePasses);
			atomicAdd(&stats[20], (unsigned long long)aloneLanes);
			atomicAdd(&stats[21], (unsigned long long)helpedLanes);
		}
		for (int off = 32; off > 0; off >>= 1) overflows += __shfl_down(overflows, off, 64);
		if (lane == 0) {
			atomicAdd(&stats[8], (unsigned long long)overflows);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// One wave per ray, for the few rays whose traversal is long (a ray skimming a finely tessellated surface
// visits hundreds of nodes; walked by one lane, it alone decides how long its level's launch lasts).
//
// Same tree, same tests, same answer as k_trace_wide; only the schedule differs.  The wave keeps the ray's open
// nodes in a stack in LDS and takes the top 16 at a time: lane 4s+c tests child c of node s.  Children that are
// entered go back on the stack, leaves into a list; once the list holds 64 leaves (or the stack is empty) every
// lane tests one leaf with the exact box predicate and the exact Triangle::hit, and the wave keeps the minimum
// of (t, rank) -- the reference's "smallest t, first candidate on ties" -- so the order of the visits cannot
// change the result.  The best t so far is uniform over the wave and culls as in k_trace_wide.
// A ray whose stack would overflow, or that meets the NaN-in-lane-0 case at a leaf, goes on to the set-aside
// list of the binary-tree kernel, which runs after this one.
constexpr int LONG_STACK = 1024;     // open nodes per ray
constexpr int LONG_LEAVES = 192;     // >= 63 (below the trigger) + 128 (one step's yield: 64 leaves of two triangles)

template <bool ANYHIT, bool STATS>
__global__ __launch_bounds__(TRACE_THREADS) void k_trace_long(DevScene sc, const float4* __restrict__ rayO,
															  const float4* __restrict__ rayD, float4* __restrict__ hits,
															  const int* __restrict__ list, const unsigned* __restrict__ listCount,
															  int* __restrict__ unsafeList,
															  unsigned* __restrict__ unsafeCount, unsigned long long* __restrict__ stats, int capN) {
	__shared__ int sStack[TRACE_THREADS / 64][LONG_STACK];
	__shared__ int sLeaf[TRACE_THREADS / 64][LONG_LEAVES];
	const unsigned lane = threadIdx.x & 63u;
	const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
	int* stk = sStack[w];
	int* lf = sLeaf[w];
	const unsigned count = min(*listCount, (unsigned)capN);  // capN: capacity of the ray arrays and of the list
	unsigned nodeVisits = 0, leafFetches = 0, boxTests = 0, triTests = 0;  // wave-level, kept uniform
	// Everything a wave does here is uniform over its lanes, and the code keeps it so by construction: the counts of
	// the two lists are computed by every lane from ballots (never updated by one lane on behalf of the others), no
	// statement is guarded by "lane == 0" (all lanes store the same value to the same place instead), and every ballot
	// and shuffle runs under the full wave or under a condition that is the same in all lanes.  The lanes of a wave
	// share one program counter; what a guarded update of a loop-carried count can do is leave the lanes with different
	// copies of it, so that they disagree on the loop condition and the next ballot sees part of the wave.  Whatever
	// the state, the loop below ends: it is bounded by the number of nodes and triangles (maxRounds).
	const unsigned nWaves = gridDim.x * (TRACE_THREADS / 64);
	for (unsigned idx = blockIdx.x * (TRACE_THREADS / 64) + (unsigned)w; idx < count; idx += nWaves) {
		const int j = list[idx];
		if ((unsigned)j >= (unsigned)capN) {
			atomicOr(sc.fault, FAULT_LIST_ENTRY);  // not a ray of this queue (uniform over the wave): skipped, reported to the host
			continue;
		}
		const float4 O = rayO[j], D = rayD[j];
		// the ray was being walked by k_trace_wide, so its reciprocals are finite or infinite, never huge
		RaySlab r;
		r.ox = O.x; r.oy = O.y; r.oz = O.z;
		const float dx = D.x, dy = D.y, dz = D.z;
		r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
		r.nx = (__float_as_uint(r.ix) >> 31) != 0;
		r.ny = (__float_as_uint(r.iy) >> 31) != 0;
		r.nz = (__float_as_uint(r.iz) >> 31) != 0;
		const float aix = fabsf(r.ix), aiy = fabsf(r.iy), aiz = fabsf(r.iz);
		const float wix = aix > FLT_MAX ? copysignf(WIDE_MAX_INV, r.ix) : r.ix;
		const float wiy = aiy > FLT_MAX ? copysignf(WIDE_MAX_INV, r.iy) : r.iy;
		const float wiz = aiz > FLT_MAX ? copysignf(WIDE_MAX_INV, r.iz) : r.iz;
		const float scale = fmaxf(fmaxf(sc.sceneAbsMax, fabsf(r.ox)), fmaxf(fabsf(r.oy), fabsf(r.oz)));
		const float eps = scale * 6.103515625e-05f;
		r.mx = eps * fabsf(wix); r.my = eps * fabsf(wiy); r.mz = eps * fabsf(wiz);

		HitState best;  // uniform over the wave
		best.t = FLT_MAX; best.alpha = 0.0f; best.beta = 0.0f; best.rank = 0x7FFFFFFF;
		bool found = false, aside = false;
		int nStk = 1, nLeaf = 0;  // uniform
		stk[0] = 0;
		__builtin_amdgcn_wave_barrier();
		const int slot = (int)(lane >> 2), child = (int)(lane & 3u);
		// every round takes at least one node or one leaf off the lists and nothing is visited twice
		const unsigned maxRounds = (unsigned)sc.nWide + (unsigned)sc.nTris + 2u;
		unsigned rounds = 0;
		while (nStk > 0 || nLeaf > 0) {
			if (++rounds > maxRounds) {
				atomicOr(sc.fault, FAULT_LONG_ROUNDS);  // the tree is not a tree
				aside = true;
				break;
			}
			if (nLeaf >= 64 || nStk == 0) {
				// leaf phase: the newest (at most 64) leaves, one per lane
				const int take = min(nLeaf, 64);
				nLeaf -= take;
				unsigned long long key = ~0ull;
				HitState cand = best;
				bool nanX = false;
				if ((int)lane < take) {
					const int rec = lf[nLeaf + (int)lane];  // a record of sc.triWide: one triangle
					unsigned bt = 0, tt = 0;
					const int res = wide_leaf_test<false>(sc.triWide + (size_t)rec * 4, r, dx, dy, dz, cand, bt, tt);
					nanX = res == 2;
					if (res == 1) key = ((unsigned long long)__float_as_uint(cand.t) << 32) | (unsigned)cand.rank;
				}
				if (STATS) { leafFetches += take; boxTests += take; triTests += take; }
				if (__builtin_amdgcn_ballot_w64(nanX) != 0ull) {
					// only the reference's own tree can tell whether that leaf is reached (see k_trace_wide)
					aside = true;
					break;
				}
				// minimum of (t, rank) over the wave; a triangle is in the list at most once, so the winner is unique
				unsigned long long m = key;
				for (int off = 32; off > 0; off >>= 1) {
					const unsigned long long o = __shfl_xor(m, off, 64);
					m = o < m ? o : m;
				}
				if (m != ~0ull) {
					const int win = __ffsll((long long)__builtin_amdgcn_ballot_w64(key == m)) - 1;
					best.t = __shfl(cand.t, win, 64);
					best.alpha = __shfl(cand.alpha, win, 64);
					best.beta = __shfl(cand.beta, win, 64);
					best.rank = __shfl(cand.rank, win, 64);
					found = true;
					if (ANYHIT) break;
				}
				continue;
			}
			// node phase: the newest (at most 16) open nodes, four lanes each
			const int take = min(nStk, 16);
			nStk -= take;
			int code = 0;
			bool h = false;
			if (slot < take) {
				const int node = stk[nStk + slot];
				const float4* W = sc.wide + (size_t)node * 4;
				const float4 w0 = W[0], w1 = W[1], w2 = W[2], w3 = W[3];
				const float Sx = w0.w * wix, Sy = w3.z * wiy, Sz = w3.w * wiz;
				const float Ax = (w0.x - r.ox) * wix, Ay = (w0.y - r.oy) * wiy, Az = (w0.z - r.oz) * wiz;
				const float Anx = Ax - r.mx, Any = Ay - r.my, Anz = Az - r.mz;
				const float Afx = Ax + r.mx, Afy = Ay + r.my, Afz = Az + r.mz;
				const unsigned lox = __float_as_uint(w1.x), hix = __float_as_uint(w1.y), loy = __float_as_uint(w1.z),
							   hiy = __float_as_uint(w1.w), loz = __float_as_uint(w2.x), hiz = __float_as_uint(w2.y);
				const unsigned nxd = r.nx ? hix : lox, fxd = r.nx ? lox : hix;
				const unsigned nyd = r.ny ? hiy : loy, fyd = r.ny ? loy : hiy;
				const unsigned nzd = r.nz ? hiz : loz, fzd = r.nz ? loz : hiz;
				const int sh = child * 8;
				float e = fmaxf(fmaxf(__fmaf_rn((float)((nxd >> sh) & 0xFFu), Sx, Anx), __fmaf_rn((float)((nyd >> sh) & 0xFFu), Sy, Any)),
								__fmaf_rn((float)((nzd >> sh) & 0xFFu), Sz, Anz));
				const float x = fminf(fminf(__fmaf_rn((float)((fxd >> sh) & 0xFFu), Sx, Afx), __fmaf_rn((float)((fyd >> sh) & 0xFFu), Sy, Afy)),
									  __fmaf_rn((float)((fzd >> sh) & 0xFFu), Sz, Afz));
				e = fmaxf(e, 0.001f);
				code = child == 0 ? __float_as_int(w2.z) : child == 1 ? __float_as_int(w2.w) : child == 2 ? __float_as_int(w3.x) : __float_as_int(w3.y);
				h = (code != 0) & !(x < e) & !(e > best.t);
			}
			if (STATS) { nodeVisits += take; boxTests += 4 * take; }
			const unsigned long long mNode = __builtin_amdgcn_ballot_w64(h && code > 0);
			const unsigned long long mLeaf = __builtin_amdgcn_ballot_w64(h && code < 0);
			if (nStk + __popcll(mNode) > LONG_STACK) {
				aside = true;
				break;
			}
			if (h && code > 0) stk[nStk + __popcll(mNode & below)] = code;
			// a leaf is one record of sc.triWide or two consecutive ones (value = first record * 2 + 1 if two): the list
			// takes them apart, one triangle per entry
			const bool two = h && code < 0 && ((~code) & 1) != 0;
			const unsigned long long mTwo = __builtin_amdgcn_ballot_w64(two);
			if (h && code < 0) {
				const int at = nLeaf + __popcll(mLeaf & below) + __popcll(mTwo & below);
				const int rec = (~code) >> 1;
#ifdef RTX_BOUNDS_CHECK
				if ((unsigned)(rec + (two ? 1 : 0)) >= (unsigned)sc.nTris || at + 1 >= LONG_LEAVES) {
					printf("[rtx] k_trace_long: leaf record %d of %d at list slot %d\n", rec, sc.nTris, at);
					atomicOr(sc.fault, FAULT_LEAF);
				}
				else
#endif
				{
					lf[at] = rec;
					if (two) lf[at + 1] = rec + 1;
				}
			}
			nStk += __popcll(mNode);
			nLeaf += __popcll(mLeaf) + __popcll(mTwo);
			__builtin_amdgcn_wave_barrier();  // the next round's lanes read what other lanes have just written
		}
		if (aside) {
			// one list slot for the wave: lane 0 adds 1 and the others 0, so what lane 0 receives is the slot
			const unsigned got = atomicAdd(unsafeCount, lane == 0 ? 1u : 0u);
			unsafeList[(unsigned)__builtin_amdgcn_readfirstlane((int)got)] = j;
		}
		else
			put_hit<ANYHIT>(hits, j, found ? (ANYHIT ? 0 : best.rank) : -1, best.t, best.alpha, best.beta);
	}
	if (STATS && lane == 0) {
		atomicAdd(&stats[0], (unsigned long long)boxTests);
		atomicAdd(&stats[1], (unsigned long long)triTests);
		atomicAdd(&stats[4], (unsigned long long)nodeVisits);
		atomicAdd(&stats[5], (unsigned long long)leafFetches);
	}
}

}  // namespace
