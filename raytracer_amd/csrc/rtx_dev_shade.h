// Ray generation, shading, gather, accumulate, tone map and export kernels of rtx_hip.hip (included there).
#pragma once

#include "rtx_dev_common.h"

namespace {

// Camera::getRay (src/Camera.cpp:37-49) plus the per-sample RNG key.
__global__ __launch_bounds__(256) void k_raygen(DevScene sc, Batch b, float4* __restrict__ rayO,
												float4* __restrict__ rayD) {
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= b.nSamples) return;
	int x, y, sub;
	uint32_t frame;
	bool valid = sample_to_pixel(sc, b, j, x, y, sub, &frame);
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
// Shade.  One thread per ray of level `level`.  Writes the path-node record of this ray:
//   A[j]  leaf: its radiance, 0.   interior: triangle colour, scalar (replaced by the node's radiance in
//         k_gather); scalar = |n.d| (dielectric) or the refracted proportion (transparent)
//   B[j]  (first child slot, tag); tag = kind | flags | the texel's three bytes << 8 (textured dielectric; the
//         gather turns the bytes into the colour with the same table a texture look-up uses)
// and appends the node's child rays to the next level's queue.  cnt[0] receives the queue
// length, cnt[1] dead slots (refraction under total internal reflection), cnt[2] shaded hits,
// cnt[3] texel fetches.

// cs: the 512 (cos, sin) pairs of the reference's tables, which the block keeps in LDS (two scattered look-ups per
// direction there instead of four in global memory).
__device__ __forceinline__ void scatter_offset(const float2* __restrict__ cs, uint32_t key, uint32_t lobe, uint32_t k, float& ox,
											   float& oy, float& oz) {
	uint32_t phi = rt_draw(key, lobe, k, 0) & 0x1FFu;
	uint32_t theta = (rt_draw(key, lobe, k, 1) & 0x1FFu) >> 1;
	const float2 a = cs[phi], b = cs[theta];
	float cp = a.x, sp = a.y, ct = b.x, st = b.y;
	// horizontal * sin(theta) + cos(theta) * UnitZ, lane by lane (src/Triangle.cpp:107-113)
	ox = cp * st + ct * 0.0f;
	oy = sp * st + ct * 0.0f;
	oz = 0.0f * st + ct * 1.0f;
}

#ifndef RTX_SHADE_THREADS
#define RTX_SHADE_THREADS 512
#endif
// The waves of a block share one queue-tail atomic per round.  72 VGPRs allow 28 waves per CU: three blocks of 512
// threads (the others work while one waits for its atomic and its barriers) beat one block of 1024; smaller blocks pay
// more in atomics on the one tail than they gain.  The grid is a multiple of three blocks per CU.
constexpr int SHADE_THREADS = RTX_SHADE_THREADS;

__global__ __launch_bounds__(SHADE_THREADS) void k_shade(DevScene sc, int level, const float4* __restrict__ rayO,
											   const float4* __restrict__ rayD, const float4* __restrict__ hits, int n,
											   float4* __restrict__ A, int2* __restrict__ B,
											   float4* __restrict__ nRayO, float4* __restrict__ nRayD,
											   unsigned* __restrict__ cnt, unsigned nextCap, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	__shared__ unsigned s_tot[SHADE_THREADS / 64];
	__shared__ unsigned s_base[SHADE_THREADS / 64];
	__shared__ unsigned s_stat[3][SHADE_THREADS / 64];
	__shared__ float2 s_cs[512];  // (cos, sin) of the reference's 512-entry tables
	__shared__ float s_gamma[256];
	__shared__ int4 s_desc[LDS_TEX + 6];
	__shared__ float4 s_mat[2 * LDS_MATS];
	for (int i = threadIdx.x; i < 512; i += SHADE_THREADS) s_cs[i] = make_float2(sc.cosT[i], sc.sinT[i]);
	for (int i = threadIdx.x; i < 2 * LDS_MATS; i += SHADE_THREADS) s_mat[i] = i < 2 * sc.nMats ? sc.mats[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	lds_tables_fill(sc, s_gamma, s_desc, SHADE_THREADS);
	const LdsTables lt{s_gamma, s_desc};
	unsigned deadAll = 0, shadedAll = 0, fetchAll = 0;  // thread 0: the block's counts over all its rounds
	// The grid does not grow with the queue: a block walks over it in steps of the grid's width (every thread
	// of a block makes the same number of rounds, so the barriers below are uniform).
	for (long long base = (long long)blockIdx.x * SHADE_THREADS; base < n; base += (long long)gridDim.x * SHADE_THREADS) {
		const int j = (int)base + (int)threadIdx.x;
		const int S = sc.nSpec, Dn = sc.nDiff;
		const bool lastNext = level + 1 == sc.maxDepth;  // the children are rays of the last level
		int kind = KIND_LEAF;
		int flags = 0;
		int nchild = 0;
		float3 val = make_float3(0.0f, 0.0f, 0.0f);
		float scalar = 0.0f;
		unsigned texel = 0;  // the three bytes of the texel
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
					val = sky_fetch(sc, lt, dx, dy, dz);
					fetches = 1;
				}
				else
					val = make_float3(sc.bg.x, sc.bg.y, sc.bg.z);
			}
			else if ((unsigned)rank >= (unsigned)sc.nTris) {
				atomicOr(sc.fault, FAULT_HIT_RANK);  // not a hit record: worth nothing, reported to the host
			}
			else if (level < sc.maxDepth) {
				// (at level == maxDepth every hit returns zero, src/RayTracer.cpp:170; the any-hit
				//  traversal reports such hits with rank 0, which is never dereferenced)
				// one aligned 64-byte record per hit: the vertex normals, the texture coordinates and the number of the
				// triangle's material; the materials themselves (colour, roughness, index, texture, flags: what triangles
				// of one model share) are a small table, the first LDS_MATS of which the block keeps in LDS
				const float4* T = sc.triShade + (size_t)rank * 4;
				const float4 s0 = T[0], s1 = T[1], s2 = T[2], s3 = T[3];
				int matId = __float_as_int(s3.w);
				if ((unsigned)matId >= (unsigned)sc.nMats) {
					atomicOr(sc.fault, FAULT_MATERIAL);  // not a record of this scene: treated as the first material, reported to the host
					matId = 0;
				}
				float4 m0 = s_mat[2 * min(matId, LDS_MATS - 1)], m1 = s_mat[2 * min(matId, LDS_MATS - 1) + 1];
				if (matId >= LDS_MATS) {
					// (the empty asm keeps the two look-ups apart: merged, they become one slow load through an address of either kind)
					m0 = sc.mats[2 * (size_t)matId];
					m1 = sc.mats[2 * (size_t)matId + 1];
					asm volatile("" : "+v"(m0.x), "+v"(m0.y), "+v"(m0.z), "+v"(m0.w), "+v"(m1.x), "+v"(m1.y), "+v"(m1.z));
				}
				const float t = H.y, alpha = H.z, beta = H.w;
				const int triFlags = __float_as_int(m1.z);
				const int texId = __float_as_int(m1.y);
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
					val = make_float3(m0.x, m0.y, m0.z);
				}
				else {
					shaded = 1;
					val = make_float3(m0.x, m0.y, m0.z);  // colour, consumed by k_gather
					rough = m0.w;
					ior = m1.x;
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
						if (texId >= sc.nTex) atomicOr(sc.fault, FAULT_TEX_ID);
						else if (texId >= 0) {
							flags |= FLAG_TEX;
							float u = (alpha * s0.w + beta * s2.w) + g * s3.y;
							float v = (alpha * s1.w + beta * s3.x) + g * s3.z;
							texel = tex_fetch_rgb(sc, lt, texId, u, v);
							fetches = 1;
						}
					}
				}
			}
		}

		// ---- queue slots for the children.  Inside a wave, per-kind ballots + popcounts give each lane its
		// offset; the wave totals of the block are scanned in LDS and the block claims its range of the
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
			// the counters share a cache line with the queue tail, and one line takes only ~90 atomics per
			// microsecond: they are added once per block, after its last round
			deadAll += d;
			shadedAll += sh;
			fetchAll += f;
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
				scatter_offset(s_cs, key, 0, (uint32_t)i, ox, oy, oz);
				float cx = rx + ox * rough, cy = ry + oy * rough, cz = rz + oz * rough;
				nRayO[childBase + i] = make_float4(hx, hy, hz, __uint_as_float(rt_child_key(key, (uint32_t)i)));
				nRayD[childBase + i] = make_float4(cx, cy, cz, (lastNext && miss_is_zero(sc, cx, cy, cz)) ? 2.0f : 0.0f);
			}
			if (kind == KIND_DIEL) {
				// Triangle::diffuse (src/Triangle.cpp:95-119)
				bool keep = dot3w(dx, dy, dz, nx, ny, nz) < 0.0f;
				float tx = keep ? nx : -nx, ty = keep ? ny : -ny, tz = keep ? nz : -nz;
				for (int i = 0; i < Dn; ++i) {
					float ox, oy, oz;
					scatter_offset(s_cs, key, 1, (uint32_t)i, ox, oy, oz);
					nRayO[childBase + S + i] = make_float4(hx, hy, hz, __uint_as_float(rt_child_key(key, (uint32_t)(S + i))));
					const float ex = tx + ox, ey = ty + oy, ez = tz + oz;
					nRayD[childBase + S + i] = make_float4(ex, ey, ez, (lastNext && miss_is_zero(sc, ex, ey, ez)) ? 2.0f : 0.0f);
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
					nRayD[childBase + S] = make_float4(ex, ey, ez, (lastNext && miss_is_zero(sc, ex, ey, ez)) ? 2.0f : 0.0f);
				}
			}
		}
		if (active) {
			A[j] = make_float4(val.x, val.y, val.z, scalar);
			B[j] = make_int2((int)childBase, kind | flags | (int)(texel << 8));
		}
		__syncthreads();  // the LDS words are reused by the next round
	}
	if (threadIdx.x == 0) {
		if (deadAll) atomicAdd(&cnt[1], deadAll);
		if (shadedAll) atomicAdd(&cnt[2], shadedAll);
		if (fetchAll) atomicAdd(&cnt[3], fetchAll);
	}
}

// The rays of the last recursion level end their paths whatever they hit: a hit returns zero
// (src/RayTracer.cpp:170), a miss the sky or the background (:148-152).  This is that value, from the ray's
// any-hit answer; -2 marks a slot without a ray, -3 a ray that was not traced because both outcomes are zero.
__device__ __forceinline__ float3 last_level_value(const DevScene& sc, const LdsTables& lt, int code, const float4* __restrict__ rayD, unsigned slot,
												   unsigned& dead, unsigned& fetches, unsigned& untraced) {
	float3 val = make_float3(0.0f, 0.0f, 0.0f);
	if (code == -2) dead++;
	else if (code == -3) untraced++;
	else if (code == -1) {
		if (sc.hasSky) {
			const float4 D = rayD[slot];
			val = sky_fetch(sc, lt, D.x, D.y, D.z);
			fetches++;
		}
		else
			val = make_float3(sc.bg.x, sc.bg.y, sc.bg.z);
	}
	return val;
}

// Combine step of RayTracer::color for the nodes of one level, children read from the next
// level (src/RayTracer.cpp:177-224).  Sums start from zero and run in child order, scale
// factors are applied after the sum, exactly as the reference does.
// LAST: the children are the rays of the last level and their values are worked out here, from the any-hit
// answers (`codes`) and the rays' directions, instead of being read from childA; cnt is then the counter block
// the last level's shade kernel would have written (dead slots, untraced rays, texel fetches).
template <bool LAST>
__global__ __launch_bounds__(256) void k_gather(DevScene sc, int n, float4* __restrict__ A, const int2* __restrict__ B,
												const float4* __restrict__ childA,
												const int* __restrict__ codes, const float4* __restrict__ childD,
												unsigned childCap, unsigned* __restrict__ cnt, const unsigned* __restrict__ nDev) {
	if (nDev) n = (int)min(*nDev, (unsigned)n);  // n: capacity of the arrays
	const int S = sc.nSpec, Dn = sc.nDiff;
	__shared__ float s_gamma[256];
	__shared__ int4 s_desc[LDS_TEX + 6];
	lds_tables_fill(sc, s_gamma, s_desc, 256);
	const LdsTables lt{s_gamma, s_desc};
	unsigned dead = 0, fetches = 0, untraced = 0;
	for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
		const int2 b = B[j];
		const int kind = b.y & 0xF;
		if (kind == KIND_LEAF) continue;
		const float4 col = A[j];
		const unsigned first = (unsigned)b.x;
		const unsigned nchild = (unsigned)(kind == KIND_METAL ? S : kind == KIND_TRANSP ? S + 1 : S + Dn);
		if (first > childCap || nchild > childCap - first) {
			atomicOr(sc.fault, FAULT_CHILD_SLOT);  // not a node record: left as it is, reported to the host
			continue;
		}
		float sx = 0.0f, sy = 0.0f, sz = 0.0f;
		for (int i = 0; i < S; ++i) {
			float3 c;
			if (LAST) c = last_level_value(sc, lt, codes[first + i], childD, first + i, dead, fetches, untraced);
			else {
				const float4 v = childA[first + i];
				c = make_float3(v.x, v.y, v.z);
			}
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
			const float rp = col.w;
			float3 rc = make_float3(0.0f, 0.0f, 0.0f);
			if (b.y & FLAG_TIR) {
				if (LAST) dead++;  // the slot of the refracted ray that does not exist
			}
			else if (LAST) rc = last_level_value(sc, lt, codes[first + S], childD, first + S, dead, fetches, untraced);
			else {
				const float4 v = childA[first + S];
				rc = make_float3(v.x, v.y, v.z);
			}
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
				float3 c;
				if (LAST) c = last_level_value(sc, lt, codes[first + S + i], childD, first + S + i, dead, fetches, untraced);
				else {
					const float4 v = childA[first + S + i];
					c = make_float3(v.x, v.y, v.z);
				}
				dx = dx + c.x;
				dy = dy + c.y;
				dz = dz + c.z;
			}
			const float fD = (float)Dn;
			dx = (dx * col.x) / fD;
			dy = (dy * col.y) / fD;
			dz = (dz * col.z) / fD;
			const float ad = col.w;
			ox = sx + dx * ad;
			oy = sy + dy * ad;
			oz = sz + dz * ad;
			if (b.y & FLAG_TEX) {
				const unsigned rgb = (unsigned)b.y >> 8;
				ox = ox * lt.gamma[rgb & 0xFFu];
				oy = oy * lt.gamma[(rgb >> 8) & 0xFFu];
				oz = oz * lt.gamma[rgb >> 16];
			}
		}
		A[j] = make_float4(ox, oy, oz, 0.0f);
	}
	if (LAST) {
		// one set of atomics per block (the three counters lie in one cache line, which takes only ~90 atomics per
		// microsecond: one per wave of a large grid would cost more than the gather itself)
		__shared__ unsigned s_red[3][256 / 64];
		for (int off = 32; off > 0; off >>= 1) {
			dead += __shfl_down(dead, off, 64);
			fetches += __shfl_down(fetches, off, 64);
			untraced += __shfl_down(untraced, off, 64);
		}
		if ((threadIdx.x & 63) == 0) {
			s_red[0][threadIdx.x >> 6] = dead;
			s_red[1][threadIdx.x >> 6] = fetches;
			s_red[2][threadIdx.x >> 6] = untraced;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned d = 0, f = 0, u = 0;
			for (int w = 0; w < 256 / 64; ++w) {
				d += s_red[0][w];
				f += s_red[1][w];
				u += s_red[2][w];
			}
			if (d) atomicAdd(&cnt[1], d);
			if (f) atomicAdd(&cnt[3], f);
			if (u) atomicAdd(&cnt[2], u);
		}
	}
}

// accumulateImg(row,col) += (c0+c1+c2+c3) * 0.25f (src/RayTracer.cpp:242-246).  One thread per pixel.
// `b` must lie within one frame: the host launches this once per frame a batch touches, in frame order, so a
// pixel receives its frames in the order of the reference's loop (src/RayTracer.cpp:234).
// True when some level's child queue was too small for this batch (slot 4 of the counter blocks): the batch's
// result is then incomplete, is not stored, and the host renders the batch again with worst-case queues.
__device__ __forceinline__ bool batch_overflowed(const unsigned* __restrict__ counts, int maxDepth) {
	unsigned f = 0;
	for (int lv = 1; lv <= maxDepth + 1; ++lv) f |= counts[lv * CNT_STRIDE + 4];
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

// Exchange of tile accumulators between GPUs.  A share is a list of tile numbers (row-major 64 x 64 tiles); slot k of
// the packed array holds tile ids[k] as TILE x TILE float4 in row-major order (pixels beyond the image: zero).
// k_pack_tiles copies a share out of an image -- and, with `clear`, leaves zeros behind, so that an image that
// collects one frame's contributions is empty again for the next frame; k_unpack_tiles copies one into an image, or
// with `add` adds it (x = x + packed: what k_accumulate would have done had the frame been rendered into that
// image directly, bit for bit).  One thread per pixel slot.
__global__ __launch_bounds__(256) void k_pack_tiles(float4* __restrict__ accum, int width, int height, int tilesX, const int* __restrict__ ids,
													int slots, int numTiles, float4* __restrict__ packed, int clear) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= slots * TILE * TILE) return;
	const int slot = i / (TILE * TILE), p = i - slot * (TILE * TILE);
	const int tile = ids[slot];
	float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if ((unsigned)tile < (unsigned)numTiles) {
		const int x = (tile % tilesX) * TILE + (p % TILE), y = (tile / tilesX) * TILE + p / TILE;
		if (x < width && y < height) {
			float4* px = accum + (size_t)y * width + x;
			v = *px;
			if (clear) *px = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		}
	}
	packed[i] = v;
}

__global__ __launch_bounds__(256) void k_unpack_tiles(float4* __restrict__ accum, int width, int height, int tilesX, const int* __restrict__ ids,
													  int slots, int numTiles, const float4* __restrict__ packed, int add) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= slots * TILE * TILE) return;
	const int slot = i / (TILE * TILE), p = i - slot * (TILE * TILE);
	const int tile = ids[slot];
	if ((unsigned)tile >= (unsigned)numTiles) return;
	const int x = (tile % tilesX) * TILE + (p % TILE), y = (tile / tilesX) * TILE + p / TILE;
	if (x >= width || y >= height) return;
	float4* px = accum + (size_t)y * width + x;
	const float4 v = packed[i];
	if (add) {
		float4 a = *px;
		a.x = a.x + v.x;
		a.y = a.y + v.y;
		a.z = a.z + v.z;
		a.w = a.w + v.w;
		*px = a;
	}
	else
		*px = v;
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
	if (rank >= sc.nTris) {
		atomicOr(sc.fault, FAULT_HIT_RANK);
		rank = -1;
	}
	ids[o] = rank >= 0 ? __float_as_int(sc.triHit[(size_t)rank * 4 + 3].x) : -1;
	if (tab) {
		tab[o * 3 + 0] = h.y;
		tab[o * 3 + 1] = h.z;
		tab[o * 3 + 2] = h.w;
	}
}

// xyz triples -> the 16-byte ray records the traversal reads (w = 0: a ray to be traced, no RNG key).
__global__ __launch_bounds__(256) void k_pack_rays(int n, const float* __restrict__ o, const float* __restrict__ d,
												   float4* __restrict__ rayO, float4* __restrict__ rayD) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	rayO[j] = make_float4(o[3 * (size_t)j], o[3 * (size_t)j + 1], o[3 * (size_t)j + 2], 0.0f);
	rayD[j] = make_float4(d[3 * (size_t)j], d[3 * (size_t)j + 1], d[3 * (size_t)j + 2], 0.0f);
}

__global__ __launch_bounds__(256) void k_export_hits_linear(DevScene sc, int n, const float4* __restrict__ hits,
															int* __restrict__ ids, float* __restrict__ tab) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	float4 h = hits[j];
	int rank = __float_as_int(h.x);
	if (rank >= sc.nTris) {
		atomicOr(sc.fault, FAULT_HIT_RANK);
		rank = -1;
	}
	ids[j] = rank >= 0 ? __float_as_int(sc.triHit[(size_t)rank * 4 + 3].x) : -1;
	if (tab) {
		tab[(size_t)j * 3 + 0] = h.y;
		tab[(size_t)j * 3 + 1] = h.z;
		tab[(size_t)j * 3 + 2] = h.w;
	}
}

}  // namespace
