// Device-side scene description, the counter-based random numbers, texture and sky look-ups, and the mapping of
// queue slots to pixels shared by the kernels of rtx_hip.hip (included there; not a stand-alone header).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace {

// ------------------------------------------------------------------------------------------------
// Device-side scene description, passed to kernels by value.

constexpr int TILE = 64;            // image tile edge in pixels (sharding granule)
constexpr int BLK = 4;              // a wave covers a BLK x BLK pixel block x 4 sub-samples
constexpr int BLOCKS_PER_TILE = (TILE / BLK) * (TILE / BLK);
constexpr int SAMPLES_PER_BLOCK = BLK * BLK * 4;  // 64 = one wave
#ifndef RTX_TRACE_THREADS
#define RTX_TRACE_THREADS 256
#endif
constexpr int TRACE_THREADS = RTX_TRACE_THREADS;  // threads per block of the traversal kernels (experiments: -DRTX_TRACE_THREADS=64 or 128)
// Counter block of a recursion level (unsigned words, zeroed at the start of a pixel batch): 0 queue length,
// 1 dead slots, 2 shaded hits, 3 texel fetches, 4 overflow flag, 5-9 cursors and list lengths of the traversal.
constexpr int CNT_STRIDE = 16;

enum NodeKind : int { KIND_LEAF = 0, KIND_METAL = 1, KIND_TRANSP = 2, KIND_DIEL = 3 };
// a node's tag word: kind in bits 0-3, flags in bits 4-5, the texel's three bytes in bits 8-31
constexpr int FLAG_TEX = 0x10;
constexpr int FLAG_TIR = 0x20;

constexpr int TRI_EMIT = 1, TRI_METAL = 2, TRI_TRANSP = 4;

struct DevScene {
	const float4* pairs;     // 4 x float4 per child pair: Lmin|codeL, Lmax|-, Rmin|codeR, Rmax|-
	const float4* wide;      // 4 x float4 per node of the 4-wide quantised collapse of the tree
	const float4* triHit;    // 4 x float4 per triangle in candidate-rank order: n, v0, v1, v2, index
	const float4* triWide;   // the same records in the order of the wide tree's leaves (a leaf's two triangles side by
	                         // side), fourth quad: index, rank, -, -
	const float4* triShade;  // 4 x float4 per triangle in candidate-rank order: n0 uv0.x | n1 uv0.y | n2 uv1.x | uv1.y uv2.x uv2.y material
	const float4* mats;      // 2 x float4 per material (the distinct colour / roughness / index / texture / flags of the triangles):
	                         // r g b roughness | refractive index, texture id, flags, -
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
	int nWide, nTris, nTex, nMats;  // record counts of `wide`, of the triangle arrays, of texDesc and of mats
	unsigned* fault;         // one word: a kernel that meets an index outside those counts sets a bit and skips
};

// Bits of DevScene::fault.  An index is checked wherever it comes out of a queue record and is about to address
// scene data: a record nobody wrote must end in an error code, never in an access outside the arrays.
constexpr unsigned FAULT_HIT_RANK = 1u, FAULT_TEX_ID = 2u, FAULT_CHILD_SLOT = 4u, FAULT_NODE = 8u, FAULT_LEAF = 16u,
				   FAULT_STACK = 32u, FAULT_LONG_ROUNDS = 64u, FAULT_LIST_ENTRY = 128u, FAULT_MATERIAL = 256u;

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

// Small read-only tables a block keeps in LDS: the 256-entry texel decode (powf(c/255, 2.2)) and the descriptors of
// the first LDS_TEX textures.  Every look-up in them is a scattered access of a whole wave; in LDS it costs no trip to
// the vector-memory path, which the shading kernels are bound by.
constexpr int LDS_TEX = 64;
constexpr int LDS_MATS = 128;  // materials the shade kernel keeps in LDS (4 KB); a scene with more reads the others from global memory
struct LdsTables {
	const float* gamma;   // 256 floats
	const int4* desc;     // LDS_TEX descriptors: those of the first min(nTex, LDS_TEX) textures, then the six sky faces' at LDS_SKY
};
constexpr int LDS_SKY = LDS_TEX;  // desc[LDS_SKY + face]: the descriptor of a sky face, without the detour over its texture id
// Fills the block's tables (every thread of the block calls it; ends with a barrier).  desc: LDS_TEX + 6 entries.
__device__ __forceinline__ void lds_tables_fill(const DevScene& sc, float* gamma, int4* desc, int nThreads) {
	for (int i = threadIdx.x; i < 256; i += nThreads) gamma[i] = sc.gammaT[i];
	for (int i = threadIdx.x; i < min(sc.nTex, LDS_TEX); i += nThreads) desc[i] = sc.texDesc[i];
	if (threadIdx.x < 6) desc[LDS_SKY + threadIdx.x] = sc.hasSky ? sc.texDesc[sc.skyTex[threadIdx.x]] : make_int4(0, 1, 1, 0);
	__syncthreads();
}

// The texel as raw bytes (r | g << 8 | b << 16).  The three bytes are taken with one four-byte load at the texel's
// (unaligned) address -- global memory takes unaligned loads, and one scattered load per lane costs a third of three;
// the texel blob is padded so that the fourth byte of its last texel exists.
__device__ __forceinline__ unsigned texel_rgb(const DevScene& sc, int4 d, float u, float v) {
	u = clamp01(u);
	v = clamp01(v);
	float uMax = (float)(d.y - 1);
	float vMax = (float)(d.z - 1);
	int sx = (int)roundf(u * uMax);
	int sy = (int)roundf(v * vMax);
	const uint8_t* c = sc.texels + (size_t)(uint32_t)d.x + ((size_t)sy * (size_t)d.y + (size_t)sx) * 3;
	unsigned w;
	__builtin_memcpy(&w, c, 4);
	return w & 0x00FFFFFFu;
}
__device__ __forceinline__ unsigned tex_fetch_rgb(const DevScene& sc, const LdsTables& lt, int id, float u, float v) {
	int4 d = lt.desc[min(id, LDS_TEX - 1)];
	if (id >= LDS_TEX) {
		// a scene with more textures than the table holds: the descriptor comes from global memory (the empty asm keeps
		// the two look-ups apart: merged, they become one slow load through an address of either kind)
		d = sc.texDesc[id];
		asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
	}
	return texel_rgb(sc, d, u, v);
}

// Texture::sampleTexture (src/Texture.cpp:41-53) with clamped coordinates and the
// powf(c/255, 2.2) decode taken from the host-computed table.
__device__ __forceinline__ float3 tex_fetch(const DevScene& sc, const LdsTables& lt, int id, float u, float v) {
	const unsigned rgb = tex_fetch_rgb(sc, lt, id, u, v);
	return make_float3(lt.gamma[rgb & 0xFFu], lt.gamma[(rgb >> 8) & 0xFFu], lt.gamma[rgb >> 16]);
}

// Face selection of Skybox::sampleBackground (src/Skybox.cpp:32-84): the first of front, back, left, right, top,
// bottom whose divisor is not negative and whose (u, v) lies in [0,1]^2; -1 when none does (the reference then
// returns zero; that is the case for every direction in the pyramid around -y, since the bottom face's v can
// never pass).
__device__ __forceinline__ int sky_face(float dx, float dy, float dz, float& u, float& v) {
	for (int i = 0; i < 6; ++i) {
		float divisor;
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
		return i;
	}
	return -1;
}

// Skybox::sampleBackground (src/Skybox.cpp:32-84).
__device__ __forceinline__ float3 sky_fetch(const DevScene& sc, const LdsTables& lt, float dx, float dy, float dz) {
	float u, v;
	const int face = sky_face(dx, dy, dz, u, v);
	if (face < 0) return make_float3(0.0f, 0.0f, 0.0f);
	const unsigned rgb = texel_rgb(sc, lt.desc[LDS_SKY + face], u, v);
	float3 c = make_float3(lt.gamma[rgb & 0xFFu], lt.gamma[(rgb >> 8) & 0xFFu], lt.gamma[rgb >> 16]);
	return make_float3(c.x * sc.skyBrightness, c.y * sc.skyBrightness, c.z * sc.skyBrightness);
}

// What a ray of the last recursion level is worth is known without tracing it when a miss would return zero:
// a hit returns zero there anyway (src/RayTracer.cpp:170).  True when the sky has no face for the direction,
// or when there is no sky and the background is black.
__device__ __forceinline__ bool miss_is_zero(const DevScene& sc, float dx, float dy, float dz) {
	if (sc.hasSky) {
		float u, v;
		return sky_face(dx, dy, dz, u, v) < 0;
	}
	return sc.bg.x == 0.0f && sc.bg.y == 0.0f && sc.bg.z == 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Level-0 sample order.  Sample j of a batch that starts at pixel block g0:
//   block G = g0 + j/64 -> (frame G / blocksPerFrame, block g = G % blocksPerFrame of the scene's tile list)
//   g -> (tile slot, block in tile); lane = j%64 -> (pixel in block, sub-sample).
// A wave therefore holds the 4 sub-samples of a 4x4 pixel block, which keeps primary rays of
// one wave spatially adjacent.  Tiles are taken from the arithmetic list first + k*stride.
// Consecutive frames follow one another in the block numbering, so one batch may hold pixel blocks of
// several frames: a small tile share then still fills the queues.

struct Batch {
	int g0;              // first pixel block (numbered over the frames of the call, then over the scene's tile list)
	int nSamples;        // (number of blocks) * 64
	int tileFirst, tileStride;
	int blocksPerFrame;  // pixel blocks of the scene's tile list
	uint32_t frame0;     // the frame that block 0 belongs to
};

__device__ __forceinline__ bool sample_to_pixel(const DevScene& sc, const Batch& b, int j, int& x, int& y, int& sub,
												uint32_t* frame = nullptr) {
	const int G = b.g0 + (j >> 6);
	const int f = G / b.blocksPerFrame;
	int g = G - f * b.blocksPerFrame;
	if (frame) *frame = b.frame0 + (uint32_t)f;
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

}  // namespace
