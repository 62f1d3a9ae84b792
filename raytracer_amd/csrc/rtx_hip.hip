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
// Traversal (rtx_dev_trace.h): for a ray with finite reciprocals the reference's candidate list is exactly
// the triangles whose own bounding box passes its slab predicate (src/BVH.cpp:14-41), in its depth-first
// (right child first) leaf order.  The kernels therefore walk a 4-wide quantised tree of their own
// (rtx_wide.h), test every leaf with the exact predicate and the exact Triangle::hit, skip boxes that lie
// wholly behind t = 0.001 or beyond the best hit (judged on boxes inflated by a position-space margin) and
// resolve equal-t ties by the precomputed leaf order.  The few rays that argument does not cover are traced
// through the reference's own tree.
//
// Files: rtx_dev_common.h (scene description, RNG, texture/sky), rtx_dev_trace.h (traversal kernels),
// rtx_dev_shade.h (ray generation, shade, gather, accumulate, tone map, exports), this file (host side and
// the C ABI), rtx_wide.h (construction of the traversal tree).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened when rtx_gather_tiles is first called
#include <dlfcn.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
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

}  // namespace

#include "rtx_dev_common.h"
#include "rtx_dev_trace.h"
#include "rtx_dev_shade.h"

// ------------------------------------------------------------------------------------------------
// Scratch memory of rtx_trace_rays_noalloc, the one entry point that allocates nothing: ray records, hit
// records, the two ray lists of the traversal and its counter block live in this file-scope array instead of in
// buffers from hipMalloc.  Every GPU the library runs on holds its own copy of the array until the process ends.
// In host code the array's name is only a handle for hipGetSymbolAddress: the address a kernel may use is the
// one that call returns, per device (scratch_for_device below), and host code passes that pointer and never
// the name.
constexpr size_t NOALLOC_COUNT_BYTES = 4096;                               // counter block of one traversal + fault word
constexpr size_t NOALLOC_BYTES_PER_RAY = 16 + 16 + 16 + 8;                 // origin, direction, hit record, two list slots
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_COUNT_BYTES + (size_t)RTX_NOALLOC_MAX_RAYS * NOALLOC_BYTES_PER_RAY;  // 14 MiB + 4 KiB
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratchMutex;
unsigned char* g_scratchPtr[MAX_DEVICES] = {nullptr};

// The device address of `scratch` on `device`, which must be the current device.  Looked up once per device.
int scratch_for_device(int device, unsigned char** out) {
	if (device < 0 || device >= MAX_DEVICES) return fail(RTX_ERR_ARG, "device id beyond the library's table of scratch arrays");
	std::lock_guard<std::mutex> lk(g_scratchMutex);
	if (!g_scratchPtr[device]) {
		void* p = nullptr;
		HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		if (!p) return fail(RTX_ERR_DEVICE, "hipGetSymbolAddress returned no address for the scratch array");
		g_scratchPtr[device] = (unsigned char*)p;
	}
	*out = g_scratchPtr[device];
	return RTX_OK;
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
		if (!count) return hipSuccess;
		const hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
		if (e == hipSuccess) n = count;  // a failed allocation leaves an empty buffer, not a size without memory
		else p = nullptr;
		return e;
	}
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		n = 0;
	}
};

struct EventPair {
	hipEvent_t a, b;
	int kind;    // 0 trace, 1 shade/gather/accumulate, 2 whole call
	int level;   // path level of a trace launch
	bool ended;    // b has been recorded
	bool sharedA;  // a is the b of the pair recorded just before: it is not this pair's to give back
};

}  // namespace

struct rtx_scene {
	int device = 0;
	int numCUs = 256;
	int longBlocksPerCU = 8;         // grid of k_trace_long (RTX_LONG_BLOCKS)
	int smallQueueFactor = 5;        // queues of up to this many rays per wave of k_trace_long go there entirely (RTX_SMALL_QUEUE): 40,960 rays, where a launch of
	                                 // the one-ray-per-lane kernel (about 0.1 ms whatever the queue) starts to be the cheaper one
	int longVisits = 64;             // nodes one lane walks for a ray before a whole wave takes the ray over (RTX_LONG_VISITS, 0 = never)
	hipEvent_t lastEnd = nullptr;    // the event recorded by the last end_event, and whether nothing has been launched since
	bool lastEndFresh = false;
	int shadeBlocksPerCU = 6;        // grid of k_shade (RTX_SHADE_BLOCKS_PER_CU): a multiple of the three blocks of 512 threads a CU holds at once, so that
	                                 // the last set of blocks is a full one (four per CU: a third of the chip works through the last quarter)
	int retireBacklog = 128;         // rays left per remaining wave below which a wave of k_trace_wide takes its last chunk (RTX_RETIRE_BACKLOG; 0: never)
	int tailLanes = 2;               // a wave down to this many rays after the queue's end hands them over at once (RTX_TAIL_LANES)
	int tailVisits = 32;             // the same allowance for rays still in flight once a wave finds the queue exhausted (RTX_TAIL_VISITS)
	long long mergeBlocks = 524288;  // frames of a multi-frame call share a batch up to this many 4x4-pixel blocks (RTX_MERGE_BLOCKS): four 1080p frames or one
	                                 // 3840x2160 frame; every level's launch ends with its waves draining their last rays, about 0.25 ms whatever the queue holds
	uint8_t* const* toneOut = nullptr;  // rtx_render_frames_tonemapped: where the 8-bit image after frame toneOutFirst + k goes (host memory), k < toneOutCount
	uint32_t toneOutFirst = 0, toneOutCount = 0;
	int traceRefill = TRACE_REFILL, traceChunk = TRACE_CHUNK, traceBlocksPerCU = 6, traceTriBatch = 28;  // tunables (RTX_TRACE_* env)
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
	size_t defaultBudget = 0;  // the default arena budget as last worked out from the free device memory (0: not yet)
	bool instrumented = false;
	// host copies
	std::vector<HostTri> tris;
	int maxTriTex = -1;  // the largest texture id a triangle names (checked against the textures at every call)
	std::vector<uint8_t> texels;
	std::vector<TexInfo> texInfo;
	std::vector<int> rankOf;  // triangle index -> candidate rank
	float sceneAbsMax = 0.0f;
	int treeDepth = 0;
	int stackNeed = 32;  // deepest interior node: the traversal pushes at most once per interior level
	int wideStackNeed = 52;  // wide traversal: entries waiting when every child of every node on a path is hit
	int wideStackCap = 16;   // RTX_WIDE_STACK: 16 or 24 entries (+ 1 row) of the wide traversal's stack
	int wideBlocksPerCU = 7; // grid of k_trace_wide in blocks per CU (RTX_TRACE_BLOCKS_PER_CU; six at most with the 24-entry stack)
	// Rays per primary sample at each level: popRatio the largest seen so far (0: nothing seen yet), capRatio what
	// the queues of the next batch are sized for.  Levels up to capKnown rest on counts a batch has reported, the
	// deeper ones on a guess.
	double popRatio[17] = {0};
	double capRatio[17] = {0};
	int capKnown = 0;
	bool haveCaps = false;
	bool syncLaunches = false;     // RTX_SYNC_LAUNCHES=1: wait for every kernel and name the one that fails (debugging)
	bool worstCaseQueues = false;  // RTX_WORST_CASE_QUEUES=1: queues for the theoretical worst case, always
	bool wideOk = false;     // the tree satisfies what k_trace_wide relies on
	bool forceBinary = false;  // RTX_TRACE_BINARY=1: binary-tree kernel only (for comparisons)
	uint32_t numWide = 0;
	uint32_t numPairs = 0;
	// device
	DevBuf<float4> dPairs, dWide, dTriHit, dTriWide, dTriShade, dMats;
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
	DevBuf<int> dExportIds;     // rtx_primary_hits: triangle ids of the image's samples, kept from call to call
	DevBuf<float> dExportTab;   // rtx_primary_hits: t, alpha, beta of the same samples
	DevBuf<float4> dExportRad;  // rtx_sample_radiance: per-sample radiance of the image
	DevBuf<float4> dRaysO, dRaysD, dRaysH;  // rtx_trace_rays: ray and hit records, kept from call to call
	DevBuf<int> dRaysList;                  // ... and the two ray lists of the traversal
	DevBuf<float4> dTilePack;   // tile exchange: this scene's share, packed (the share that is sent)
	DevBuf<float4> dTileRecv;   // tile exchange, scene 0: the shares of the others as they arrive
	DevBuf<int> dTileIds;       // tile exchange: the tile numbers of this scene's share (what k_pack_tiles walks over)
	DevBuf<int> dTileIdsRecv;   // tile exchange, scene 0: the tile numbers of every sender's share, in the order of the receive buffer
	int* hTileIds = nullptr;    // pinned staging of the two id arrays (own share first, then what scene 0 receives)
	size_t hTileIdsCap = 0;
	hipEvent_t exchEvent = nullptr;  // "this scene's share is packed", for a receiver on another stream of the same GPU
	std::vector<uint8_t> tileLog;    // per tile: rendered into the accumulator since the log was last cleared (rtx_exchange_tiles)
	DevBuf<uint8_t> arena;
	DevBuf<unsigned> dCounts;           // 8 per level
	DevBuf<unsigned long long> dStats;  // node tests, tri tests
	unsigned* hCounts = nullptr;        // pinned
	unsigned long long* hStats = nullptr;
	std::vector<EventPair> events;       // pairs recorded since the last harvest
	std::vector<hipEvent_t> eventPool;   // events waiting to be used again (created on demand, destroyed with the scene)
	unsigned char* noallocScratch = nullptr;  // this device's copy of the file-scope scratch array
	unsigned* hNoallocFault = nullptr;        // pinned word (obtained once, at rtx_scene_create) the fault word of rtx_trace_rays_noalloc is copied into
	bool timeEvents = true;              // false: kernels are launched without timing events (the entry point that allocates nothing)
	rtx_stats stats;

	~rtx_scene() {
		(void)hipSetDevice(device);
		dPairs.release(); dWide.release(); dTriHit.release(); dTriWide.release(); dTriShade.release(); dMats.release(); dToneBits.release(); dToneVal.release(); dRgb8.release(); dTexels.release(); dTexDesc.release();
		dTables.release(); dAccumOwn.release(); dTilePack.release(); dTileRecv.release(); arena.release(); dCounts.release(); dStats.release();
		if (hCounts) (void)hipHostFree(hCounts);
		if (hStats) (void)hipHostFree(hStats);
		if (hNoallocFault) (void)hipHostFree(hNoallocFault);
		if (hTileIds) (void)hipHostFree(hTileIds);
		if (exchEvent) (void)hipEventDestroy(exchEvent);
		dTileIds.release(); dTileIdsRecv.release(); dExportIds.release(); dExportTab.release(); dExportRad.release(); dRaysO.release(); dRaysD.release(); dRaysH.release(); dRaysList.release();
		for (auto& e : events) {
			(void)hipEventDestroy(e.a);
			(void)hipEventDestroy(e.b);
		}
		for (hipEvent_t e : eventPool) (void)hipEventDestroy(e);
		if (ownStream) (void)hipStreamDestroy(ownStream);
	}
};

namespace {

constexpr int MAX_DEPTH = 15;
constexpr int STATS_WORDS = 24;  // test counters of the instrumented traversal kernels (rtx_scene_set_instrumented)
constexpr int FAULT_WORD = (MAX_DEPTH + 2) * CNT_STRIDE;  // behind the level blocks: the kernels' consistency checks (DevScene::fault)
constexpr int COUNT_WORDS = FAULT_WORD + CNT_STRIDE;

// The kernels' consistency word (DevScene::fault) as read back after the stream has drained.
int check_fault(unsigned bits) {
	if (!bits) return RTX_OK;
	return fail(RTX_ERR_DEVICE, "device consistency check failed (fault bits " + std::to_string(bits) +
									"): a queue record held an index outside the scene; the image of this call is not valid");
}

// After every kernel launch: the launch error, and with RTX_SYNC_LAUNCHES=1 (debugging) the kernel's own
// completion, reported under the kernel's name.
int launched(rtx_scene* s, const char* kernel) {
	s->lastEndFresh = false;
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && s->syncLaunches) e = hipStreamSynchronize(s->stream);
	if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string(kernel) + ": " + hipGetErrorString(e));
	return RTX_OK;
}
#define RTX_LAUNCHED(kernel)                \
	do {                                    \
		const int rcl_ = launched(s, kernel); \
		if (rcl_) return rcl_;              \
	} while (0)

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
	d.triWide = s->dTriWide.p;
	d.triShade = s->dTriShade.p;
	d.mats = s->dMats.p;
	d.nMats = (int)(s->dMats.n / 2);
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
	d.nWide = (int)s->numWide;
	d.nTris = (int)s->tris.size();
	d.nTex = (int)s->texInfo.size();
	d.fault = s->dCounts.p + FAULT_WORD;
	return d;
}

int upload_textures(rtx_scene* s) {
	if (!s->texDirty) return RTX_OK;
	HIPCHK(s->dTexels.alloc(s->texels.size() + 16));  // a texel is read with a four-byte load: the last one's fourth byte must exist
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
	if (s->maxTriTex >= ntex) return fail(RTX_ERR_ARG, "a triangle refers to a texture id that was never added");
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
		HIPCHK(s->dCounts.alloc(COUNT_WORDS));
		HIPCHK(s->dStats.alloc(STATS_WORDS));
		HIPCHK(hipHostMalloc((void**)&s->hCounts, COUNT_WORDS * sizeof(unsigned), hipHostMallocDefault));
		HIPCHK(hipHostMalloc((void**)&s->hStats, STATS_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
	}
	return RTX_OK;
}

constexpr size_t NO_EVENT = ~(size_t)0;

int take_event(rtx_scene* s, hipEvent_t& e) {
	if (!s->eventPool.empty()) {
		e = s->eventPool.back();
		s->eventPool.pop_back();
		return RTX_OK;
	}
	HIPCHK(hipEventCreate(&e));
	return RTX_OK;
}

// Timing events come from a pool that lives as long as the scene: a launch inside the timed region costs two
// hipEventRecord calls and nothing else.
int begin_event(rtx_scene* s, int kind, size_t& idx, int level = 0) {
	idx = NO_EVENT;
	if (!s->timeEvents) return RTX_OK;
	EventPair e;
	// Where one timed group ends and the next begins with nothing launched in between, the end of the one is the beginning
	// of the other: a record in the queue costs about 5 us between two kernels, which a frame of a small tile share or of a
	// small scene pays twenty times.
	e.sharedA = s->lastEndFresh;
	int rc = RTX_OK;
	if (e.sharedA) e.a = s->lastEnd;
	else rc = take_event(s, e.a);
	if (rc) return rc;
	rc = take_event(s, e.b);
	if (rc) {
		if (!e.sharedA) s->eventPool.push_back(e.a);
		return rc;
	}
	e.kind = kind;
	e.level = level;
	e.ended = false;
	s->events.push_back(e);
	idx = s->events.size() - 1;
	if (!e.sharedA) HIPCHK(hipEventRecord(e.a, s->stream));
	return RTX_OK;
}
int end_event(rtx_scene* s, size_t idx) {
	if (idx == NO_EVENT) return RTX_OK;
	HIPCHK(hipEventRecord(s->events[idx].b, s->stream));
	s->events[idx].ended = true;
	s->lastEnd = s->events[idx].b;
	s->lastEndFresh = true;
	return RTX_OK;
}
// Puts every recorded pair back into the pool without reading it: what an error path and the start of a call do,
// so that a pair whose second event was never recorded cannot poison a later harvest.
void drop_events(rtx_scene* s) {
	for (auto& e : s->events) {
		if (!e.sharedA) s->eventPool.push_back(e.a);
		s->eventPool.push_back(e.b);
	}
	s->events.clear();
	s->lastEndFresh = false;
}
// After a stream sync: fold event durations into the stats and recycle the events.  The list is emptied whether
// or not every pair could be read.
int harvest_events(rtx_scene* s) {
	hipError_t bad = hipSuccess;
	for (auto& e : s->events) {
		float ms = 0.0f;
		const hipError_t err = e.ended ? hipEventElapsedTime(&ms, e.a, e.b) : hipSuccess;
		if (err != hipSuccess) bad = err;
		else if (e.ended) {
			if (e.kind == 0) {
				s->stats.trace_ms += ms;
				s->stats.trace_ms_per_level[e.level & 15] += ms;
				s->stats.trace_launches++;
			}
			else if (e.kind == 1)
				s->stats.shade_ms += ms;
			else
				s->stats.total_ms += ms;
		}
	}
	drop_events(s);
	if (bad != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("hipEventElapsedTime: ") + hipGetErrorString(bad));
	return RTX_OK;
}

// Slots of a level's counter block (CNT_STRIDE unsigned, zeroed at the start of a pixel batch).
constexpr int WORK_SLOT = 5;     // queue cursor of the wide traversal
constexpr int UNSAFE_SLOT = 6;   // number of rays set aside for the binary-tree kernel
constexpr int WORK2_SLOT = 7;    // queue cursor of the binary-tree kernel
constexpr int LONG_SLOT = 8;     // number of rays handed to the wave-per-ray kernel

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
	RTX_LAUNCHED("k_trace");
	return RTX_OK;
}

template <bool ANYHIT>
int launch_wide_t(rtx_scene* s, const DevScene& d, const float4* ro, const float4* rd, float4* hits, int n, unsigned* work,
				  int* unsafeList, unsigned* unsafeCount, const unsigned* nDev, int* longList, unsigned* longCount) {
	// The stack has 17 rows however deep the tree, and a node's children take the four rows below the top before the
	// top moves: a ray that comes to a node with more than thirteen entries waiting moves to the wave-per-ray kernel
	// (the atrium: 1 ray in 12,000).  With 18 KB of LDS and 72 registers seven blocks are resident per CU, which is
	// worth 2.7 % over the six that 25 rows allow (RTX_WIDE_STACK=24).
	const bool smallStack = s->wideStackCap <= 16;
	const int perCU = smallStack ? s->wideBlocksPerCU : std::min(s->wideBlocksPerCU, 6);
	const int wanted = nDev ? s->numCUs * perCU : (n + TRACE_THREADS - 1) / TRACE_THREADS;
	dim3 grid(std::min(wanted, s->numCUs * perCU)), block(TRACE_THREADS);
	const bool st = s->instrumented;
	unsigned long long* stats = s->dStats.p;
#define RTX_LAUNCH(STK, ST)                                                                                              \
	hipLaunchKernelGGL((k_trace_wide<STK, ANYHIT, ST>), grid, block, 0, s->stream, d, ro, rd, hits, n, stats, work,      \
					   unsafeList, unsafeCount, s->traceRefill, s->traceChunk, s->traceTriBatch, nDev, longList, longCount, \
					   s->longVisits, s->longVisits == INT_MAX ? -1 : s->numCUs * s->longBlocksPerCU * (TRACE_THREADS / 64) * s->smallQueueFactor, \
					   s->longVisits == INT_MAX ? INT_MAX : s->tailVisits, s->longVisits == INT_MAX ? 0 : s->tailLanes, s->retireBacklog)
	if (smallStack) {
		if (st) RTX_LAUNCH(16, true); else RTX_LAUNCH(16, false);
	}
	else {
		if (st) RTX_LAUNCH(24, true); else RTX_LAUNCH(24, false);
	}
#undef RTX_LAUNCH
	RTX_LAUNCHED("k_trace_wide");
	return RTX_OK;
}

// The wave-per-ray kernel over the rays the wide traversal found too long for one lane; the count lives on the
// device, so the grid is a fixed small one.
template <bool ANYHIT>
int launch_long_t(rtx_scene* s, const DevScene& d, const float4* ro, const float4* rd, float4* hits, int n, const int* longList,
				  const unsigned* longCount, int* unsafeList, unsigned* unsafeCount) {
	dim3 grid(std::max(1, s->numCUs * s->longBlocksPerCU)), block(TRACE_THREADS);
	if (s->instrumented)
		hipLaunchKernelGGL((k_trace_long<ANYHIT, true>), grid, block, 0, s->stream, d, ro, rd, hits, longList, longCount, unsafeList,
						   unsafeCount, s->dStats.p, n);
	else
		hipLaunchKernelGGL((k_trace_long<ANYHIT, false>), grid, block, 0, s->stream, d, ro, rd, hits, longList, longCount, unsafeList,
						   unsafeCount, s->dStats.p, n);
	RTX_LAUNCHED("k_trace_long");
	return RTX_OK;
}

// One queue of rays -> hit records.  `cnt` is the level's zeroed counter block, `unsafeList` room for 2n ints (the
// set-aside list in the first n, the list of long rays behind it).
// Production: wide traversal, then the wave-per-ray kernel over the rays that proved long, then the binary-tree
// kernel over the rays either of them set aside.
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
		rc = launch_wide_t<true>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, unsafeList, cnt + UNSAFE_SLOT, nDev, unsafeList + n, cnt + LONG_SLOT);
		if (rc == RTX_OK && s->longVisits != INT_MAX)
			rc = launch_long_t<true>(s, d, ro, rd, hits, n, unsafeList + n, cnt + LONG_SLOT, unsafeList, cnt + UNSAFE_SLOT);
		if (rc == RTX_OK)
			rc = launch_exact_t<true, false>(s, d, ro, rd, hits, n, cnt + WORK2_SLOT, unsafeList, cnt + UNSAFE_SLOT, nullptr);
	}
	else {
		rc = launch_wide_t<false>(s, d, ro, rd, hits, n, cnt + WORK_SLOT, unsafeList, cnt + UNSAFE_SLOT, nDev, unsafeList + n, cnt + LONG_SLOT);
		if (rc == RTX_OK && s->longVisits != INT_MAX)
			rc = launch_long_t<false>(s, d, ro, rd, hits, n, unsafeList + n, cnt + LONG_SLOT, unsafeList, cnt + UNSAFE_SLOT);
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

// Most rays a node can spawn, and the rays per primary sample at level lv if every node did (a double: 65^15
// does not fit an integer).
double branch_max(const rtx_scene* s) { return (double)std::max(s->nSpec + s->nDiff, s->nSpec + 1); }
double worst_ratio(const rtx_scene* s, int lv) { return std::pow(branch_max(s), (double)lv); }

// Queue capacities before anything has been seen: the worst case while that is small (the first frame then cannot
// overflow), a bounded guess where it is not.  A batch that outgrows its queues says so (the device keeps
// counting what was asked for) and is rendered again with larger ones; see grow_caps.
void reset_caps(rtx_scene* s) {
	const double guess = std::max(32.0, branch_max(s));
	for (int lv = 0; lv <= MAX_DEPTH + 1; ++lv) {
		s->popRatio[lv] = 0.0;
		s->capRatio[lv] = s->worstCaseQueues ? worst_ratio(s, lv) : std::min(worst_ratio(s, lv), guess);
	}
	s->capKnown = 0;
	s->haveCaps = true;
}

// Bytes of the work arena per primary sample for the given level capacities: a 24-byte node record for every
// level but the last (the primary-hit export, levels == 0, keeps one), and two ping-pong ray sets plus the two
// ray lists, sized for the widest level.
constexpr size_t NODE_BYTES = 24, RAY_BYTES = 48;
double bytes_per_sample(const double* ratio, int levels) {
	double sum = 0.0, widest = 1.0;
	for (int lv = 0; lv <= levels; ++lv) {
		if (lv < levels || levels == 0) sum += ratio[lv];
		widest = std::max(widest, ratio[lv]);
	}
	return sum * NODE_BYTES + widest * (2.0 * RAY_BYTES + 8.0);
}

struct RenderTarget {
	int mode;      // 0 = accumulate, 1 = export per-sample radiance to dOut, 2 = primary hits to dIds/dTab
	int hitMode;   // mode 2: 0 culled / 1 exhaustive
	float4* dOut;
	int* dIds;
	float* dTab;
};

// Renders the 4x4-pixel blocks [gBegin, gEnd) in as few pixel batches as the arena allows.  Blocks are numbered
// over the frames of the call first (`frame` is the frame of block 0), then over the scene's tile share
// (blocksPerFrame of them), so a batch may span frames.
int render_range(rtx_scene* s, const DevScene& d, uint32_t frame, const RenderTarget& tg, long long gBegin, long long gEnd) {
	const long long blocksPerFrame = (long long)tile_slots(s) * BLOCKS_PER_TILE;
	const int mode = tg.mode;
	const int levels = (mode == 2) ? 0 : s->maxDepth;
	if (!s->haveCaps) reset_caps(s);
	const double fixedBytes = (double)(MAX_DEPTH + 8) * 4096 + (double)(levels + 1) * 1024.0 * NODE_BYTES + 1024.0 * (2.0 * RAY_BYTES + 8.0) + 65536.0;
	// Frames are merged into one batch up to about the samples of one 1080p frame (whole frames only; a share
	// that is larger than that is rendered frame by frame as the arena allows): enough to fill the queues of a
	// small tile share without multiplying the arena of a big one.
	const long long mergeBlocks = std::max(blocksPerFrame, (s->mergeBlocks / blocksPerFrame) * blocksPerFrame);

	// Default budget: 60 % of what is free, at most 64 GiB.  Each batch costs one synchronisation plus, per
	// level, the latency of its longest ray, so bigger batches are better; allocating and mapping hundreds of GB
	// costs seconds, so the arena is not grown beyond that.  The arena never exceeds the budget.
	// The driver is asked for the free memory (tens of microseconds, once per frame otherwise) only when the answer can
	// matter: the first time, and whenever a batch would want more than the arena holds.
	auto default_budget = [&](size_t& out) {
		size_t freeB = 0, totalB = 0;
		HIPCHK(hipMemGetInfo(&freeB, &totalB));
		out = std::min<size_t>((size_t)((double)(freeB + s->arena.n) * 0.6), (size_t)64 << 30);
		return (int)RTX_OK;
	};
	if (!s->arenaLimit && !s->defaultBudget) {
		const int rcb = default_budget(s->defaultBudget);
		if (rcb) return rcb;
	}
	size_t budget = s->arenaLimit ? (size_t)s->arenaLimit : s->defaultBudget;
	auto over_budget = [&](double needBytes) {
		return fail(RTX_ERR_MEMORY, "the ray queues of one 4x4-pixel block need " + std::to_string((unsigned long long)(needBytes / 1048576.0) + 1) +
										" MiB at the ray populations seen so far (max_depth " + std::to_string(s->maxDepth) + ", " +
										std::to_string(s->nSpec) + " specular + " + std::to_string(s->nDiff) + " diffuse rays per hit), beyond the work-arena limit of " +
										std::to_string((unsigned long long)(budget >> 20)) + " MiB (" +
										(s->arenaLimit ? "rtx_scene_set_arena_bytes" : "the default: 60 % of free device memory, at most 64 GiB") + ")");
	};

	int attempts = 0;  // tries of the batch that starts at g0
	int rc = RTX_OK;
	bool budgetFresh = false;  // the default budget has been refreshed during this call
	for (long long g0 = gBegin; g0 < gEnd;) {
		// ---- plan this batch from the capacities as they stand now (an overflow or a first success changes them)
		double ratio[MAX_DEPTH + 2];
		for (int lv = 0; lv <= levels; ++lv) ratio[lv] = lv == 0 ? 1.0 : s->capRatio[lv];
		double perSample = bytes_per_sample(ratio, levels);
		if (!s->arenaLimit && !budgetFresh && perSample * SAMPLES_PER_BLOCK + fixedBytes > (double)budget) {
			const int rcb = default_budget(s->defaultBudget);  // before anything is refused: what is free now
			if (rcb) return rcb;
			budget = s->defaultBudget;
			budgetFresh = true;
			continue;
		}
		if (perSample * SAMPLES_PER_BLOCK + fixedBytes > (double)budget) {
			// Not even one pixel block fits.  Levels beyond capKnown are guesses: they share what the known levels
			// leave of the budget, equally, and the device reports which of them was too small.
			const int known = std::min(s->capKnown, levels);
			double knownNodes = 0.0, knownWidest = 1.0;
			int guessedWithNodes = 0;
			for (int lv = 0; lv <= levels; ++lv) {
				if (lv <= known) {
					if (lv < levels) knownNodes += ratio[lv];
					knownWidest = std::max(knownWidest, ratio[lv]);
				}
				else if (lv < levels) guessedWithNodes++;
			}
			const double avail = ((double)budget - fixedBytes) / SAMPLES_PER_BLOCK - knownNodes * NODE_BYTES;
			const double rayBytes = 2.0 * RAY_BYTES + 8.0;
			if (known >= levels || avail <= knownWidest * rayBytes) return over_budget(perSample * SAMPLES_PER_BLOCK + fixedBytes);
			// guessed levels all get c rays per sample: c is the widest level if it exceeds the known ones
			double c = avail / (guessedWithNodes * (double)NODE_BYTES + rayBytes);
			if (c < knownWidest) c = guessedWithNodes ? (avail - knownWidest * rayBytes) / (guessedWithNodes * (double)NODE_BYTES) : knownWidest;
			c *= 0.999;
			if (!(c > 0.0)) return over_budget(perSample * SAMPLES_PER_BLOCK + fixedBytes);
			for (int lv = known + 1; lv <= levels; ++lv) ratio[lv] = std::min(ratio[lv], c);
			perSample = bytes_per_sample(ratio, levels);
		}
		double maxRatio = 1.0;
		for (int lv = 0; lv <= levels; ++lv) maxRatio = std::max(maxRatio, ratio[lv]);
		const long long totalBlocks = std::min(gEnd - g0, mergeBlocks);
		const double needAll = perSample * (double)totalBlocks * SAMPLES_PER_BLOCK + fixedBytes;
		// The arena's size is kept a multiple of 4 KiB (rounded down: never beyond the budget), because the ray sets
		// are carved from its top end: a size that is not a multiple of 16 would put every float4 of them at a
		// misaligned address, and the scalar loads of the wave-per-ray kernel would read the wrong words.
		if (!s->arenaLimit && needAll + 4096.0 > (double)s->arena.n && !budgetFresh) {
			// the arena would have to grow: what is free now decides how far (the loop plans this batch once more)
			const int rcb = default_budget(s->defaultBudget);
			if (rcb) return rcb;
			budget = s->defaultBudget;
			budgetFresh = true;
			continue;
		}
		size_t want = (size_t)std::min((double)budget, needAll + 4096.0) & ~(size_t)4095;
		// an arena that is at least nearly as big as wanted is used as it is rather than reallocated
		if (s->arena.n && (double)s->arena.n >= 0.9 * (double)want && s->arena.n <= budget) want = std::max(want, s->arena.n);
		long long blocksPerBatch = (long long)(((double)want - fixedBytes) / (perSample * SAMPLES_PER_BLOCK));
		if (blocksPerBatch < 1) return over_budget(perSample * SAMPLES_PER_BLOCK + fixedBytes);
		if (blocksPerBatch > totalBlocks) blocksPerBatch = totalBlocks;
		// keep ray indices inside int32
		const long long maxBlocksInt = (long long)(0x7FFFF000ll / SAMPLES_PER_BLOCK);
		if ((double)blocksPerBatch * maxRatio > (double)maxBlocksInt) {
			blocksPerBatch = (long long)((double)maxBlocksInt / maxRatio);
			if (blocksPerBatch < 1) return fail(RTX_ERR_MEMORY, "one 4x4-pixel block spawns more than 2^31 rays at one level: beyond the queue index range");
		}
		rc = ensure_arena(s, want);
		if (rc) return rc;

		const long long nb = blocksPerBatch;
		Batch b;
		b.g0 = (int)g0;
		b.nSamples = (int)(nb * SAMPLES_PER_BLOCK);
		b.tileFirst = s->tileFirst;
		b.tileStride = s->tileStride;
		b.blocksPerFrame = (int)blocksPerFrame;
		b.frame0 = frame;
		s->stats.chunks++;

		// capacities of the levels of this batch
		size_t cap[MAX_DEPTH + 2];
		size_t widestCap = (size_t)b.nSamples;
		for (int lv = 0; lv <= levels; ++lv) {
			const double worst = (double)b.nSamples * worst_ratio(s, lv);
			const double wanted = std::ceil((double)b.nSamples * ratio[lv]) + 1024.0;
			cap[lv] = lv == 0 ? (size_t)b.nSamples : (size_t)std::min(worst, wanted);
			widestCap = std::max(widestCap, cap[lv]);
		}
		cap[levels + 1] = 0;

		// arena carve-up: transient ray sets at the top, node arrays bump-allocated from the bottom
		const size_t setBytes = ((widestCap * 16 + 255) & ~(size_t)255);
		const size_t listBytes = ((widestCap * 8 + 255) & ~(size_t)255);  // set-aside list + list of long rays
		Bump bump{s->arena.p, s->arena.n & ~(size_t)255, 0, 0};
		if (6 * setBytes + listBytes > bump.cap) return fail(RTX_ERR_MEMORY, "work arena too small for one pixel block");
		bump.hi = bump.cap - 6 * setBytes - listBytes;  // a multiple of 256 like every term
		if (((uintptr_t)bump.base | bump.hi) & 255) return fail(RTX_ERR_DEVICE, "work arena is not 256-byte aligned");
		int* unsafeList = (int*)(bump.base + bump.hi + 6 * setBytes);
		float4* rayO[2] = {(float4*)(bump.base + bump.hi), (float4*)(bump.base + bump.hi + 3 * setBytes)};
		float4* rayD[2] = {(float4*)(bump.base + bump.hi + setBytes), (float4*)(bump.base + bump.hi + 4 * setBytes)};
		float4* hit[2] = {(float4*)(bump.base + bump.hi + 2 * setBytes), (float4*)(bump.base + bump.hi + 5 * setBytes)};

		HIPCHK(hipMemsetAsync(s->dCounts.p, 0, COUNT_WORDS * sizeof(unsigned), s->stream));
		// a batch's first timed group starts at an event of its own: the end of the previous batch lies before that
		// batch's counter copy, the host's planning and a possible reallocation of the arena, none of which is shading
		s->lastEndFresh = false;
		size_t ev;
		rc = begin_event(s, 1, ev);
		if (rc) return rc;
		hipLaunchKernelGGL(k_raygen, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, rayO[0], rayD[0]);
		RTX_LAUNCHED("k_raygen");
		rc = end_event(s, ev);
		if (rc) return rc;

		if (mode == 2) {
			rc = launch_trace(s, d, false, tg.hitMode == 1, rayO[0], rayD[0], hit[0], b.nSamples, s->dCounts.p, unsafeList);
			if (rc) return rc;
			hipLaunchKernelGGL(k_export_hits, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, hit[0], tg.dIds, tg.dTab);
			RTX_LAUNCHED("k_export_hits");
			HIPCHK(hipMemcpyAsync(s->hCounts + FAULT_WORD, s->dCounts.p + FAULT_WORD, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
			HIPCHK(hipStreamSynchronize(s->stream));
			rc = check_fault(s->hCounts[FAULT_WORD]);
			if (rc) return rc;
			g0 += nb;
			continue;
		}

		// Every level is enqueued at once: a level's queue length stays on the device (slot 0 of its counter
		// block, written by the shade kernel of the level above), arrays are carved at the level's capacity
		// and grids are fixed.  The host reads the counters once, after the batch.
		float4* A[MAX_DEPTH + 1];
		int2* B[MAX_DEPTH + 1];
		for (int lv = 0; lv <= s->maxDepth; ++lv) {
			const size_t n = cap[lv];
			const bool keepsNodes = lv < s->maxDepth;
			A[lv] = keepsNodes ? (float4*)bump.take(n * 16) : nullptr;
			B[lv] = keepsNodes ? (int2*)bump.take(n * 8) : nullptr;
			if (keepsNodes && (!A[lv] || !B[lv])) return fail(RTX_ERR_MEMORY, "work arena exhausted (internal sizing error)");
			const int cur = lv & 1, nxt = cur ^ 1;
			const unsigned* nDev = lv ? s->dCounts.p + (size_t)lv * CNT_STRIDE : nullptr;
			rc = launch_trace(s, d, lv == s->maxDepth, false, rayO[cur], rayD[cur], hit[cur], (int)n,
							  s->dCounts.p + (size_t)lv * CNT_STRIDE, unsafeList, lv, nDev);
			if (rc) return rc;
			// the last level (max_recursion_depth >= 1) has no shade pass: the gather of the level above forms
			// its values from the any-hit answers
			if (lv == s->maxDepth) break;
			unsigned* cnt = s->dCounts.p + (size_t)(lv + 1) * CNT_STRIDE;
			rc = begin_event(s, 1, ev);
			if (rc) return rc;
			hipLaunchKernelGGL(k_shade, dim3((unsigned)std::min<size_t>((n + SHADE_THREADS - 1) / SHADE_THREADS, (size_t)s->numCUs * (size_t)s->shadeBlocksPerCU)), dim3(SHADE_THREADS), 0, s->stream, d, lv,
							   rayO[cur], rayD[cur], hit[cur], (int)n, A[lv], B[lv], rayO[nxt], rayD[nxt], cnt, (unsigned)cap[lv + 1],
							   nDev);
			RTX_LAUNCHED("k_shade");
			rc = end_event(s, ev);
			if (rc) return rc;
		}
		rc = begin_event(s, 1, ev);
		if (rc) return rc;
		for (int lv = s->maxDepth - 1; lv >= 0; --lv) {
			const unsigned* nDev = lv ? s->dCounts.p + (size_t)lv * CNT_STRIDE : nullptr;
			const dim3 grid((unsigned)std::min<size_t>((cap[lv] + 255) / 256, (size_t)s->numCUs * 8));  // one resident set of 256-thread blocks
			if (lv == s->maxDepth - 1) {
				// the children are the last level: its rays and any-hit answers lie in the buffers of its parity
				const int last = s->maxDepth & 1;
				hipLaunchKernelGGL(k_gather<true>, grid, dim3(256), 0, s->stream, d, (int)cap[lv], A[lv], B[lv], (const float4*)nullptr,
								   (const int*)hit[last], (const float4*)rayD[last], (unsigned)cap[lv + 1], s->dCounts.p + (size_t)(s->maxDepth + 1) * CNT_STRIDE, nDev);
			}
			else
				hipLaunchKernelGGL(k_gather<false>, grid, dim3(256), 0, s->stream, d, (int)cap[lv], A[lv], B[lv], (const float4*)A[lv + 1],
								   (const int*)nullptr, (const float4*)nullptr, (unsigned)cap[lv + 1], (unsigned*)nullptr, nDev);
			RTX_LAUNCHED("k_gather");
		}
		if (mode == 0) {
			// one launch per frame the batch touches, in frame order
			for (long long f = g0 / blocksPerFrame; f * blocksPerFrame < g0 + nb; ++f) {
				const long long segBegin = std::max(g0, f * blocksPerFrame), segEnd = std::min(g0 + nb, (f + 1) * blocksPerFrame);
				Batch seg = b;
				seg.g0 = (int)segBegin;
				seg.nSamples = (int)((segEnd - segBegin) * SAMPLES_PER_BLOCK);
				const int npix = seg.nSamples / 4;
				hipLaunchKernelGGL(k_accumulate, dim3((npix + 255) / 256), dim3(256), 0, s->stream, d, seg,
								   (const float4*)(A[0] + (size_t)(segBegin - g0) * SAMPLES_PER_BLOCK), s->dAccum, s->dCounts.p);
				// The frame's last blocks: the accumulator now is the image after that frame, which the caller may have
				// asked for (the frames behind it in this batch are added only after the copy, in stream order).  A batch
				// that overflowed a queue stores nothing and is rendered again: its images are then written a second time.
				const long long frameNo = (long long)frame + f;
				if (s->toneOut && segEnd == (f + 1) * blocksPerFrame && frameNo >= s->toneOutFirst && frameNo < (long long)s->toneOutFirst + s->toneOutCount) {
					const size_t npixAll = (size_t)s->width * s->height;
					hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((npixAll + 1023) / 1024)), dim3(256), 0, s->stream, s->dAccum, (int)npixAll, (float)frameNo,
									   s->dToneBits.p, s->dToneVal.p, s->toneN, s->toneNeg, s->toneNan, s->dRgb8.p);
					RTX_LAUNCHED("k_tonemap");
					HIPCHK(hipMemcpyAsync(s->toneOut[frameNo - s->toneOutFirst], s->dRgb8.p, npixAll * 3, hipMemcpyDeviceToHost, s->stream));
				}
			}
		}
		else
			hipLaunchKernelGGL(k_export_samples, dim3((b.nSamples + 255) / 256), dim3(256), 0, s->stream, d, b, A[0], tg.dOut, s->dCounts.p);
		RTX_LAUNCHED("k_accumulate / k_export_samples");
		rc = end_event(s, ev);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(s->hCounts, s->dCounts.p, COUNT_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
		HIPCHK(hipStreamSynchronize(s->stream));

		bool overflow = false;
		for (int lv = 0; lv <= s->maxDepth; ++lv) overflow = overflow || s->hCounts[(size_t)(lv + 1) * CNT_STRIDE + 4] != 0;
		// the kernel times of an attempt that is rendered again belong to no frame
		if (overflow) drop_events(s);
		else rc = harvest_events(s);
		if (rc) return rc;
		// what the levels asked for, whether or not it fitted (the cursor runs on past the capacity)
		double asked[MAX_DEPTH + 2];
		asked[0] = 1.0;
		for (int lv = 1; lv <= s->maxDepth; ++lv) asked[lv] = (double)s->hCounts[(size_t)lv * CNT_STRIDE] / (double)b.nSamples;
		if (overflow) {
			// Nothing of this batch was stored.  The first level that asked for more than it had was fed by complete
			// levels, so its count is the true one; the levels behind it saw only part of their parents.  They grow
			// by the factor that level grew over its parent, level by level, and the batch is rendered again (in
			// smaller batches if the arena no longer holds it at once).
			int first = 1;
			while (first < s->maxDepth && s->hCounts[(size_t)first * CNT_STRIDE] <= cap[first]) ++first;
			if (s->worstCaseQueues || ++attempts > 40)
				return fail(RTX_ERR_MEMORY, "child queue overflow at level " + std::to_string(first) + " (internal sizing error)");
			const double growth = std::max(1.0, asked[first] / std::max(asked[first - 1], 1.0 / 64.0));
			for (int lv = 1; lv <= s->maxDepth; ++lv) {
				double want = asked[lv] * 1.3 + 1.0 / 64.0;
				if (lv <= first) {
					// a true count: it replaces the guess the level started with, so that a later plan adds up what the
					// levels hold and not what was guessed for them
					s->popRatio[lv] = std::max(s->popRatio[lv], asked[lv]);
					s->capRatio[lv] = std::min(worst_ratio(s, lv), lv > s->capKnown ? std::max(want, s->popRatio[lv] * 1.3 + 1.0 / 64.0) : std::max(s->capRatio[lv], want));
				}
				else {
					want = std::max(want, s->capRatio[lv - 1] * growth);
					s->capRatio[lv] = std::min(worst_ratio(s, lv), std::max(s->capRatio[lv], want));
				}
			}
			s->capKnown = std::max(s->capKnown, first);
			s->stats.chunks--;
			s->stats.queue_retries++;
			continue;  // the same blocks once more, planned anew
		}
		for (int lv = 1; lv <= s->maxDepth; ++lv) {
			s->popRatio[lv] = std::max(s->popRatio[lv], asked[lv]);
			// 1.3 times the most seen so far plus a little, capped by the worst case
			if (!s->worstCaseQueues) s->capRatio[lv] = std::min(worst_ratio(s, lv), s->popRatio[lv] * 1.3 + 1.0 / 64.0);
		}
		s->capKnown = s->maxDepth;
		rc = check_fault(s->hCounts[FAULT_WORD]);
		if (rc) return rc;
		uint64_t n = (uint64_t)b.nSamples;
		for (int lv = 0; lv <= s->maxDepth; ++lv) {
			const unsigned* hc = s->hCounts + (size_t)(lv + 1) * CNT_STRIDE;
			s->stats.rays_per_level[lv] += n - hc[1];
			s->stats.rays += n - hc[1];
			if (lv == s->maxDepth) s->stats.rays_untraced += hc[2];  // the last level shades nothing
			else s->stats.shaded_hits += hc[2];
			s->stats.texture_fetches += hc[3];
			s->stats.rays_set_aside += s->hCounts[(size_t)lv * CNT_STRIDE + UNSAFE_SLOT];
			s->stats.rays_wave_traced += s->hCounts[(size_t)lv * CNT_STRIDE + LONG_SLOT];
			n = hc[0];
		}
		g0 += nb;
		attempts = 0;
	}
	HIPCHK(hipStreamSynchronize(s->stream));
	return harvest_events(s);
}

// Frames frame .. frame + nFrames - 1 of the scene's tile share (nFrames > 1: accumulate mode only).
int render_batches(rtx_scene* s, uint32_t frame, uint32_t nFrames, int mode, int hitMode, float4* dOut, int* dIds, float* dTab) {
	const DevScene d = make_dev_scene(s);
	const long long blocksPerFrame = (long long)tile_slots(s) * BLOCKS_PER_TILE;
	if (blocksPerFrame == 0) return RTX_OK;
	const RenderTarget tg{mode, hitMode, dOut, dIds, dTab};
	// block numbers stay inside int32: more frames than that go in several rounds
	const uint32_t perRound = (uint32_t)std::max<long long>(1, std::min<long long>(nFrames, 0x7FFF0000ll / SAMPLES_PER_BLOCK / blocksPerFrame));
	for (uint32_t f = 0; f < nFrames; f += perRound) {
		const uint32_t nf = std::min(perRound, nFrames - f);
		int rc = render_range(s, d, frame + f, tg, 0, blocksPerFrame * nf);
		if (rc) {
			// an error may have left kernels of the batch enqueued: nothing may be freed or reused under them
			(void)hipStreamSynchronize(s->stream);
			return rc;
		}
	}
	return RTX_OK;
}

int finish_stats(rtx_scene* s) {
	if (s->instrumented) {
		HIPCHK(hipMemcpy(s->hStats, s->dStats.p, STATS_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		s->stats.node_tests = s->hStats[0];
		s->stats.tri_tests = s->hStats[1];
		if (std::getenv("RTX_TRACE_DEBUG")) {
			std::fprintf(stderr, "[rtx] wide: walk iterations %llu with %.1f lanes, leaf phases %llu with %.1f lanes, %llu stack overflows\n", s->hStats[2],
						 (double)s->hStats[3] / (double)std::max(1ull, s->hStats[2]), s->hStats[6],
						 (double)s->hStats[7] / (double)std::max(1ull, s->hStats[6]), s->hStats[8]);
			const double wi = (double)std::max(1ull, s->hStats[2]), li = (double)std::max(1ull, s->hStats[6]);
			std::fprintf(stderr, "[rtx] wide: per walk iteration %.1f lanes without a ray, %.1f holding two leaves, %.1f with leaves only; per leaf phase %.1f without a ray, %.1f walkers waiting; %llu refills of %.1f lanes\n",
						 (double)s->hStats[12] / wi, (double)s->hStats[13] / wi, (double)s->hStats[14] / wi, (double)s->hStats[15] / li, (double)s->hStats[16] / li,
						 s->hStats[17], (double)s->hStats[18] / (double)std::max(1ull, s->hStats[17]));
			std::fprintf(stderr, "[rtx] wide: second triangles left without a helper: a second pass in %llu leaf phases, for %.1f lanes (%.1f helped)\n", s->hStats[19],
						 (double)s->hStats[20] / (double)std::max(1ull, s->hStats[19]), (double)s->hStats[21] / (double)std::max(1ull, s->hStats[19]));
			std::fprintf(stderr, "[rtx] wide: most node visits of one ray %llu, rays over 100: %llu, over 300: %llu\n", s->hStats[9], s->hStats[10], s->hStats[11]);
		}
		s->stats.wide_visits = s->hStats[4];
		s->stats.leaf_fetches = s->hStats[5];
	}
	return RTX_OK;
}

int begin_call(rtx_scene* s) {
	std::memset(&s->stats, 0, sizeof(s->stats));
	drop_events(s);  // pairs an earlier call left behind when it failed half-way
	if (s->instrumented) HIPCHK(hipMemsetAsync(s->dStats.p, 0, STATS_WORDS * sizeof(unsigned long long), s->stream));
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
	{
		int rc = scratch_for_device(device, &s->noallocScratch);
		// the one host word rtx_trace_rays_noalloc reads back: pinned, so that its copy is a plain DMA into memory that
		// exists already (a copy into pageable memory is staged by the runtime, through memory the runtime obtains)
		if (rc == RTX_OK && hipHostMalloc((void**)&s->hNoallocFault, 64, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			s->hNoallocFault = nullptr;
			rc = fail(RTX_ERR_MEMORY, "cannot allocate the scene's pinned status word");
		}
		if (rc) {
			delete s;
			return rc;
		}
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->numCUs = prop.multiProcessorCount;
	if (const char* e = std::getenv("RTX_TRACE_REFILL")) s->traceRefill = std::max(1, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_TRACE_CHUNK")) s->traceChunk = std::max(64, std::atoi(e));
	if (const char* e = std::getenv("RTX_LONG_BLOCKS")) s->longBlocksPerCU = std::max(1, std::min(8, std::atoi(e)));
	if (const char* e = std::getenv("RTX_SMALL_QUEUE")) s->smallQueueFactor = std::max(0, std::atoi(e));
	if (const char* e = std::getenv("RTX_LONG_VISITS")) s->longVisits = std::atoi(e) > 0 ? std::atoi(e) : INT_MAX;
	if (const char* e = std::getenv("RTX_TAIL_VISITS")) s->tailVisits = std::max(0, std::atoi(e));
	if (const char* e = std::getenv("RTX_SHADE_BLOCKS_PER_CU")) s->shadeBlocksPerCU = std::max(1, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_RETIRE_BACKLOG")) s->retireBacklog = std::max(0, std::atoi(e));
	if (const char* e = std::getenv("RTX_TAIL_LANES")) s->tailLanes = std::max(0, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_MERGE_BLOCKS")) s->mergeBlocks = std::max(1ll, std::atoll(e));
	if (const char* e = std::getenv("RTX_TRACE_TRI_BATCH")) s->traceTriBatch = std::max(1, std::min(64, std::atoi(e)));
	if (const char* e = std::getenv("RTX_TRACE_BLOCKS_PER_CU")) s->wideBlocksPerCU = s->traceBlocksPerCU = std::max(1, std::atoi(e));
	if (const char* e = std::getenv("RTX_TRACE_BINARY")) s->forceBinary = std::atoi(e) != 0;
	if (const char* e = std::getenv("RTX_SYNC_LAUNCHES")) s->syncLaunches = std::atoi(e) != 0;
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
	s->haveCaps = false;
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
	if (n > (1u << 27)) return fail(RTX_ERR_ARG, "too many triangles");  // a leaf record keeps its triangle's rank in 28 bits
	s->tris.resize(n);
	if (n) std::memcpy((void*)s->tris.data(), triangles, (size_t)n * sizeof(HostTri));
	float amax = 0.0f;
	int maxTex = -1;
	for (const HostTri& t : s->tris) {
		maxTex = std::max(maxTex, (int)t.tex);
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
	s->maxTriTex = maxTex;
	s->haveTris = true;
	s->haveBvh = false;
	s->haveCaps = false;
	return RTX_OK;
}

RTX_API int rtx_scene_set_bvh(rtx_scene* s, const void* linear_nodes, uint32_t n) {
	if (!s || !linear_nodes) return fail(RTX_ERR_ARG, "null argument");
	const bool timing = std::getenv("RTX_TRACE_DEBUG") != nullptr;
	auto tPrev = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[rtx] set_bvh: %-28s %.3f s\n", what, std::chrono::duration<double>(now - tPrev).count());
		tPrev = now;
	};
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

	lap("ranks (reference DFS)");
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

	lap("pair records");
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
	lap("premise check");
	std::vector<float> wide;
	std::vector<int> recTri;  // triangle of every leaf record of the wide tree, in the order the records are stored
	int wideStackNeed = 1;
	size_t pairedLeaves = 0;
	const char* treeKind = "none";
	if (wideOk) {
		// own surface-area tree over the triangles' boxes; the reference's tree if that one cannot be used
		bool built = false;
		const char* want = std::getenv("RTX_WIDE_TREE");
		if (!(want && std::strcmp(want, "reference") == 0)) {
			std::vector<float> box((size_t)ntri * 6), verts((size_t)ntri * 9);
			for (uint32_t i = 0; i < ntri; ++i) {
				const HostTri& t = s->tris[i];
				for (int c = 0; c < 3; ++c) {
					box[(size_t)i * 6 + c] = std::min(std::min(t.v[0][c], t.v[1][c]), t.v[2][c]);
					box[(size_t)i * 6 + 3 + c] = std::max(std::max(t.v[0][c], t.v[1][c]), t.v[2][c]);
					for (int k = 0; k < 3; ++k) verts[(size_t)i * 9 + k * 3 + c] = t.v[k][c];
				}
			}
			const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
			lap("triangle boxes");
			// leaves: pairs of triangles that share an edge and a box (RTX_LEAF_PAIRS=0: one triangle each)
			std::vector<int> mate(ntri, -1);
			const char* lp = std::getenv("RTX_LEAF_PAIRS");
			if (!(lp && std::atoi(lp) == 0)) rtxwide::pair_triangles(verts.data(), box, (int)ntri, mate, threads);
			lap("leaf pairs");
			std::vector<float> primBox;
			std::vector<int> leafValue;
			primBox.reserve((size_t)ntri * 6);
			recTri.reserve(ntri);
			for (uint32_t i = 0; i < ntri; ++i) {
				const int m = mate[i];
				if (m >= 0 && (uint32_t)m < i) continue;  // emitted with its mate
				const size_t rec = recTri.size();
				float b6[6];
				std::memcpy(b6, &box[(size_t)i * 6], sizeof(b6));
				if (m >= 0) {
					for (int c = 0; c < 3; ++c) {
						b6[c] = std::min(b6[c], box[(size_t)m * 6 + c]);
						b6[3 + c] = std::max(b6[3 + c], box[(size_t)m * 6 + 3 + c]);
					}
					// the triangle the reference would come to first stands first
					const bool iFirst = rankOf[i] < rankOf[m];
					recTri.push_back(iFirst ? (int)i : m);
					recTri.push_back(iFirst ? m : (int)i);
					++pairedLeaves;
				}
				else
					recTri.push_back((int)i);
				primBox.insert(primBox.end(), b6, b6 + 6);
				leafValue.push_back((int)(rec * 2 + (m >= 0 ? 1 : 0)));
			}
			std::vector<rtxwide::BinNode> own;
			rtxwide::build_sah(primBox, (int)leafValue.size(), own, threads);
			lap("surface-area tree");
			built = rtxwide::collapse_wide(own, leafValue, wide, wideStackNeed, threads);
			lap("4-wide collapse + quantise");
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
			// one triangle per leaf, records in rank order
			std::vector<int> leafValue(ntri);
			recTri.assign(ntri, 0);
			pairedLeaves = 0;
			for (uint32_t i = 0; i < ntri; ++i) {
				leafValue[i] = rankOf[i] * 2;
				recTri[(size_t)rankOf[i]] = (int)i;
			}
			built = rtxwide::collapse_wide(ref, leafValue, wide, wideStackNeed);
			treeKind = "reference";
		}
		wideOk = built;
	}

	if (std::getenv("RTX_TRACE_DEBUG"))
		std::fprintf(stderr, "[rtx] wide tree (%s): ok %d, %zu nodes, %zu of %u triangles in two-triangle leaves, stack need %d (binary: %zu pairs, stack need %d)\n", treeKind,
					 (int)wideOk, wide.size() / 16, pairedLeaves * 2, ntri, wideStackNeed, npairs, maxInteriorDepth + 1);

	lap("-");
	// Triangle records in rank order.
	// Shading records: normals, texture coordinates and the number of the triangle's material.  What the triangles of
	// a model share (src/RayTracer.cpp:76-103 copies it into every Triangle) -- colour, roughness, refractive index,
	// texture, the three flags -- is kept once per distinct combination, compared bit for bit.
	std::vector<float4> th((size_t)std::max<uint32_t>(ntri, 1) * 4), ts((size_t)std::max<uint32_t>(ntri, 1) * 4);
	std::vector<float4> mats;
	struct MatKey {
		uint32_t w[7];
		bool operator==(const MatKey& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
	};
	struct MatHash {
		size_t operator()(const MatKey& k) const {
			uint64_t h = 0xCBF29CE484222325ull;
			for (uint32_t v : k.w) h = (h ^ v) * 0x100000001B3ull;
			return (size_t)h;
		}
	};
	std::unordered_map<MatKey, int, MatHash> matIndex;
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
		MatKey mk;
		std::memcpy(&mk.w[0], t.color, 12);
		std::memcpy(&mk.w[3], &t.rough, 4);
		std::memcpy(&mk.w[4], &t.ior, 4);
		mk.w[5] = (uint32_t)t.tex;
		mk.w[6] = (uint32_t)flags;
		auto found = matIndex.find(mk);
		if (found == matIndex.end()) {
			found = matIndex.emplace(mk, (int)(mats.size() / 2)).first;
			mats.push_back(make_float4(t.color[0], t.color[1], t.color[2], t.rough));
			mats.push_back(make_float4(t.ior, texf, flagf, 0.0f));
		}
		float matf;
		const int32_t matId = found->second;
		std::memcpy(&matf, &matId, 4);
		ts[r * 4 + 0] = make_float4(t.n[0][0], t.n[0][1], t.n[0][2], t.uv[0][0]);
		ts[r * 4 + 1] = make_float4(t.n[1][0], t.n[1][1], t.n[1][2], t.uv[0][1]);
		ts[r * 4 + 2] = make_float4(t.n[2][0], t.n[2][1], t.n[2][2], t.uv[1][0]);
		ts[r * 4 + 3] = make_float4(t.uv[1][1], t.uv[2][0], t.uv[2][1], matf);
	}
	if (mats.empty()) {  // a scene without triangles still has a table to point at
		float noTex;
		const int32_t minusOne = -1;
		std::memcpy(&noTex, &minusOne, 4);
		mats.push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
		mats.push_back(make_float4(1.0f, noTex, 0.0f, 0.0f));
	}

	// The same hit records once more in the order of the wide tree's leaves (the two triangles of a leaf side by
	// side), each with its rank in the fourth quad: what k_trace_wide and k_trace_long read.
	std::vector<float4> tw;
	if (wideOk) {
		tw.resize((size_t)std::max<uint32_t>(ntri, 1) * 4);
		for (size_t rec = 0; rec < recTri.size(); ++rec) {
			const size_t r = (size_t)rankOf[recTri[rec]];
			tw[rec * 4 + 0] = th[r * 4 + 0];
			tw[rec * 4 + 1] = th[r * 4 + 1];
			tw[rec * 4 + 2] = th[r * 4 + 2];
			// The part of Triangle::hit's elimination (src/Triangle.cpp:41-79) that does not depend on the ray, with the
			// reference's operations in the reference's order: rows (a1_i, a2_i), first pivot = largest |a1_i| (strict >,
			// rows tried in order), its row divided by the pivot, the other two rows reduced by one fused multiply-add,
			// second pivot = the larger |column 1| of those (strict >).  See tri_test_pre.
			const float4 Q0 = th[r * 4 + 0], Q1 = th[r * 4 + 1], Q2 = th[r * 4 + 2];
			const float v0[3] = {Q0.w, Q1.x, Q1.y}, v1[3] = {Q1.z, Q1.w, Q2.x}, v2[3] = {Q2.y, Q2.z, Q2.w};
			float c0[3], c1[3];
			for (int c = 0; c < 3; ++c) {
				c0[c] = v0[c] - v2[c];
				c1[c] = v1[c] - v2[c];
			}
			const bool s1 = std::fabs(c0[1]) > std::fabs(c0[0]);
			const float mx = s1 ? std::fabs(c0[1]) : std::fabs(c0[0]);
			const bool s2 = std::fabs(c0[2]) > mx;
			const int rowP = s2 ? 2 : (s1 ? 1 : 0);
			const int rowA = (s1 && !s2) ? 0 : 1, rowC = s2 ? 0 : 2;
			const float p1q = c1[rowP] / c0[rowP];
			const float a1 = std::fmaf(-p1q, c0[rowA], c1[rowA]);
			const float d1 = std::fmaf(-p1q, c0[rowC], c1[rowC]);
			const bool s3 = std::fabs(d1) > std::fabs(a1);
			const float e1 = s3 ? d1 : a1;
			const int rowE = s3 ? rowC : rowA;
			float rankf;
			const int32_t ri = (int32_t)(((uint32_t)r << 4) | (uint32_t)(rowP | (rowE << 2)));
			std::memcpy(&rankf, &ri, 4);
			tw[rec * 4 + 3] = make_float4(th[r * 4 + 3].x, rankf, p1q, e1);
		}
	}
	lap("triangle records");
	HIPCHK(hipSetDevice(s->device));
	HIPCHK(s->dPairs.alloc(pairs.size()));
	HIPCHK(hipMemcpy(s->dPairs.p, pairs.data(), pairs.size() * sizeof(float4), hipMemcpyHostToDevice));
	if (wideOk) {
		HIPCHK(s->dWide.alloc(wide.size() / 4));
		HIPCHK(hipMemcpy(s->dWide.p, wide.data(), wide.size() * sizeof(float), hipMemcpyHostToDevice));
		HIPCHK(s->dTriWide.alloc(tw.size()));
		HIPCHK(hipMemcpy(s->dTriWide.p, tw.data(), tw.size() * sizeof(float4), hipMemcpyHostToDevice));
	}
	s->wideOk = wideOk;
	s->wideStackNeed = wideStackNeed;
	s->numWide = (uint32_t)(wide.size() / 16);
	HIPCHK(s->dTriHit.alloc(th.size()));
	HIPCHK(hipMemcpy(s->dTriHit.p, th.data(), th.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(s->dTriShade.alloc(ts.size()));
	HIPCHK(hipMemcpy(s->dTriShade.p, ts.data(), ts.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(s->dMats.alloc(mats.size()));
	HIPCHK(hipMemcpy(s->dMats.p, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
	lap("copies to the device");
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
	s->defaultBudget = 0;
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
	std::fill(s->tileLog.begin(), s->tileLog.end(), 0);
	return RTX_OK;
}

namespace {

int render_frames_call(rtx_scene* s, uint32_t first_frame, uint32_t n_frames) {
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	// the pair around the whole call is kept apart from the list the per-batch harvests empty
	hipEvent_t totalA, totalB;
	rc = take_event(s, totalA);
	if (rc) return rc;
	rc = take_event(s, totalB);
	if (rc) {
		s->eventPool.push_back(totalA);
		return rc;
	}
	hipError_t e = hipEventRecord(totalA, s->stream);
	if (e == hipSuccess) {
		rc = render_batches(s, first_frame, n_frames, 0, 0, nullptr, nullptr, nullptr);
		if (rc == RTX_OK) e = hipEventRecord(totalB, s->stream);
		if (rc == RTX_OK && e == hipSuccess) e = hipStreamSynchronize(s->stream);
		float ms = 0.0f;
		if (rc == RTX_OK && e == hipSuccess) e = hipEventElapsedTime(&ms, totalA, totalB);
		s->stats.total_ms = ms;
	}
	s->eventPool.push_back(totalA);
	s->eventPool.push_back(totalB);
	if (rc) {
		drop_events(s);
		return rc;
	}
	if (e != hipSuccess) return fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	rc = harvest_events(s);
	if (rc) return rc;
	// the tiles of the share now hold what this call rendered: rtx_exchange_tiles moves exactly those
	{
		const int total = num_tiles(s);
		if ((int)s->tileLog.size() != total) s->tileLog.assign((size_t)total, 0);
		for (int k = 0, n = tile_slots(s); k < n; ++k) s->tileLog[(size_t)(s->tileFirst + k * s->tileStride)] = 1;
	}
	return finish_stats(s);
}

}  // namespace

RTX_API int rtx_render_frames(rtx_scene* s, uint32_t first_frame, uint32_t n_frames) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	return render_frames_call(s, first_frame, n_frames);
}

RTX_API int rtx_render_frames_tonemapped(rtx_scene* s, uint32_t first_frame, uint32_t n_frames, uint8_t* const* host_rgb8) {
	if (!s || !host_rgb8) return fail(RTX_ERR_ARG, "null argument");
	if (first_frame < 1) return fail(RTX_ERR_ARG, "frame must be >= 1");
	for (uint32_t k = 0; k < n_frames; ++k)
		if (!host_rgb8[k]) return fail(RTX_ERR_ARG, "null image pointer");
	if (!s->toneN) return fail(RTX_ERR_STATE, "rtx_scene_set_tonemap_table has not been called");
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	const size_t npix = (size_t)s->width * s->height;
	if (!s->dRgb8.p || s->dRgb8.n < npix * 3 + 16) HIPCHK(s->dRgb8.alloc(npix * 3 + 16));
	s->toneOut = host_rgb8;
	s->toneOutFirst = first_frame;
	s->toneOutCount = n_frames;
	rc = render_frames_call(s, first_frame, n_frames);
	s->toneOut = nullptr;
	s->toneOutCount = 0;
	return rc;
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
	// the export arrays stay with the scene: a call allocates only when the image has grown
	if (s->dExportIds.n < ns) HIPCHK(s->dExportIds.alloc(ns));
	if (t_alpha_beta && s->dExportTab.n < ns * 3) HIPCHK(s->dExportTab.alloc(ns * 3));
	HIPCHK(hipMemsetAsync(s->dExportIds.p, 0xFF, ns * sizeof(int), s->stream));
	if (t_alpha_beta) HIPCHK(hipMemsetAsync(s->dExportTab.p, 0, ns * 3 * sizeof(float), s->stream));
	rc = render_batches(s, 0, 1, 2, mode, nullptr, s->dExportIds.p, t_alpha_beta ? s->dExportTab.p : nullptr);
	if (rc) return rc;
	// both arrays travel behind the kernels on the scene's stream, one wait for the two; into page-locked memory
	// (rtx_host_alloc) they go at the speed of the link, into pageable memory the runtime stages them
	HIPCHK(hipMemcpyAsync(tri_ids, s->dExportIds.p, ns * sizeof(int), hipMemcpyDeviceToHost, s->stream));
	if (t_alpha_beta) HIPCHK(hipMemcpyAsync(t_alpha_beta, s->dExportTab.p, ns * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
	HIPCHK(hipStreamSynchronize(s->stream));
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

RTX_API void* rtx_host_alloc(size_t bytes) {
	void* p = nullptr;
	if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		fail(RTX_ERR_MEMORY, "cannot allocate " + std::to_string(bytes) + " bytes of page-locked host memory");
		return nullptr;
	}
	return p;
}

RTX_API int rtx_host_free(void* p) {
	if (p) HIPCHK(hipHostFree(p));
	return RTX_OK;
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
	// device arrays stay with the scene and grow on demand; ids and t/alpha/beta share the export arrays of rtx_primary_hits
	hipError_t e = hipSuccess;
	if (s->dRaysO.n < n) e = s->dRaysO.alloc(n);
	if (e == hipSuccess && s->dRaysD.n < n) e = s->dRaysD.alloc(n);
	if (e == hipSuccess && s->dRaysH.n < n) e = s->dRaysH.alloc(n);
	if (e == hipSuccess && s->dRaysList.n < (size_t)n * 2) e = s->dRaysList.alloc((size_t)n * 2);
	if (e == hipSuccess && s->dExportIds.n < n) e = s->dExportIds.alloc(n);
	if (e == hipSuccess && s->dExportTab.n < (size_t)n * 3) e = s->dExportTab.alloc((size_t)n * 3);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return fail(RTX_ERR_MEMORY, std::string("rtx_trace_rays: ") + hipGetErrorString(e));
	}
	e = hipMemcpy(s->dRaysO.p, ho.data(), (size_t)n * 16, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(s->dRaysD.p, hd.data(), (size_t)n * 16, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		const DevScene d = make_dev_scene(s);
		e = hipMemsetAsync(s->dCounts.p, 0, CNT_STRIDE * sizeof(unsigned), s->stream);
		if (e == hipSuccess) e = hipMemsetAsync(s->dCounts.p + FAULT_WORD, 0, sizeof(unsigned), s->stream);
		if (e == hipSuccess) rc = launch_trace(s, d, false, mode == 1, s->dRaysO.p, s->dRaysD.p, s->dRaysH.p, (int)n, s->dCounts.p, s->dRaysList.p);
		if (e == hipSuccess && rc == RTX_OK) {
			hipLaunchKernelGGL(k_export_hits_linear, dim3((n + 255) / 256), dim3(256), 0, s->stream, d, (int)n, s->dRaysH.p, s->dExportIds.p, s->dExportTab.p);
			e = hipGetLastError();
			if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
			if (e == hipSuccess) e = hipMemcpy(s->hCounts + FAULT_WORD, s->dCounts.p + FAULT_WORD, sizeof(unsigned), hipMemcpyDeviceToHost);
			if (e == hipSuccess) e = hipMemcpy(s->hCounts + UNSAFE_SLOT, s->dCounts.p + UNSAFE_SLOT, sizeof(unsigned), hipMemcpyDeviceToHost);
			if (e == hipSuccess) rc = check_fault(s->hCounts[FAULT_WORD]);
			if (e == hipSuccess) e = hipMemcpy(tri_ids, s->dExportIds.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
			if (e == hipSuccess && t_alpha_beta) e = hipMemcpy(t_alpha_beta, s->dExportTab.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost);
		}
	}
	if (e != hipSuccess || rc) (void)hipStreamSynchronize(s->stream);  // nothing of a failed call stays in flight over the scene's arrays
	if (e != hipSuccess) return fail(RTX_ERR_DEVICE, hipGetErrorString(e));
	if (rc) return rc;
	rc = harvest_events(s);
	if (rc) return rc;
	s->stats.rays = n;
	s->stats.rays_per_level[0] = n;
	s->stats.rays_set_aside = s->hCounts[UNSAFE_SLOT];
	return finish_stats(s);
}

RTX_API int rtx_trace_rays_noalloc(rtx_scene* s, int mode, uint32_t n, const float* d_origins, const float* d_directions,
								   int32_t* d_tri_ids, float* d_t_alpha_beta) {
	if (!s || !d_origins || !d_directions || !d_tri_ids) return fail(RTX_ERR_ARG, "null argument");
	if (mode != 0 && mode != 1) return fail(RTX_ERR_ARG, "mode must be 0 or 1");
	if (n > RTX_NOALLOC_MAX_RAYS)
		return fail(RTX_ERR_ARG, "rtx_trace_rays_noalloc takes at most " + std::to_string(RTX_NOALLOC_MAX_RAYS) + " rays per call (" +
									 std::to_string(n) + " given): its work memory is a fixed array");
	if (!s->haveTris || !s->haveBvh) return fail(RTX_ERR_STATE, "rtx_scene_set_triangles and rtx_scene_set_bvh must come first");
	if (!s->noallocScratch || !s->hNoallocFault) return fail(RTX_ERR_STATE, "the scene has no scratch array");
	HIPCHK(hipSetDevice(s->device));
	std::memset(&s->stats, 0, sizeof(s->stats));
	if (n == 0) return RTX_OK;
	// carve the array: counters | origins | directions | hit records | set-aside list and list of long rays
	unsigned char* base = s->noallocScratch;
	unsigned* counts = (unsigned*)base;
	float4* rayO = (float4*)(base + NOALLOC_COUNT_BYTES);
	float4* rayD = rayO + RTX_NOALLOC_MAX_RAYS;
	float4* hits = rayD + RTX_NOALLOC_MAX_RAYS;
	int* lists = (int*)(hits + RTX_NOALLOC_MAX_RAYS);
	static_assert(COUNT_WORDS * sizeof(unsigned) <= NOALLOC_COUNT_BYTES, "counter block");
	DevScene d = make_dev_scene(s);
	d.fault = counts + FAULT_WORD;
	// the array keeps the last call's contents: the cursors, list lengths and the fault word must start at zero
	HIPCHK(hipMemsetAsync(counts, 0, NOALLOC_COUNT_BYTES, s->stream));
	hipLaunchKernelGGL(k_pack_rays, dim3((n + 255) / 256), dim3(256), 0, s->stream, (int)n, d_origins, d_directions, rayO, rayD);
	HIPCHK(hipGetLastError());
	// no timing events and no test counters: both would need memory of their own
	const bool savedTime = s->timeEvents, savedInstr = s->instrumented;
	s->timeEvents = false;
	s->instrumented = false;
	int rc = launch_trace(s, d, false, mode == 1, rayO, rayD, hits, (int)n, counts, lists);
	s->timeEvents = savedTime;
	s->instrumented = savedInstr;
	if (rc) return rc;
	hipLaunchKernelGGL(k_export_hits_linear, dim3((n + 255) / 256), dim3(256), 0, s->stream, d, (int)n, hits, d_tri_ids, d_t_alpha_beta);
	HIPCHK(hipGetLastError());
	// the fault word travels into the scene's own pinned word (never into a stack slot the copy could outlive); the
	// wait below makes the call blocking, which is also why it cannot be part of a HIP graph capture
	*s->hNoallocFault = 0;
	hipError_t e = hipMemcpyAsync(s->hNoallocFault, counts + FAULT_WORD, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream);
	const hipError_t eSync = hipStreamSynchronize(s->stream);  // also after a failed copy: nothing of this call stays in flight
	if (e == hipSuccess) e = eSync;
	if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("rtx_trace_rays_noalloc: ") + hipGetErrorString(e));
	rc = check_fault(*s->hNoallocFault);
	if (rc) return rc;
	s->stats.rays = n;
	s->stats.rays_per_level[0] = n;
	return RTX_OK;
}

RTX_API int rtx_sample_radiance(rtx_scene* s, uint32_t frame, float* host_out) {
	if (!s || !host_out) return fail(RTX_ERR_ARG, "null argument");
	int rc = ensure_ready(s, true);
	if (rc) return rc;
	rc = begin_call(s);
	if (rc) return rc;
	const size_t ns = (size_t)s->width * s->height * 4;
	if (s->dExportRad.n < ns) HIPCHK(s->dExportRad.alloc(ns));
	HIPCHK(hipMemcpy(s->dExportRad.p, host_out, ns * sizeof(float4), hipMemcpyHostToDevice));
	rc = render_batches(s, frame, 1, 1, 0, s->dExportRad.p, nullptr, nullptr);
	if (rc) return rc;
	HIPCHK(hipMemcpy(host_out, s->dExportRad.p, ns * sizeof(float4), hipMemcpyDeviceToHost));
	return finish_stats(s);
}

// ------------------------------------------------------------------------------------------------
// Several GPUs in one process: the exchange step.

namespace {

struct Rccl {
	void* lib = nullptr;
	decltype(&ncclCommInitAll) commInitAll = nullptr;
	decltype(&ncclCommDestroy) commDestroy = nullptr;
	decltype(&ncclGroupStart) groupStart = nullptr;
	decltype(&ncclGroupEnd) groupEnd = nullptr;
	decltype(&ncclSend) send = nullptr;
	decltype(&ncclRecv) recv = nullptr;
	decltype(&ncclGetErrorString) errorString = nullptr;
	std::vector<int> devices;        // the communicators below were made for these devices, in this order
	std::vector<ncclComm_t> comms;
};
Rccl g_rccl;
std::mutex g_rcclMutex;

// Opens RCCL (the copy the process already holds, if any: a Python process that imported torch has torch's) and
// looks the entry points up.
int rccl_open(Rccl& r) {
	if (r.lib) return RTX_OK;
	const char* names[] = {"librccl.so", "librccl.so.1"};
	for (const char* n : names)
		if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
	for (const char* n : {"librccl.so.1", "librccl.so"})
		if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
	if (!r.lib) return fail(RTX_ERR_DEVICE, std::string("cannot open RCCL (librccl.so): ") + dlerror());
	r.commInitAll = (decltype(r.commInitAll))dlsym(r.lib, "ncclCommInitAll");
	r.commDestroy = (decltype(r.commDestroy))dlsym(r.lib, "ncclCommDestroy");
	r.groupStart = (decltype(r.groupStart))dlsym(r.lib, "ncclGroupStart");
	r.groupEnd = (decltype(r.groupEnd))dlsym(r.lib, "ncclGroupEnd");
	r.send = (decltype(r.send))dlsym(r.lib, "ncclSend");
	r.recv = (decltype(r.recv))dlsym(r.lib, "ncclRecv");
	r.errorString = (decltype(r.errorString))dlsym(r.lib, "ncclGetErrorString");
	if (!r.commInitAll || !r.commDestroy || !r.groupStart || !r.groupEnd || !r.send || !r.recv || !r.errorString) {
		r.lib = nullptr;
		return fail(RTX_ERR_DEVICE, "the RCCL library lacks an entry point this library needs");
	}
	return RTX_OK;
}

#define RCCLCHK(expr)                                                                                             \
	do {                                                                                                          \
		ncclResult_t r_ = (expr);                                                                                 \
		if (r_ != ncclSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.errorString(r_));   \
	} while (0)

}  // namespace

namespace {

// What one exchange moves: for every scene the tile numbers of its share, and for every sender the place of its
// message in scene 0's receive buffer.  Scene 0's own share stays where it is, unless it is the only scene: a
// single share travels to itself, so that one GPU can exercise every step.
// Messages lie one behind the other in the order of the scenes: recvOffset[r] = TILE * TILE * (tiles of the senders
// before r), in float4; recvTotal the same sum over all senders.
struct ExchangePlan {
	int firstSender = 1;
	std::vector<std::vector<int>> ids;
	std::vector<size_t> recvOffset;
	size_t recvTotal = 0;
};

void plan_offsets(ExchangePlan& p) {
	const int n = (int)p.ids.size();
	p.firstSender = n == 1 ? 0 : 1;
	p.recvOffset.assign((size_t)n, 0);
	p.recvTotal = 0;
	for (int r = p.firstSender; r < n; ++r) {
		p.recvOffset[(size_t)r] = p.recvTotal;
		p.recvTotal += p.ids[(size_t)r].size() * (size_t)(TILE * TILE);
	}
}

// The tile numbers of a scene's current share (rtx_scene_set_tiles): first, first + stride, ...
std::vector<int> share_ids(const rtx_scene* s) {
	std::vector<int> ids;
	const int n = tile_slots(s);
	ids.reserve((size_t)std::max(n, 0));
	for (int k = 0; k < n; ++k) ids.push_back(s->tileFirst + k * s->tileStride);
	return ids;
}

int ensure_id_staging(rtx_scene* s, size_t count) {
	if (s->hTileIdsCap >= count) return RTX_OK;
	if (s->hTileIds) (void)hipHostFree(s->hTileIds);
	s->hTileIds = nullptr;
	s->hTileIdsCap = 0;
	const size_t cap = std::max<size_t>(count, 4096);
	HIPCHK(hipHostMalloc((void**)&s->hTileIds, cap * sizeof(int), hipHostMallocDefault));
	s->hTileIdsCap = cap;
	return RTX_OK;
}

// Pack on every sender, one message per sender to scene 0, unpack there.  Transport: RCCL (ncclSend / ncclRecv in
// one group: one message per xGMI link, all at once) between different GPUs -- and for the single scene that sends
// to itself --, a device-to-device copy where all scenes share one GPU.
// add: scene 0 adds what it receives to its accumulator instead of overwriting it; clear: a sender's tiles are zero
// afterwards.
int run_exchange(rtx_scene* const* scenes, int n, ExchangePlan& plan, bool add, bool clear) {
	rtx_scene* s0 = scenes[0];
	plan_offsets(plan);
	const int firstSender = plan.firstSender;
	const int tilesX = (s0->width + TILE - 1) / TILE;
	const int numTiles = num_tiles(s0);
	bool sameDevice = n > 1;
	for (int r = 1; r < n; ++r) sameDevice = sameDevice && scenes[r]->device == s0->device;
	if (!sameDevice)
		for (int r = 0; r < n; ++r)
			for (int q = 0; q < r; ++q)
				if (scenes[q]->device == scenes[r]->device)
					return fail(RTX_ERR_ARG, "the scenes of one exchange live either on n different GPUs or all on the same one");
	for (int r = 0; r < n; ++r)
		for (int t : plan.ids[(size_t)r])
			if (t < 0 || t >= numTiles) return fail(RTX_ERR_ARG, "a tile number outside the image");
	if (plan.recvTotal == 0) return RTX_OK;  // nothing to move

	std::lock_guard<std::mutex> lk(g_rcclMutex);
	if (!sameDevice) {
		int rc = rccl_open(g_rccl);
		if (rc) return rc;
		std::vector<int> devices((size_t)n);
		for (int r = 0; r < n; ++r) devices[(size_t)r] = scenes[r]->device;
		if (g_rccl.devices != devices) {
			for (ncclComm_t c : g_rccl.comms)
				if (c) (void)g_rccl.commDestroy(c);
			g_rccl.comms.assign((size_t)n, nullptr);
			g_rccl.devices.clear();
			RCCLCHK(g_rccl.commInitAll(g_rccl.comms.data(), n, devices.data()));
			g_rccl.devices = devices;
		}
	}

	// scene 0: room for every message and for the tile numbers that go with them
	HIPCHK(hipSetDevice(s0->device));
	if (s0->dTileRecv.n < plan.recvTotal) HIPCHK(s0->dTileRecv.alloc(plan.recvTotal));
	const size_t recvIds = plan.recvTotal / (size_t)(TILE * TILE);
	if (s0->dTileIdsRecv.n < recvIds) HIPCHK(s0->dTileIdsRecv.alloc(recvIds));
	{
		const size_t own = firstSender == 0 ? plan.ids[0].size() : 0;  // the single scene also packs: its own list comes first
		int rc = ensure_id_staging(s0, own + recvIds);
		if (rc) return rc;
		int* stage = s0->hTileIds + own;
		for (int r = firstSender; r < n; ++r)
			std::copy(plan.ids[(size_t)r].begin(), plan.ids[(size_t)r].end(), stage + plan.recvOffset[(size_t)r] / (size_t)(TILE * TILE));
		HIPCHK(hipMemcpyAsync(s0->dTileIdsRecv.p, stage, recvIds * sizeof(int), hipMemcpyHostToDevice, s0->stream));
	}
	// senders: tile numbers up, share packed
	for (int r = firstSender; r < n; ++r) {
		rtx_scene* s = scenes[r];
		const std::vector<int>& ids = plan.ids[(size_t)r];
		if (ids.empty()) continue;
		HIPCHK(hipSetDevice(s->device));
		const size_t count = ids.size() * (size_t)(TILE * TILE);
		if (s->dTilePack.n < count) HIPCHK(s->dTilePack.alloc(count));
		if (s->dTileIds.n < ids.size()) HIPCHK(s->dTileIds.alloc(ids.size()));
		int rc = ensure_id_staging(s, ids.size() + (r == 0 ? recvIds : 0));
		if (rc) return rc;
		std::copy(ids.begin(), ids.end(), s->hTileIds);
		HIPCHK(hipMemcpyAsync(s->dTileIds.p, s->hTileIds, ids.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s->stream, s->dAccum, s->width, s->height, tilesX,
						   (const int*)s->dTileIds.p, (int)ids.size(), numTiles, s->dTilePack.p, clear ? 1 : 0);
		RTX_LAUNCHED("k_pack_tiles");
		// RTX_GATHER_POISON=1 (tests, single scene only): what was packed is wiped from the image, so the image is
		// whole again only if pack, send, receive and unpack all did their part
		if (n == 1 && std::getenv("RTX_GATHER_POISON"))
			HIPCHK(hipMemsetAsync(s->dAccum, 0xFF, (size_t)s->width * s->height * sizeof(float4), s->stream));
		if (sameDevice) {
			if (!s->exchEvent) HIPCHK(hipEventCreateWithFlags(&s->exchEvent, hipEventDisableTiming));
			HIPCHK(hipEventRecord(s->exchEvent, s->stream));
		}
	}
	if (sameDevice) {
		// every message is a copy inside the one GPU, ordered behind its sender's pack
		HIPCHK(hipSetDevice(s0->device));
		for (int r = firstSender; r < n; ++r) {
			rtx_scene* s = scenes[r];
			const size_t count = plan.ids[(size_t)r].size() * (size_t)(TILE * TILE);
			if (!count) continue;
			HIPCHK(hipStreamWaitEvent(s0->stream, s->exchEvent, 0));
			HIPCHK(hipMemcpyAsync(s0->dTileRecv.p + plan.recvOffset[(size_t)r], s->dTilePack.p, count * sizeof(float4), hipMemcpyDeviceToDevice, s0->stream));
		}
	}
	else {
		// one message per peer, all in one group: every peer's share crosses its own link to GPU 0 at the same time
		RCCLCHK(g_rccl.groupStart());
		for (int r = firstSender; r < n; ++r) {
			rtx_scene* s = scenes[r];
			const size_t count = plan.ids[(size_t)r].size() * (size_t)(TILE * TILE) * 4;  // floats
			if (!count) continue;
			ncclResult_t a = g_rccl.send(s->dTilePack.p, count, ncclFloat, 0, g_rccl.comms[(size_t)r], s->stream);
			ncclResult_t b = a == ncclSuccess ? g_rccl.recv(s0->dTileRecv.p + plan.recvOffset[(size_t)r], count, ncclFloat, r, g_rccl.comms[0], s0->stream) : a;
			if (b != ncclSuccess) {
				(void)g_rccl.groupEnd();
				return fail(RTX_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + g_rccl.errorString(b));
			}
		}
		RCCLCHK(g_rccl.groupEnd());
	}
	HIPCHK(hipSetDevice(s0->device));
	for (int r = firstSender; r < n; ++r) {
		const size_t slots = plan.ids[(size_t)r].size();
		if (!slots) continue;
		rtx_scene* s = s0;  // the launch check reports on the receiving scene
		hipLaunchKernelGGL(k_unpack_tiles, dim3((unsigned)((slots * TILE * TILE + 255) / 256)), dim3(256), 0, s->stream, s->dAccum, s->width, s->height, tilesX,
						   (const int*)(s->dTileIdsRecv.p + plan.recvOffset[(size_t)r] / (size_t)(TILE * TILE)), (int)slots, numTiles,
						   (const float4*)(s->dTileRecv.p + plan.recvOffset[(size_t)r]), add ? 1 : 0);
		RTX_LAUNCHED("k_unpack_tiles");
	}
	for (int r = n - 1; r >= 0; --r) {
		HIPCHK(hipSetDevice(scenes[r]->device));
		HIPCHK(hipStreamSynchronize(scenes[r]->stream));
	}
	return RTX_OK;
}

int check_exchange_scenes(rtx_scene* const* scenes, int n) {
	if (!scenes || n < 1) return fail(RTX_ERR_ARG, "scenes is null or n < 1");
	rtx_scene* s0 = scenes[0];
	for (int r = 0; r < n; ++r) {
		rtx_scene* s = scenes[r];
		if (!s) return fail(RTX_ERR_ARG, "a scene is null");
		for (int q = 0; q < r; ++q)
			if (scenes[q] == s) return fail(RTX_ERR_ARG, "the same scene twice in one exchange");
		if (!s->haveParams || !s0->haveParams || s->width != s0->width || s->height != s0->height) return fail(RTX_ERR_ARG, "the scenes differ in image size");
		const int rc = ensure_ready(s, false);
		if (rc) return rc;
	}
	return RTX_OK;
}

}  // namespace

RTX_API int rtx_gather_tiles(rtx_scene* const* scenes, int n) {
	int rc = check_exchange_scenes(scenes, n);
	if (rc) return rc;
	ExchangePlan plan;
	plan.ids.resize((size_t)n);
	for (int r = 0; r < n; ++r) {
		const rtx_scene* s = scenes[r];
		if (s->tileFirst != r || s->tileStride != n || s->tileCount >= 0)
			return fail(RTX_ERR_ARG, "scene r must hold the tile share (r, n): rtx_scene_set_tiles(scenes[r], r, n, -1)");
		plan.ids[(size_t)r] = share_ids(s);
	}
	return run_exchange(scenes, n, plan, false, false);
}

RTX_API int rtx_exchange_tiles(rtx_scene* const* scenes, int n) {
	int rc = check_exchange_scenes(scenes, n);
	if (rc) return rc;
	const int total = num_tiles(scenes[0]);
	ExchangePlan plan;
	plan.ids.resize((size_t)n);
	std::vector<int> owner((size_t)total, -1);
	for (int r = 0; r < n; ++r) {
		const rtx_scene* s = scenes[r];
		for (int t = 0; t < (int)s->tileLog.size() && t < total; ++t) {
			if (!s->tileLog[(size_t)t]) continue;
			if (owner[(size_t)t] >= 0)
				return fail(RTX_ERR_ARG, "tile " + std::to_string(t) + " was rendered by scenes " + std::to_string(owner[(size_t)t]) + " and " + std::to_string(r) +
											 " since the last exchange: a frame's tiles must be shared out without overlap");
			owner[(size_t)t] = r;
			plan.ids[(size_t)r].push_back(t);
		}
	}
	rc = run_exchange(scenes, n, plan, true, true);
	if (rc) return rc;
	for (int r = 0; r < n; ++r) std::fill(scenes[r]->tileLog.begin(), scenes[r]->tileLog.end(), 0);
	return RTX_OK;
}

RTX_API int rtx_scene_tile_log(rtx_scene* s, int* tiles, int capacity) {
	if (!s || (capacity > 0 && !tiles) || capacity < 0) return fail(RTX_ERR_ARG, "bad argument");
	int n = 0;
	for (size_t t = 0; t < s->tileLog.size(); ++t)
		if (s->tileLog[t]) {
			if (n < capacity) tiles[n] = (int)t;
			++n;
		}
	return n;
}

RTX_API int rtx_scene_tile_log_clear(rtx_scene* s) {
	if (!s) return fail(RTX_ERR_ARG, "scene is null");
	std::fill(s->tileLog.begin(), s->tileLog.end(), 0);
	return RTX_OK;
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
