// Host-side scene preparation behind the rtxh_* half of include/rtx.h: the parts of the
// reference that stay on the CPU (camera set-up, BVH build, single-triangle construction,
// lookup tables, tone map).  No GPU code here; librtx_host.so loads without a device.
//
// Everything is written to reproduce the reference's results bit for bit when the reference
// is compiled with g++ -O2 -mavx2 -mfma -ffp-contract=off: expressions keep Eigen's evaluation
// order (4-float dot reduces as (x+z)+(y+w)), and the BVH builder performs the same
// std::sort calls with the same comparator so that equal keys land in the same order.
// Build with -ffp-contract=off.

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_host_internal.h"
#include "rtx_image.h"

#define RTXH_API extern "C" __attribute__((visibility("default")))

namespace rtxh {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
	g_err = msg;
	return code;
}

// ---- 4-float vector arithmetic in the reference's evaluation order ---------------------------

float dot(const Vec4& a, const Vec4& b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }

Vec4 sub(const Vec4& a, const Vec4& b) { return Vec4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
Vec4 scale(const Vec4& a, float s) { return Vec4{a.x * s, a.y * s, a.z * s, a.w * s}; }
Vec4 divide(const Vec4& a, float s) { return Vec4{a.x / s, a.y / s, a.z / s, a.w / s}; }

// Eigen normalized(): unchanged when the squared norm is not positive (Eigen/src/Core/Dot.h:124-134)
Vec4 normalized(const Vec4& a) {
	float z = dot(a, a);
	if (z > 0.0f) return divide(a, std::sqrt(z));
	return a;
}

// Eigen cross3 (Eigen/src/Geometry/arch/Geometry_SSE.h:72-75)
Vec4 cross3(const Vec4& a, const Vec4& b) {
	return Vec4{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, a.w * b.w - a.w * b.w};
}

// Camera::setCamera, src/Camera.cpp:4-32.
void camera(const Vec4& origin, const Vec4& viewPoint, float focal, float rotateDeg, int width, int height, Vec4 out[4]) {
	const float fW = (float)width, fH = (float)height;
	const Vec4 dir = normalized(sub(viewPoint, origin));
	const Vec4 right = normalized(cross3(dir, Vec4{0.0f, 1.0f, 0.0f, 0.0f}));
	const Vec4 down = normalized(cross3(dir, right));
	Vec4 upHalf, rightHalf;
	const float aspect = fW / fH;
	if (aspect >= 1.5f) {
		rightHalf = scale(right, 18.0f / focal);
		upHalf = scale(down, 18.0f / focal / aspect);
	}
	else {
		upHalf = scale(down, 12.0f / focal);
		rightHalf = scale(right, 12.0f / focal * aspect);
	}
	const Vec4 corner = sub(sub(dir, rightHalf), upHalf);
	const Vec4 stepR = divide(rightHalf, fW / 2.0f);
	const Vec4 stepD = divide(upHalf, fH / 2.0f);

	// rotation about the view direction: AngleAxisf(angle, dir).toRotationMatrix()
	// (Eigen/src/Geometry/AngleAxis.h:219-244) placed in the upper-left 3x3 of a zero 4x4.
	const float angle = (3.1415926f / 180.0f) * rotateDeg;
	const float sn = std::sin(angle), cs = std::cos(angle);
	const float ax[3] = {dir.x, dir.y, dir.z};
	float sa[3], ca[3];
	for (int i = 0; i < 3; ++i) {
		sa[i] = sn * ax[i];
		ca[i] = (1.0f - cs) * ax[i];
	}
	float R[4][4];
	std::memset(R, 0, sizeof(R));
	float t;
	t = ca[0] * ax[1];
	R[0][1] = t - sa[2];
	R[1][0] = t + sa[2];
	t = ca[0] * ax[2];
	R[0][2] = t + sa[1];
	R[2][0] = t - sa[1];
	t = ca[1] * ax[2];
	R[1][2] = t - sa[0];
	R[2][1] = t + sa[0];
	for (int i = 0; i < 3; ++i) R[i][i] = ca[i] * ax[i] + cs;
	// 4x4 * 4x1 as Eigen's coefficient-based product evaluates it: first column times v0, then one
	// fused multiply-add per further column (pmadd is an FMA when the reference is built with -mfma).
	auto rotate = [&](const Vec4& v) {
		const float in[4] = {v.x, v.y, v.z, v.w};
		float o[4];
		for (int i = 0; i < 4; ++i) {
			float acc = R[i][0] * in[0];
			for (int k = 1; k < 4; ++k) acc = std::fma(R[i][k], in[k], acc);
			o[i] = acc;
		}
		return Vec4{o[0], o[1], o[2], o[3]};
	};
	out[0] = origin;
	out[1] = rotate(corner);
	out[2] = rotate(stepR);
	out[3] = rotate(stepD);
}

// RayTracer::addTriangle, src/RayTracer.cpp:108-134.
void make_triangle(const Vec4& v0, const Vec4& v1, const Vec4& v2, const Vec4& side, const Vec4& color, bool metal,
				   bool emit, bool transp, float rough, float ior, TriangleRec& T) {
	std::memset(&T, 0, sizeof(T));
	T.v[0] = v0;
	T.v[1] = v1;
	T.v[2] = v2;
	Vec4 pn = normalized(cross3(sub(v1, v0), sub(v2, v0)));
	if (dot(side, pn) < 0.0f) pn = Vec4{-pn.x, -pn.y, -pn.z, -pn.w};
	T.pn = pn;
	for (int i = 0; i < 3; ++i) T.n[i] = pn;
	T.color = color;
	T.metal = metal;
	T.emit = emit;
	T.transp = transp;
	T.rough = rough;
	T.ior = ior;
	T.tex = -1;
}

// ---- BVH::buildTree, src/BVH.cpp:51-160 ------------------------------------------------------
//
// Same tree, different machinery: leaf records live in one flat array that is sorted in place
// by value (the reference sorts unique_ptr elements; the permutation depends only on the
// comparator, which is the same), interior nodes come from a pool instead of one heap
// allocation each, child boxes are merged from the per-triangle boxes (min of mins is the
// same float as the min over vertices), and independent subtrees are built by several
// threads.  The breadth-first numbering at the end is the reference's (:144-159).

namespace {

struct LeafRec {
	float bmin[3], bmax[3], center[3];
	int index;
};

struct BNode {
	float bmin[3], bmax[3];
	int left = -1, right = -1, tri = -1;
};

struct Range {
	int node, start, end;
};

inline void box_reset(float* mn, float* mx) {
	for (int c = 0; c < 3; ++c) {
		mn[c] = FLT_MAX;
		mx[c] = -FLT_MAX;
	}
}
inline void box_grow(float* mn, float* mx, const LeafRec& l) {
	for (int c = 0; c < 3; ++c) {
		mn[c] = l.bmin[c] < mn[c] ? l.bmin[c] : mn[c];
		mx[c] = l.bmax[c] > mx[c] ? l.bmax[c] : mx[c];
	}
}

// Builds the subtree under pool[work.node] for leaf range [start,end); new nodes are appended to
// `pool` (a pool private to the calling thread, or the shared one during the serial top phase).
// Ranges no larger than `handoff` are pushed to `deferred` instead of being split (0 = never).
void build_range(std::vector<BNode>& pool, std::vector<LeafRec>& leaf, Range first, int handoff,
				 std::vector<Range>* deferred) {
	std::vector<Range> st;
	st.push_back(first);
	while (!st.empty()) {
		Range w = st.back();
		st.pop_back();
		const int count = w.end - w.start;
		if (deferred && count <= handoff && !(w.node == first.node)) {
			deferred->push_back(w);
			continue;
		}
		if (count <= 2) {
			// one or two triangles left: they become the children (src/BVH.cpp:80-90)
			for (int k = 0; k < count; ++k) {
				const LeafRec& l = leaf[w.start + k];
				BNode c;
				std::memcpy(c.bmin, l.bmin, 12);
				std::memcpy(c.bmax, l.bmax, 12);
				c.tri = l.index;
				pool.push_back(c);
				(k == 0 ? pool[w.node].left : pool[w.node].right) = (int)pool.size() - 1;
			}
			continue;
		}
		// longest axis of the node's own box, first maximum wins (src/BVH.cpp:93-102)
		const BNode& nd = pool[w.node];
		float len = nd.bmax[0] - nd.bmin[0];
		int axis = 0;
		for (int c = 1; c < 3; ++c) {
			float d = nd.bmax[c] - nd.bmin[c];
			if (d > len) {
				len = d;
				axis = c;
			}
		}
		std::sort(leaf.begin() + w.start, leaf.begin() + w.end,
				  [axis](const LeafRec& a, const LeafRec& b) { return a.center[axis] < b.center[axis]; });
		const int split = (w.start + w.end + 1) / 2;  // src/BVH.cpp:111
		BNode l, r;
		box_reset(l.bmin, l.bmax);
		for (int i = w.start; i < split; ++i) box_grow(l.bmin, l.bmax, leaf[i]);
		box_reset(r.bmin, r.bmax);
		for (int i = split; i < w.end; ++i) box_grow(r.bmin, r.bmax, leaf[i]);
		pool.push_back(l);
		const int li = (int)pool.size() - 1;
		pool.push_back(r);
		const int ri = li + 1;
		pool[w.node].left = li;
		pool[w.node].right = ri;
		st.push_back(Range{li, w.start, split});
		st.push_back(Range{ri, split, w.end});
	}
}

}  // namespace

int build_bvh(const TriangleRec* tris, uint32_t n, std::vector<NodeRec>& out, int nthreads) {
	std::vector<LeafRec> leaf(n);
	BNode root;
	box_reset(root.bmin, root.bmax);
	for (uint32_t i = 0; i < n; ++i) {
		LeafRec& l = leaf[i];
		box_reset(l.bmin, l.bmax);
		for (int j = 0; j < 3; ++j) {
			const float p[3] = {tris[i].v[j].x, tris[i].v[j].y, tris[i].v[j].z};
			for (int c = 0; c < 3; ++c) {
				l.bmin[c] = p[c] < l.bmin[c] ? p[c] : l.bmin[c];
				l.bmax[c] = p[c] > l.bmax[c] ? p[c] : l.bmax[c];
			}
		}
		for (int c = 0; c < 3; ++c) l.center[c] = (l.bmin[c] + l.bmax[c]) * 0.5f;
		l.index = (int)i;
		box_grow(root.bmin, root.bmax, l);
	}

	// Phase 1 (serial): split from the root until ranges are small enough to hand to workers.
	// Phase 2 (parallel): every deferred range is an independent subtree; each worker builds into its
	// own pool and the pools are spliced behind the shared one with index fix-ups.
	std::vector<BNode> pool;
	pool.reserve(3 * (size_t)n + 2);
	pool.push_back(root);
	if (n > 0) {
		if (nthreads < 1) nthreads = 1;
		const bool parallel = nthreads > 1 && n >= 16384;
		std::vector<Range> deferred;
		const int handoff = parallel ? std::max<int>(4096, (int)(n / (uint32_t)(nthreads * 8))) : 0;
		build_range(pool, leaf, Range{0, 0, (int)n}, handoff, parallel ? &deferred : nullptr);
		if (parallel && !deferred.empty()) {
			struct Job {
				Range r;
				std::vector<BNode> local;  // local[0] mirrors the subtree root
			};
			std::vector<Job> jobs(deferred.size());
			for (size_t k = 0; k < deferred.size(); ++k) jobs[k].r = deferred[k];
			std::atomic<size_t> next(0);
			auto worker = [&] {
				for (;;) {
					size_t k = next.fetch_add(1);
					if (k >= jobs.size()) break;
					Job& jb = jobs[k];
					jb.local.reserve(3 * (size_t)(jb.r.end - jb.r.start) + 2);
					jb.local.push_back(pool[jb.r.node]);
					build_range(jb.local, leaf, Range{0, jb.r.start, jb.r.end}, 0, nullptr);
				}
			};
			std::vector<std::thread> th;
			for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
			for (auto& t : th) t.join();
			for (Job& jb : jobs) {
				const int base = (int)pool.size() - 1;  // local index i>0 -> pool index base+i
				auto fix = [&](int c) { return c > 0 ? base + c : c; };
				pool[jb.r.node].left = fix(jb.local[0].left);
				pool[jb.r.node].right = fix(jb.local[0].right);
				for (size_t i = 1; i < jb.local.size(); ++i) {
					BNode b = jb.local[i];
					b.left = fix(b.left);
					b.right = fix(b.right);
					pool.push_back(b);
				}
			}
		}
	}

	// breadth-first linearisation, root at 0, left before right (src/BVH.cpp:144-159)
	out.clear();
	out.reserve(pool.size());
	std::vector<int> queue;
	queue.reserve(pool.size());
	queue.push_back(0);
	auto emit = [&](const BNode& b) {
		NodeRec r;
		std::memset(&r, 0, sizeof(r));
		const bool empty = (n == 0);
		r.bmin = Vec4{b.bmin[0], b.bmin[1], b.bmin[2], empty ? FLT_MAX : 0.0f};
		r.bmax = Vec4{b.bmax[0], b.bmax[1], b.bmax[2], empty ? -FLT_MAX : 0.0f};
		r.left = -1;
		r.right = -1;
		r.tri = b.tri;
		out.push_back(r);
	};
	emit(pool[0]);
	for (size_t i = 0; i < queue.size(); ++i) {
		const BNode& b = pool[queue[i]];
		if (b.left >= 0) {
			queue.push_back(b.left);
			out[i].left = (int)queue.size() - 1;
			emit(pool[b.left]);
		}
		if (b.right >= 0) {
			queue.push_back(b.right);
			out[i].right = (int)queue.size() - 1;
			emit(pool[b.right]);
		}
	}
	return (int)out.size();
}

// src/Triangle.cpp:8-21 and src/Texture.cpp:49-51
void tables(float* cos512, float* sin512, float* gamma256) {
	for (int i = 0; i < 512; ++i) {
		cos512[i] = cosf(i * 3.1415926f * 2.0f / 512);
		sin512[i] = sinf(i * 3.1415926f * 2.0f / 512);
	}
	for (int c = 0; c < 256; ++c) gamma256[c] = powf(c / 255.0f, 2.2f);
}

// src/RayTracer.cpp:248-257
void tonemap(const float* accum, int frame, size_t npix, uint8_t* rgb8) {
	for (size_t p = 0; p < npix; ++p)
		for (int k = 0; k < 3; ++k) {
			float averaged = accum[p * 4 + k] / frame;
			float gammaCorrected = powf(averaged, 1.0f / 2.2f);
			int clipNum = (int)lroundf(gammaCorrected * 255.0f);
			if (clipNum > 255) clipNum = 255;
			if (clipNum < 0) clipNum = 0;
			rgb8[p * 3 + k] = (uint8_t)clipNum;
		}
}

// The tone map above as a step table over the bit patterns of non-negative floats (see rtx.h).
bool tonemap_table(uint32_t* bits, uint8_t* values, int cap, int* n_out, uint8_t special[2]) {
	auto f = [](uint32_t b) {
		float x[4] = {0, 0, 0, 0};
		std::memcpy(&x[0], &b, 4);
		uint8_t out[3];
		tonemap(x, 1, 1, out);
		return out[0];
	};
	// Seeds spread over [0, 1e15]; between two neighbouring seeds the function is monotone.  Beyond 1.7e15 the
	// scaled value no longer fits the `int` the reference narrows lroundf's result to (src/RayTracer.cpp:251)
	// and the byte it stores there is whatever the wrap-around leaves; the table ends before that, so the
	// device saturates such inputs (and +inf) to the last entry, 255.
	const float seedsF[] = {0.0f, 1e-30f, 1e-12f, 1e-6f, 1e-3f, 0.5f, 1.0f, 4.0f, 1e6f, 1e15f};
	std::vector<uint32_t> seeds;
	for (float v : seedsF) {
		uint32_t b;
		std::memcpy(&b, &v, 4);
		seeds.push_back(b);
	}
	std::vector<std::pair<uint32_t, uint8_t>> steps;
	steps.push_back({0u, f(0u)});
	for (size_t k = 0; k + 1 < seeds.size(); ++k) {
		// all breakpoints in (seeds[k], seeds[k+1]] by bisection of every sub-interval whose ends differ
		std::vector<std::pair<uint32_t, uint32_t>> st;
		st.push_back({seeds[k], seeds[k + 1]});
		std::vector<std::pair<uint32_t, uint8_t>> found;
		while (!st.empty()) {
			auto [lo, hi] = st.back();
			st.pop_back();
			if (f(lo) == f(hi)) continue;
			if (hi == lo + 1) {
				found.push_back({hi, f(hi)});
				continue;
			}
			const uint32_t mid = lo + (hi - lo) / 2;
			st.push_back({lo, mid});
			st.push_back({mid, hi});
		}
		std::sort(found.begin(), found.end());
		for (auto& e : found) steps.push_back(e);
	}
	if ((int)steps.size() > cap) return false;
	for (size_t i = 0; i < steps.size(); ++i) {
		bits[i] = steps[i].first;
		values[i] = steps[i].second;
	}
	*n_out = (int)steps.size();
	special[0] = f(0xBF800000u);  // -1: powf gives NaN; -0 gives +0; both must agree for one entry to do
	if (f(0x80000000u) != special[0]) return false;
	special[1] = f(0x7FC00000u);
	return true;
}


}  // namespace rtxh

// ================================================================================================

using namespace rtxh;

RTXH_API const char* rtxh_last_error(void) { return g_err.c_str(); }

RTXH_API int rtxh_camera(const float eye[3], const float view_point[3], float focal, float rotate_deg, int width,
						 int height, float out16[16]) {
	if (!eye || !view_point || !out16) return fail(RTX_ERR_ARG, "null argument");
	if (width <= 0 || height <= 0) return fail(RTX_ERR_ARG, "image size must be positive");
	Vec4 o[4];
	camera(Vec4{eye[0], eye[1], eye[2], 0.0f}, Vec4{view_point[0], view_point[1], view_point[2], 0.0f}, focal,
		   rotate_deg, width, height, o);
	std::memcpy(out16, o, 64);
	return RTX_OK;
}

RTXH_API int rtxh_build_bvh(const void* triangles, uint32_t n, void* nodes_out, uint32_t capacity) {
	if ((!triangles && n) || !nodes_out) return fail(RTX_ERR_ARG, "null argument");
	if (n > (1u << 30)) return fail(RTX_ERR_ARG, "too many triangles");
	std::vector<NodeRec> nodes;
	unsigned hw = std::thread::hardware_concurrency();
	int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 16);
	int cnt = build_bvh((const TriangleRec*)triangles, n, nodes, nthreads);
	if ((uint32_t)cnt > capacity) return fail(RTX_ERR_ARG, "nodes_out too small: need " + std::to_string(cnt) + " records");
	std::memcpy(nodes_out, nodes.data(), (size_t)cnt * sizeof(NodeRec));
	return cnt;
}

RTXH_API int rtxh_make_triangle(const float v0[3], const float v1[3], const float v2[3], const float normal_side[3],
								const float color[3], int is_metal, int is_light_emitting, int is_transparent,
								float specular_roughness, float refractive_index, void* triangle_out) {
	if (!v0 || !v1 || !v2 || !normal_side || !color || !triangle_out) return fail(RTX_ERR_ARG, "null argument");
	auto v = [](const float* p) { return Vec4{p[0], p[1], p[2], 0.0f}; };
	TriangleRec T;
	make_triangle(v(v0), v(v1), v(v2), v(normal_side), v(color), is_metal != 0, is_light_emitting != 0,
				  is_transparent != 0, specular_roughness, refractive_index, T);
	std::memcpy(triangle_out, &T, sizeof(T));
	return RTX_OK;
}

RTXH_API int rtxh_tables(float cos512[512], float sin512[512], float gamma256[256]) {
	if (!cos512 || !sin512 || !gamma256) return fail(RTX_ERR_ARG, "null argument");
	tables(cos512, sin512, gamma256);
	return RTX_OK;
}

RTXH_API long long rtxh_load_texture(const char* path, int* width, int* height, uint8_t* rgb, size_t capacity) {
	if (!path || !width || !height) return fail(RTX_ERR_ARG, "null argument");
	Image img;
	if (!load_texture_file(path, img)) return fail(RTX_ERR_ARG, std::string("Can't load texture in ") + path);
	*width = img.w;
	*height = img.h;
	if (rgb && capacity >= img.rgb.size()) std::memcpy(rgb, img.rgb.data(), img.rgb.size());
	return (long long)img.rgb.size();
}

RTXH_API int rtxh_tonemap_table(uint32_t* bits, uint8_t* values, int cap, int* n_out, uint8_t special[2]) {
	if (!bits || !values || !n_out || !special) return fail(RTX_ERR_ARG, "null argument");
	if (!tonemap_table(bits, values, cap, n_out, special))
		return fail(RTX_ERR_STATE, "the tone map of this host is not a step table of at most `cap` entries");
	return RTX_OK;
}

RTXH_API int rtxh_tonemap(const float* accum_rgba, int frame, size_t npix, uint8_t* rgb8) {
	if (!accum_rgba || !rgb8) return fail(RTX_ERR_ARG, "null argument");
	if (frame <= 0) return fail(RTX_ERR_ARG, "frame must be >= 1");
	tonemap(accum_rgba, frame, npix, rgb8);
	return RTX_OK;
}
