// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's render hot path.
//
// Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this
// library, and only as the checker. The product path (raytracer_amd/) never includes,
// links or calls anything in this directory.
//
// Plain scalar C++ (no Eigen, no SSE) that restates, operation for operation, what the
// reference computes when its sources are compiled with
//   g++ -O2 -mavx2 -mfma -ffp-contract=off   (see oracle/build_ref.sh).
// Every function cites the reference lines it follows. Parity of this restatement is
// pinned against oracle/_ref/libref.so (the reference's own objects) by
// tests/test_oracle_vs_reference.py and, where that library cannot travel, against the
// golden vectors in tests/golden/ that were produced from it (tests/golden/make_golden.py).
//
// Two deliberate differences from the reference, both required for reproducibility:
//  * RNG. The reference draws from two thread_local std::mt19937 (src/Triangle.cpp:96,122)
//    whose consumption order depends on thread scheduling. Here every draw is either taken
//    from two explicit mt19937 objects in the reference's depth-first order (RNG_MT, used
//    only to pin this file against the reference) or from an integer counter hash of
//    (seed, frame, pixel, sub-sample, path node, lobe, ray, dimension) (RNG_COUNTER, the mode
//    the HIP path implements bit-for-bit).
//  * Undefined behaviour in the reference is given a meaning: texture coordinates are
//    clamped to [0,1] (NaN -> 0) before rounding (src/Texture.cpp:45-47 reads out of bounds);
//    on total internal reflection the refracted child is not traced and contributes zero
//    (src/Triangle.cpp:164-165 returns an uninitialised direction); an empty scene has a
//    root node without children (src/BVH.cpp:80-83 indexes an empty list).
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off; see oracle/Makefile)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <tuple>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

struct V4 {
	float x, y, z, w;
};
inline V4 mk(float x, float y, float z, float w = 0.0f) { return V4{x, y, z, w}; }
inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator-(V4 a) { return V4{-a.x, -a.y, -a.z, -a.w}; }
inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 operator*(float s, V4 a) { return V4{s * a.x, s * a.y, s * a.z, s * a.w}; }
inline V4 operator/(V4 a, float s) { return V4{a.x / s, a.y / s, a.z / s, a.w / s}; }
inline V4 cmul(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

// Eigen's 4-float dot: lane products reduced as (l0+l2)+(l1+l3)
// (include/Eigen/src/Core/arch/SSE/PacketMath.h:548-549).
inline float dot4(V4 a, V4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }

// Eigen normalized()/normalize(): divide by sqrt(squaredNorm) only when it is > 0
// (include/Eigen/src/Core/Dot.h:124-151).
inline V4 normalized(V4 a) {
	float z = dot4(a, a);
	if (z > 0.0f) return a / sqrtf(z);
	return a;
}

// Eigen cross3 (include/Eigen/src/Geometry/arch/Geometry_SSE.h:72-75).
inline V4 cross3(V4 a, V4 b) {
	return V4{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, a.w * b.w - a.w * b.w};
}

// In-memory layout of the reference's Triangle (include/RayTracer/Triangle.h:6-34), 176 bytes.
struct Tri {
	V4 v[3];
	V4 n[3];
	V4 pn;
	V4 color;
	float uv[3][2];
	float rough;
	float ior;
	int32_t tex;
	uint8_t emit, metal, transp, pad0;
	uint8_t pad1[8];
};
static_assert(sizeof(Tri) == 176, "Triangle layout");

// In-memory layout of the reference's LinearNode (include/RayTracer/BVH.h:35-43), 48 bytes.
struct Node {
	V4 bmin, bmax;
	int32_t left, right, tri;
	int32_t pad;
};
static_assert(sizeof(Node) == 48, "LinearNode layout");

struct Tex {
	std::vector<uint8_t> rgb;
	int w = 0, h = 0;
};

struct Ray {
	V4 o, d;
};

constexpr int RNG_COUNTER = 0;
constexpr int RNG_MT = 1;

struct Rng {
	int mode = RNG_COUNTER;
	std::mt19937 diff, spec;  // RNG_MT: stand for the two thread_local generators
};

// Counter hash shared (by specification, not by code) with the HIP kernels.
inline uint32_t rt_mix(uint32_t h, uint32_t v) {
	h = (h + 0x7F4A7C15u) ^ (v * 0x9E3779B1u);
	h ^= h >> 16;
	h *= 0x85EBCA6Bu;
	h ^= h >> 13;
	h *= 0xC2B2AE35u;
	h ^= h >> 16;
	return h;
}
inline uint32_t rt_sample_key(uint64_t seed, uint32_t frame, uint32_t sample) {
	uint32_t k = rt_mix((uint32_t)seed, (uint32_t)(seed >> 32));
	k = rt_mix(k, frame);
	return rt_mix(k, sample);
}
inline uint32_t rt_child_key(uint32_t key, uint32_t ordinal) { return rt_mix(key, 1u + ordinal); }
inline uint32_t rt_draw(uint32_t key, uint32_t lobe, uint32_t k, uint32_t dim) {
	return rt_mix(key, 0x80000000u | (lobe << 20) | (k << 1) | dim);
}

struct Scene {
	int width = 0, height = 0, maxDepth = 1, nDiff = 1, nSpec = 1;
	V4 bg{0, 0, 0, 0};
	uint64_t seed = 0;
	V4 camO{}, camLUC{}, camRS{}, camDS{};
	std::vector<Tri> tris;
	std::vector<Node> nodes;
	std::vector<Tex> texs;
	bool hasSky = false;
	float skyBrightness = 1.0f;
	int skyTex[6] = {-1, -1, -1, -1, -1, -1};
	float cosT[512], sinT[512];
	std::atomic<uint64_t> rayCount{0};
	std::atomic<uint64_t> tirCount{0};
	float worst[8] = {0};  // candidates, depth, origin xyz, direction xyz of the worst ray of the last render call

	Scene() {
		// src/Triangle.cpp:6-21
		for (int i = 0; i < 512; ++i) {
			cosT[i] = cosf(i * 3.1415926f * 2.0f / 512);
			sinT[i] = sinf(i * 3.1415926f * 2.0f / 512);
		}
	}
};

// ---------------------------------------------------------------- camera (src/Camera.cpp)

// Camera::setCamera, src/Camera.cpp:4-32. The 4x4 * 4x1 products at :29-31 are Eigen's
// coefficient-based product, which accumulates column by column with pmadd; with -mfma that
// is a fused multiply-add (include/Eigen/src/Core/ProductEvaluators.h, etor_product_packet_impl).
void set_camera(Scene& s, V4 origin, V4 viewPoint, float focal, float rotateAngle) {
	float fWidth = (float)s.width;
	float fHeight = (float)s.height;
	s.camO = origin;
	V4 direction = normalized(viewPoint - origin);
	V4 right = normalized(cross3(direction, mk(0, 1, 0, 0)));
	V4 down = normalized(cross3(direction, right));
	V4 upHalf, rightHalf;
	if (fWidth / fHeight >= 1.5f) {
		rightHalf = right * (18.0f / focal);
		upHalf = down * (18.0f / focal / (fWidth / fHeight));
	}
	else {
		upHalf = down * (12.0f / focal);
		rightHalf = right * (12.0f / focal * (fWidth / fHeight));
	}
	V4 luc = direction - rightHalf - upHalf;
	V4 rs = rightHalf / (fWidth / 2.0f);
	V4 ds = upHalf / (fHeight / 2.0f);

	// Eigen::AngleAxisf::toRotationMatrix (include/Eigen/src/Geometry/AngleAxis.h:219-244)
	float angle = (3.1415926f / 180.0f) * rotateAngle;
	float ax = direction.x, ay = direction.y, az = direction.z;
	float sn = sinf(angle), c = cosf(angle);
	float sax = sn * ax, say = sn * ay, saz = sn * az;
	float c1x = (1.0f - c) * ax, c1y = (1.0f - c) * ay, c1z = (1.0f - c) * az;
	float m[4][4] = {{0}};
	float tmp;
	tmp = c1x * ay;
	m[0][1] = tmp - saz;
	m[1][0] = tmp + saz;
	tmp = c1x * az;
	m[0][2] = tmp + say;
	m[2][0] = tmp - say;
	tmp = c1y * az;
	m[1][2] = tmp - sax;
	m[2][1] = tmp + sax;
	m[0][0] = c1x * ax + c;
	m[1][1] = c1y * ay + c;
	m[2][2] = c1z * az + c;
	auto mul = [&](V4 v) {
		float in[4] = {v.x, v.y, v.z, v.w};
		float out[4];
		for (int i = 0; i < 4; ++i) {
			float r = m[i][0] * in[0];
			r = fmaf(m[i][1], in[1], r);
			r = fmaf(m[i][2], in[2], r);
			r = fmaf(m[i][3], in[3], r);
			out[i] = r;
		}
		return V4{out[0], out[1], out[2], out[3]};
	};
	s.camLUC = mul(luc);
	s.camRS = mul(rs);
	s.camDS = mul(ds);
}

// Camera::getRay, src/Camera.cpp:37-49.
const float kXOff[4] = {-0.25f, 0.25f, -0.25f, 0.25f};
const float kYOff[4] = {-0.25f, -0.25f, 0.25f, 0.25f};
inline Ray get_ray(const Scene& s, int x, int y, int i) {
	float xTemp = x + kXOff[i];
	float yTemp = y + kYOff[i];
	Ray r;
	r.o = s.camO;
	r.d = s.camLUC + xTemp * s.camRS + yTemp * s.camDS;
	return r;
}

// ---------------------------------------------------------------- BVH (src/BVH.cpp)

inline V4 vmin(V4 a, V4 b) {
	// _mm_min_ps(a,b) = a < b ? a : b
	return V4{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z, a.w < b.w ? a.w : b.w};
}
inline V4 vmax(V4 a, V4 b) {
	return V4{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z, a.w > b.w ? a.w : b.w};
}
inline float comp(V4 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }

struct LeafTemp {
	V4 bmin, bmax, center;
	int index;
};
struct TNode {
	V4 bmin, bmax;
	std::unique_ptr<TNode> left, right;
	int tri;
	TNode(int t, V4 a, V4 b) : bmin(a), bmax(b), tri(t) {}
};

// BVH::buildTree, src/BVH.cpp:51-160. The sort at :105 is std::sort on unique_ptr elements;
// the same libstdc++ std::sort on raw pointers with the same comparator performs the same
// comparisons and moves, so ties between equal centres resolve identically.
void build_tree(Scene& s) {
	const auto& T = s.tris;
	s.nodes.clear();
	V4 mn = mk(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
	V4 mx = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
	std::vector<std::unique_ptr<LeafTemp>> leaf(T.size());
	for (size_t i = 0; i < T.size(); ++i) {
		V4 tmn = mk(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
		V4 tmx = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
		for (int j = 0; j < 3; ++j) {
			tmn = vmin(tmn, T[i].v[j]);
			tmx = vmax(tmx, T[i].v[j]);
			mn = vmin(mn, T[i].v[j]);
			mx = vmax(mx, T[i].v[j]);
		}
		leaf[i].reset(new LeafTemp{tmn, tmx, (tmn + tmx) * 0.5f, (int)i});
	}
	auto root = std::make_unique<TNode>(-1, mn, mx);
	if (!T.empty()) {
		std::vector<std::tuple<TNode*, int, int>> st;
		st.emplace_back(root.get(), 0, (int)T.size());
		do {
			auto [node, start, end] = st.back();
			st.pop_back();
			if (end - start <= 2) {
				const auto& a = leaf[start];
				node->left = std::make_unique<TNode>(a->index, a->bmin, a->bmax);
				if (end - start == 2) {
					const auto& b = leaf[start + 1];
					node->right = std::make_unique<TNode>(b->index, b->bmin, b->bmax);
				}
			}
			else {
				V4 diff = node->bmax - node->bmin;
				float axisLength = diff.x;
				int axis = 0;
				for (int i = 1; i < 3; ++i) {
					if (comp(diff, i) > axisLength) {
						axisLength = comp(diff, i);
						axis = i;
					}
				}
				std::sort(leaf.begin() + start, leaf.begin() + end, [axis](const auto& l, const auto& r) {
					return comp(l->center, axis) < comp(r->center, axis);
				});
				int split = (start + end + 1) / 2;
				mn = mk(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
				mx = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
				for (int i = start; i < split; ++i) {
					const auto& vt = T[leaf[i]->index].v;
					for (int j = 0; j < 3; ++j) {
						mn = vmin(mn, vt[j]);
						mx = vmax(mx, vt[j]);
					}
				}
				node->left = std::make_unique<TNode>(-1, mn, mx);
				st.emplace_back(node->left.get(), start, split);
				mn = mk(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
				mx = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
				for (int i = split; i < end; ++i) {
					const auto& vt = T[leaf[i]->index].v;
					for (int j = 0; j < 3; ++j) {
						mn = vmin(mn, vt[j]);
						mx = vmax(mx, vt[j]);
					}
				}
				node->right = std::make_unique<TNode>(-1, mn, mx);
				st.emplace_back(node->right.get(), split, end);
			}
		} while (!st.empty());
	}
	// breadth-first linearisation, src/BVH.cpp:144-159
	std::vector<TNode*> q;
	q.push_back(root.get());
	auto push = [&](TNode* t) {
		Node n;
		std::memset(&n, 0, sizeof(n));
		n.bmin = t->bmin;
		n.bmax = t->bmax;
		n.left = -1;
		n.right = -1;
		n.tri = t->tri;
		s.nodes.push_back(n);
	};
	push(root.get());
	for (size_t i = 0; i < q.size(); ++i) {
		TNode* p = q[i];
		if (p->left) {
			q.push_back(p->left.get());
			s.nodes[i].left = (int)q.size() - 1;
			push(p->left.get());
		}
		if (p->right) {
			q.push_back(p->right.get());
			s.nodes[i].right = (int)q.size() - 1;
			push(p->right.get());
		}
	}
	// unique_ptr chains of a deep tree are torn down iteratively to spare the stack
	std::vector<std::unique_ptr<TNode>> pending;
	pending.push_back(std::move(root));
	while (!pending.empty()) {
		auto n = std::move(pending.back());
		pending.pop_back();
		if (n->left) pending.push_back(std::move(n->left));
		if (n->right) pending.push_back(std::move(n->right));
	}
}

// AABB::hit, src/BVH.cpp:14-41.
inline bool aabb_hit(V4 bmin, V4 bmax, const Ray& r) {
	float inv[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
	float t0v[3] = {(bmin.x - r.o.x) * inv[0], (bmin.y - r.o.y) * inv[1], (bmin.z - r.o.z) * inv[2]};
	float t1v[3] = {(bmax.x - r.o.x) * inv[0], (bmax.y - r.o.y) * inv[1], (bmax.z - r.o.z) * inv[2]};
	float t0o[3], t1o[3];
	for (int i = 0; i < 3; ++i) {
		uint32_t bits;
		std::memcpy(&bits, &inv[i], 4);
		bool neg = (bits >> 31) != 0;  // _mm_blendv_ps selects on the sign bit
		t0o[i] = neg ? t1v[i] : t0v[i];
		t1o[i] = neg ? t0v[i] : t1v[i];
	}
	float tmin = t0o[0];
	float tmax = t1o[0];
	for (int i = 1; i < 3; ++i) {
		float t0 = t0o[i];
		float t1 = t1o[i];
		tmin = t0 > tmin ? t0 : tmin;
		tmax = t1 < tmax ? t1 : tmax;
	}
	if (tmax < tmin) return false;
	return true;
}

// BVH::hit, src/BVH.cpp:162-192.
void bvh_hit(const Scene& s, const Ray& r, std::vector<int>& result) {
	result.clear();
	std::vector<int> stack;
	stack.reserve(64);
	stack.push_back(0);
	do {
		int idx = stack.back();
		stack.pop_back();
		const Node& n = s.nodes[idx];
		if (aabb_hit(n.bmin, n.bmax, r)) {
			if (n.tri >= 0)
				result.push_back(n.tri);
			else {
				if (n.left > 0) stack.push_back(n.left);
				if (n.right > 0) stack.push_back(n.right);
			}
		}
	} while (!stack.empty());
}

// ---------------------------------------------------------------- triangle (src/Triangle.cpp)

struct HitRes {
	float alpha, beta, t;
};

// Triangle::hit, src/Triangle.cpp:23-93. Only lanes 1 and 2 of the three rows influence the
// outputs after the first pivot, so lanes 0 and 3 are not carried.
inline HitRes tri_hit(const Tri& T, const Ray& r) {
	const HitRes miss{FLT_MAX, FLT_MAX, FLT_MAX};
	float temp = dot4(T.pn, T.v[0] - r.o);
	float t = temp / dot4(T.pn, r.d);
	if (!(t > 0.001f)) return miss;
	V4 P = r.o + t * r.d;
	V4 a1 = T.v[0] - T.v[2];
	V4 a2 = T.v[1] - T.v[2];
	V4 b = P - T.v[2];
	float R[3][3] = {{a1.x, a2.x, b.x}, {a1.y, a2.y, b.y}, {a1.z, a2.z, b.z}};
	auto swap_rows = [&](int i, int j) {
		if (i == j) return;
		for (int k = 0; k < 3; ++k) std::swap(R[i][k], R[j][k]);
	};
	int index = 0;
	float mx = fabsf(R[0][0]);
	if (fabsf(R[1][0]) > mx) {
		mx = fabsf(R[1][0]);
		index = 1;
	}
	if (fabsf(R[2][0]) > mx) index = 2;
	swap_rows(0, index);
	float div = R[0][0];
	R[0][1] = R[0][1] / div;
	R[0][2] = R[0][2] / div;
	for (int i = 1; i < 3; ++i) {
		float m = R[i][0];
		R[i][1] = fmaf(-R[0][1], m, R[i][1]);
		R[i][2] = fmaf(-R[0][2], m, R[i][2]);
	}
	index = 1;
	if (fabsf(R[2][1]) > fabsf(R[1][1])) index = 2;
	swap_rows(1, index);
	div = R[1][1];
	R[1][2] = R[1][2] / div;
	float m = R[0][1];
	R[0][2] = fmaf(-R[1][2], m, R[0][2]);
	float alpha = R[0][2];
	float beta = R[1][2];
	bool accept = (0.0f <= alpha) & (alpha <= 1.0f) & (0.0f <= beta) & (beta <= 1.0f) & (alpha + beta <= 1.0f);
	if (accept) return HitRes{alpha, beta, t};
	return miss;
}

// Offset vector shared by diffuse/specular, src/Triangle.cpp:106-113 and :133-140:
// horizontal * sin(theta) + cos(theta) * UnitZ, all four lanes as Eigen evaluates them.
inline V4 scatter_offset(const Scene& s, unsigned phi, unsigned theta) {
	V4 horizontal = mk(s.cosT[phi], s.sinT[phi], 0.0f, 0.0f);
	return horizontal * s.sinT[theta] + s.cosT[theta] * mk(0.0f, 0.0f, 1.0f, 0.0f);
}

inline void draw_angles(const Scene&, Rng& rng, uint32_t key, int lobe, int k, unsigned& phi, unsigned& theta) {
	if (rng.mode == RNG_MT) {
		std::mt19937& g = lobe == 0 ? rng.spec : rng.diff;
		phi = (unsigned)g() & 0x1FFu;
		theta = ((unsigned)g() & 0x1FFu) >> 1;
	}
	else {
		phi = rt_draw(key, (uint32_t)lobe, (uint32_t)k, 0) & 0x1FFu;
		theta = (rt_draw(key, (uint32_t)lobe, (uint32_t)k, 1) & 0x1FFu) >> 1;
	}
}

// Triangle::diffuse, src/Triangle.cpp:95-119.
void tri_diffuse(const Scene& s, Rng& rng, uint32_t key, V4 normal, const Ray& r, int num, std::vector<V4>& out) {
	V4 tempNormal = dot4(r.d, normal) < 0.0f ? normal : -normal;
	out.resize(num);
	for (int i = 0; i < num; ++i) {
		unsigned phi, theta;
		draw_angles(s, rng, key, 1, i, phi, theta);
		out[i] = tempNormal + scatter_offset(s, phi, theta);
	}
}

// Triangle::specular, src/Triangle.cpp:121-146.
void tri_specular(const Scene& s, Rng& rng, uint32_t key, const Tri& T, V4 normal, const Ray& r, int num,
				  std::vector<V4>& out) {
	float normalProjection = dot4(r.d, normal);
	V4 direction = r.d - (2.0f * normalProjection) * normal;
	out.resize(num);
	for (int i = 0; i < num; ++i) {
		unsigned phi, theta;
		draw_angles(s, rng, key, 0, i, phi, theta);
		out[i] = direction + scatter_offset(s, phi, theta) * T.rough;
	}
}

// Triangle::refract, src/Triangle.cpp:149-179. Returns false on total internal reflection.
bool tri_refract(const Tri& T, V4 normal, const Ray& r, float& proportion, V4& outRay) {
	float dot = dot4(r.d, normal);
	float indexRatio = dot > 0.0f ? T.ior : (1.0f / T.ior);
	V4 tempNormal = dot > 0.0f ? -normal : normal;
	float cosineTheta = fabsf(dot);
	float squareSine1 = 1.0f - cosineTheta * cosineTheta;
	float squareSine2 = (indexRatio * indexRatio) * squareSine1;
	if (squareSine2 > 1.0f) {
		proportion = 0.0f;
		outRay = mk(0, 0, 0, 0);
		return false;
	}
	float cosine2 = sqrtf(1.0f - squareSine2);
	outRay = indexRatio * (r.d + cosineTheta * tempNormal) - tempNormal * cosine2;
	float temp = (1.0f - indexRatio) / (1.0f + indexRatio);
	float r0 = temp * temp;
	float squareTemp = (1.0f - cosineTheta) * (1.0f - cosineTheta);
	float pow5 = squareTemp * squareTemp * (1.0f - cosineTheta);
	float reflectProportion = r0 + (1.0f - r0) * pow5;
	proportion = 1.0f - reflectProportion;
	return true;
}

// ---------------------------------------------------------------- textures

inline float clamp01(float u) {
	u = (u >= 0.0f) ? u : 0.0f;  // NaN -> 0
	u = (u <= 1.0f) ? u : 1.0f;
	return u;
}

// Texture::sampleTexture, src/Texture.cpp:41-53, with the coordinates clamped to [0,1].
inline V4 sample_texture(const Tex& t, float u, float v) {
	u = clamp01(u);
	v = clamp01(v);
	float uMax = (float)(t.w - 1);
	float vMax = (float)(t.h - 1);
	int sx = (int)lroundf(u * uMax);
	int sy = (int)lroundf(v * vMax);
	const uint8_t* c = t.rgb.data() + ((size_t)sy * t.w + sx) * 3;
	float r = powf(c[0] / 255.0f, 2.2f);
	float g = powf(c[1] / 255.0f, 2.2f);
	float b = powf(c[2] / 255.0f, 2.2f);
	return mk(r, g, b, 0.0f);
}

// Skybox::sampleBackground, src/Skybox.cpp:32-84.
V4 sample_sky(const Scene& s, const Ray& ray) {
	V4 d = ray.d;
	for (int i = 0; i < 6; ++i) {
		float divisor = 0.0f, u = 0.0f, v = 0.0f;
		V4 p;
		switch (i) {
		case 0:
			divisor = d.z;
			p = d / divisor;
			u = (1.0f - p.x) * 0.5f;
			v = (1.0f + p.y) * 0.5f;
			break;
		case 1:
			divisor = -d.z;
			p = d / divisor;
			u = (1.0f + p.x) * 0.5f;
			v = (1.0f + p.y) * 0.5f;
			break;
		case 2:
			divisor = -d.x;
			p = d / divisor;
			u = (1.0f - p.z) * 0.5f;
			v = (1.0f + p.y) * 0.5f;
			break;
		case 3:
			divisor = d.x;
			p = d / divisor;
			u = (p.z + 1.0f) * 0.5f;
			v = (p.y + 1.0f) * 0.5f;
			break;
		case 4:
			divisor = d.y;
			p = d / divisor;
			u = (p.x + 1.0f) * 0.5f;
			v = (p.z + 1.0f) * 0.5f;
			break;
		case 5:
			divisor = -d.y;
			p = d / divisor;
			u = (p.x + 1.0f) * 0.5f;
			v = (p.z - 1.0f) * 0.5f;
			break;
		}
		if (divisor < 0.0f || u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) continue;
		return sample_texture(s.texs[s.skyTex[i]], u, v) * s.skyBrightness;
	}
	return mk(0, 0, 0, 0);
}

// ---------------------------------------------------------------- shading (src/RayTracer.cpp)

struct Closest {
	int index;
	float t, alpha, beta;
};

// Head of RayTracer::color, src/RayTracer.cpp:137-151.
inline Closest closest_hit(const Scene& s, const Ray& r, std::vector<int>& scratch) {
	bvh_hit(s, r, scratch);
	Closest c{-1, FLT_MAX, 0.0f, 0.0f};
	for (size_t i = 0; i < scratch.size(); ++i) {
		HitRes h = tri_hit(s.tris[scratch[i]], r);
		if (h.t < c.t) {
			c.index = scratch[i];
			c.t = h.t;
			c.alpha = h.alpha;
			c.beta = h.beta;
		}
	}
	return c;
}

struct Ctx {
	const Scene* s;
	Rng rng;
	std::vector<int> scratch;
	uint64_t rays = 0;
	uint64_t tir = 0;
	// diagnostics: the ray with the longest candidate list seen by this context
	size_t worstCandidates = 0;
	Ray worstRay{};
	int worstDepth = 0;
};

// RayTracer::color, src/RayTracer.cpp:136-226.
V4 color(Ctx& cx, int depth, const Ray& r, uint32_t key) {
	const Scene& s = *cx.s;
	cx.rays++;
	Closest c = closest_hit(s, r, cx.scratch);
	if (cx.scratch.size() > cx.worstCandidates) {
		cx.worstCandidates = cx.scratch.size();
		cx.worstRay = r;
		cx.worstDepth = depth;
	}
	if (c.index == -1) {
		if (s.hasSky) return sample_sky(s, r);
		return s.bg;
	}
	const Tri& T = s.tris[c.index];
	float alpha = c.alpha, beta = c.beta;
	V4 hitPoint = r.o + c.t * r.d;
	V4 normal = alpha * T.n[0] + beta * T.n[1] + (1.0f - (alpha + beta)) * T.n[2];
	normal = normalized(normal);
	float cosine = dot4(normal, r.d);
	if (depth == s.maxDepth || (!T.transp && cosine >= 0.0f)) return mk(0, 0, 0, 0);
	if (T.emit) return T.color;

	const int S = s.nSpec, D = s.nDiff;
	std::vector<V4> specOut, diffOut;
	if (T.metal) {
		tri_specular(s, cx.rng, key, T, normal, r, S, specOut);
		V4 sc = mk(0, 0, 0, 0);
		for (int i = 0; i < S; ++i) sc = sc + color(cx, depth + 1, Ray{hitPoint, specOut[i]}, rt_child_key(key, i));
		return cmul(sc, T.color) / (float)S;
	}
	if (T.transp) {
		tri_specular(s, cx.rng, key, T, normal, r, S, specOut);
		V4 sc = mk(0, 0, 0, 0);
		for (int i = 0; i < S; ++i) sc = sc + color(cx, depth + 1, Ray{hitPoint, specOut[i]}, rt_child_key(key, i));
		sc = sc / (float)S;
		float rp;
		V4 refr;
		V4 rc = mk(0, 0, 0, 0);
		if (tri_refract(T, normal, r, rp, refr))
			rc = color(cx, depth + 1, Ray{hitPoint, refr}, rt_child_key(key, S));
		else
			cx.tir++;
		return rp * rc + (1.0f - rp) * sc;
	}
	tri_specular(s, cx.rng, key, T, normal, r, S, specOut);
	V4 sc = mk(0, 0, 0, 0);
	for (int i = 0; i < S; ++i) sc = sc + color(cx, depth + 1, Ray{hitPoint, specOut[i]}, rt_child_key(key, i));
	sc = sc * (0.04f / (float)S);
	tri_diffuse(s, cx.rng, key, normal, r, D, diffOut);
	V4 dc = mk(0, 0, 0, 0);
	for (int i = 0; i < D; ++i) dc = dc + color(cx, depth + 1, Ray{hitPoint, diffOut[i]}, rt_child_key(key, S + i));
	dc = cmul(dc, T.color) / (float)D;
	V4 outc = sc + dc * fabsf(dot4(normal, r.d));
	if (T.tex < 0) return outc;
	float g = 1.0f - (alpha + beta);
	float u = (alpha * T.uv[0][0] + beta * T.uv[1][0]) + g * T.uv[2][0];
	float v = (alpha * T.uv[0][1] + beta * T.uv[1][1]) + g * T.uv[2][1];
	return cmul(outc, sample_texture(s.texs[T.tex], u, v));
}

// Tone map of one accumulator value, src/RayTracer.cpp:248-257.
inline uint8_t tone(float acc, int frame) {
	float averaged = acc / frame;
	float gammaCorrected = powf(averaged, 1.0f / 2.2f);
	int clipNum = (int)lroundf(gammaCorrected * 255.0f);
	if (clipNum > 255) clipNum = 255;
	if (clipNum < 0) clipNum = 0;
	return (uint8_t)clipNum;
}

inline V4 ld3(const float* p) { return mk(p[0], p[1], p[2], 0.0f); }

}  // namespace

// ================================================================= C ABI (tests only)

ORC_API void* orc_create() { return new Scene(); }
ORC_API void orc_destroy(void* h) { delete (Scene*)h; }

ORC_API void orc_set_params(void* h, int width, int height, int maxDepth, int nDiff, int nSpec, const float* bg,
							uint64_t seed) {
	Scene& s = *(Scene*)h;
	s.width = width;
	s.height = height;
	s.maxDepth = maxDepth;
	s.nDiff = nDiff;
	s.nSpec = nSpec;
	s.bg = ld3(bg);
	s.seed = seed;
}

ORC_API void orc_set_camera(void* h, const float* eye, const float* at, float focal, float rotateDeg) {
	set_camera(*(Scene*)h, ld3(eye), ld3(at), focal, rotateDeg);
}

// out[16]: origin, leftUpCorner, rightStep, downStep
ORC_API void orc_get_camera(void* h, float* out) {
	Scene& s = *(Scene*)h;
	std::memcpy(out + 0, &s.camO, 16);
	std::memcpy(out + 4, &s.camLUC, 16);
	std::memcpy(out + 8, &s.camRS, 16);
	std::memcpy(out + 12, &s.camDS, 16);
}

ORC_API void orc_put_camera(void* h, const float* in) {
	Scene& s = *(Scene*)h;
	std::memcpy(&s.camO, in + 0, 16);
	std::memcpy(&s.camLUC, in + 4, 16);
	std::memcpy(&s.camRS, in + 8, 16);
	std::memcpy(&s.camDS, in + 12, 16);
}

ORC_API void orc_get_ray(void* h, int x, int y, float* out) {
	Scene& s = *(Scene*)h;
	for (int i = 0; i < 4; ++i) {
		Ray r = get_ray(s, x, y, i);
		std::memcpy(out + i * 8, &r.o, 16);
		std::memcpy(out + i * 8 + 4, &r.d, 16);
	}
}

ORC_API void orc_set_triangles(void* h, const void* tris, int n) {
	Scene& s = *(Scene*)h;
	s.tris.resize((size_t)n);
	if (n) std::memcpy((void*)s.tris.data(), tris, sizeof(Tri) * (size_t)n);
}

ORC_API int orc_num_triangles(void* h) { return (int)((Scene*)h)->tris.size(); }

ORC_API void orc_get_triangles(void* h, void* out) {
	Scene& s = *(Scene*)h;
	std::memcpy(out, (const void*)s.tris.data(), sizeof(Tri) * s.tris.size());
}

// RayTracer::addTriangle, src/RayTracer.cpp:108-134.
ORC_API void orc_add_triangle_cfg(void* h, const float* v0, const float* v1, const float* v2, const float* side,
								  const float* colorv, int metal, int emit, int transp, float rough, float ior) {
	Scene& s = *(Scene*)h;
	Tri T;
	std::memset(&T, 0, sizeof(T));
	T.v[0] = ld3(v0);
	T.v[1] = ld3(v1);
	T.v[2] = ld3(v2);
	T.pn = normalized(cross3(T.v[1] - T.v[0], T.v[2] - T.v[0]));
	if (dot4(ld3(side), T.pn) < 0.0f) T.pn = -T.pn;
	for (int i = 0; i < 3; ++i) T.n[i] = T.pn;
	T.color = ld3(colorv);
	T.metal = metal != 0;
	T.emit = emit != 0;
	T.transp = transp != 0;
	T.rough = rough;
	T.ior = ior;
	T.tex = -1;
	s.tris.push_back(T);
}

// rgb is the image as the reference holds it in memory after its flipped load
// (src/Texture.cpp:6-7): row 0 is the bottom row of the file.
ORC_API int orc_add_texture(void* h, const uint8_t* rgb, int w, int hgt) {
	Scene& s = *(Scene*)h;
	Tex t;
	t.w = w;
	t.h = hgt;
	t.rgb.assign(rgb, rgb + (size_t)w * hgt * 3);
	s.texs.push_back(std::move(t));
	return (int)s.texs.size() - 1;
}

ORC_API void orc_set_skybox(void* h, float brightness, const int* texIds) {
	Scene& s = *(Scene*)h;
	s.hasSky = true;
	s.skyBrightness = brightness;
	for (int i = 0; i < 6; ++i) s.skyTex[i] = texIds[i];
}

ORC_API void orc_sample_texture(void* h, int index, float u, float v, float* out) {
	Scene& s = *(Scene*)h;
	V4 c = sample_texture(s.texs[index], u, v);
	std::memcpy(out, &c, 16);
}

ORC_API void orc_sample_sky(void* h, const float* dir, float* out) {
	Scene& s = *(Scene*)h;
	Ray r{mk(0, 0, 0, 0), ld3(dir)};
	V4 c = sample_sky(s, r);
	std::memcpy(out, &c, 16);
}

ORC_API int orc_build_bvh(void* h) {
	Scene& s = *(Scene*)h;
	build_tree(s);
	return (int)s.nodes.size();
}

ORC_API int orc_num_nodes(void* h) { return (int)((Scene*)h)->nodes.size(); }

ORC_API void orc_get_nodes(void* h, void* out) {
	Scene& s = *(Scene*)h;
	std::memcpy(out, (const void*)s.nodes.data(), sizeof(Node) * s.nodes.size());
}

ORC_API void orc_set_nodes(void* h, const void* nodes, int n) {
	Scene& s = *(Scene*)h;
	s.nodes.resize((size_t)n);
	std::memcpy((void*)s.nodes.data(), nodes, sizeof(Node) * (size_t)n);
}

ORC_API int orc_aabb_hit(const float* bmin, const float* bmax, const float* o, const float* d) {
	Ray r{ld3(o), ld3(d)};
	return aabb_hit(ld3(bmin), ld3(bmax), r) ? 1 : 0;
}

ORC_API int orc_bvh_hit(void* h, const float* o, const float* d, int* out, int cap) {
	Scene& s = *(Scene*)h;
	std::vector<int> res;
	bvh_hit(s, Ray{ld3(o), ld3(d)}, res);
	for (size_t i = 0; i < res.size() && (int)i < cap; ++i) out[i] = res[i];
	return (int)res.size();
}

ORC_API void orc_tri_hit(void* h, int tri, int n, const float* o, const float* d, float* out) {
	Scene& s = *(Scene*)h;
	for (int i = 0; i < n; ++i) {
		HitRes r = tri_hit(s.tris[tri], Ray{ld3(o + 3 * i), ld3(d + 3 * i)});
		out[4 * i + 0] = r.alpha;
		out[4 * i + 1] = r.beta;
		out[4 * i + 2] = r.t;
		out[4 * i + 3] = r.t;
	}
}

ORC_API void orc_closest_hits(void* h, int n, const float* o, const float* d, int* ids, float* tab) {
	Scene& s = *(Scene*)h;
	std::vector<int> scratch;
	for (int i = 0; i < n; ++i) {
		Closest c = closest_hit(s, Ray{ld3(o + 3 * i), ld3(d + 3 * i)}, scratch);
		ids[i] = c.index;
		tab[3 * i + 0] = c.t;
		tab[3 * i + 1] = c.alpha;
		tab[3 * i + 2] = c.beta;
	}
}

// lobe 0 = specular, 1 = diffuse. RNG_MT with freshly default-seeded generators, like a
// new thread of the reference.
ORC_API void orc_scatter_mt(void* h, int tri, int lobe, const float* normal, const float* o, const float* d, int num,
							float* out) {
	Scene& s = *(Scene*)h;
	Rng rng;
	rng.mode = RNG_MT;
	std::vector<V4> res;
	Ray r{ld3(o), ld3(d)};
	if (lobe == 0)
		tri_specular(s, rng, 0, s.tris[tri], ld3(normal), r, num, res);
	else
		tri_diffuse(s, rng, 0, ld3(normal), r, num, res);
	for (int i = 0; i < num; ++i) std::memcpy(out + 4 * i, &res[i], 16);
}

// out[6] = proportion, direction xyzw, tir flag
ORC_API void orc_refract(void* h, int tri, const float* normal, const float* o, const float* d, float* out) {
	Scene& s = *(Scene*)h;
	float p;
	V4 dir;
	bool ok = tri_refract(s.tris[tri], ld3(normal), Ray{ld3(o), ld3(d)}, p, dir);
	out[0] = p;
	std::memcpy(out + 1, &dir, 16);
	out[5] = ok ? 0.0f : 1.0f;
}

// color() with the reference's RNG discipline: fresh_each != 0 restarts both generators at
// the default seed for every ray (one fresh reference thread per ray), otherwise one pair of
// generators runs through all n rays. tir[n] (may be null) receives the number of total
// internal reflections met under each ray, which are the rays where the reference's own
// result is undefined.
ORC_API void orc_color_mt(void* h, int depth, int n, const float* o, const float* d, float* out, int fresh_each,
						  int* tir) {
	Scene& s = *(Scene*)h;
	Ctx cx;
	cx.s = &s;
	cx.rng.mode = RNG_MT;
	for (int i = 0; i < n; ++i) {
		if (fresh_each) {
			cx.rng.diff = std::mt19937();
			cx.rng.spec = std::mt19937();
		}
		uint64_t before = cx.tir;
		V4 c = color(cx, depth, Ray{ld3(o + 3 * i), ld3(d + 3 * i)}, 0);
		std::memcpy(out + 4 * i, &c, 16);
		if (tir) tir[i] = (int)(cx.tir - before);
	}
}

// color() in counter mode for explicit rays with explicit path keys.
ORC_API void orc_color_keys(void* h, int depth, int n, const float* o, const float* d, const uint32_t* keys,
							float* out) {
	Scene& s = *(Scene*)h;
	Ctx cx;
	cx.s = &s;
	cx.rng.mode = RNG_COUNTER;
	for (int i = 0; i < n; ++i) {
		V4 c = color(cx, depth, Ray{ld3(o + 3 * i), ld3(d + 3 * i)}, keys[i]);
		std::memcpy(out + 4 * i, &c, 16);
	}
}

ORC_API uint32_t orc_sample_key(uint64_t seed, uint32_t frame, uint32_t sample) {
	return rt_sample_key(seed, frame, sample);
}
ORC_API uint32_t orc_child_key(uint32_t key, uint32_t ordinal) { return rt_child_key(key, ordinal); }
ORC_API uint32_t orc_draw(uint32_t key, uint32_t lobe, uint32_t k, uint32_t dim) { return rt_draw(key, lobe, k, dim); }

// Per-sample radiance of frame `frame` for the pixel rectangle [x0,x1) x [y0,y1) in counter
// mode: out[(y-y0)*(x1-x0)+(x-x0)][4 sub-samples][4].
ORC_API void orc_sample_radiance(void* h, int frame, int x0, int y0, int x1, int y1, float* out) {
	Scene& s = *(Scene*)h;
	Ctx cx;
	cx.s = &s;
	cx.rng.mode = RNG_COUNTER;
	size_t k = 0;
	for (int y = y0; y < y1; ++y)
		for (int x = x0; x < x1; ++x)
			for (int i = 0; i < 4; ++i) {
				uint32_t sample = ((uint32_t)y * (uint32_t)s.width + (uint32_t)x) * 4u + (uint32_t)i;
				V4 c = color(cx, 0, get_ray(s, x, y, i), rt_sample_key(s.seed, (uint32_t)frame, sample));
				std::memcpy(out + 4 * k, &c, 16);
				++k;
			}
}

// Loop body of RayTracer::render, src/RayTracer.cpp:237-259, for rows [row0,row1) of frame
// `frame` (1-based), counter RNG, std::thread workers pulling rows from an atomic counter.
// accum: H*W*4 floats in/out. out8: H*W*3 or null. Returns wall seconds.
ORC_API double orc_render_rows(void* h, int frame, int row0, int row1, float* accum, uint8_t* out8, int nthreads) {
	Scene& s = *(Scene*)h;
	std::atomic<int> next(row0);
	std::atomic<uint64_t> rays(0), tirs(0);
	auto worker = [&] {
		Ctx cx;
		cx.s = &s;
		cx.rng.mode = RNG_COUNTER;
		for (;;) {
			int row = next.fetch_add(1);
			if (row >= row1) break;
			for (int col = 0; col < s.width; ++col) {
				V4 temp = mk(0, 0, 0, 0);
				for (int i = 0; i < 4; ++i) {
					uint32_t sample = ((uint32_t)row * (uint32_t)s.width + (uint32_t)col) * 4u + (uint32_t)i;
					temp = temp + color(cx, 0, get_ray(s, col, row, i), rt_sample_key(s.seed, (uint32_t)frame, sample));
				}
				float* px = accum + ((size_t)row * s.width + col) * 4;
				V4 add = temp * 0.25f;
				px[0] += add.x;
				px[1] += add.y;
				px[2] += add.z;
				px[3] += add.w;
				if (out8)
					for (int k = 0; k < 3; ++k) out8[((size_t)row * s.width + col) * 3 + k] = tone(px[k], frame);
			}
		}
		rays += cx.rays;
		tirs += cx.tir;
		static std::mutex mu;
		std::lock_guard<std::mutex> lk(mu);
		if ((float)cx.worstCandidates > s.worst[0]) {
			s.worst[0] = (float)cx.worstCandidates;
			s.worst[1] = (float)cx.worstDepth;
			s.worst[2] = cx.worstRay.o.x; s.worst[3] = cx.worstRay.o.y; s.worst[4] = cx.worstRay.o.z;
			s.worst[5] = cx.worstRay.d.x; s.worst[6] = cx.worstRay.d.y; s.worst[7] = cx.worstRay.d.z;
		}
	};
	s.worst[0] = 0.0f;
	auto t0 = std::chrono::steady_clock::now();
	if (nthreads < 1) nthreads = 1;
	std::vector<std::thread> pool;
	for (int i = 0; i < nthreads; ++i) pool.emplace_back(worker);
	for (auto& th : pool) th.join();
	auto t1 = std::chrono::steady_clock::now();
	s.rayCount += rays.load();
	s.tirCount += tirs.load();
	return std::chrono::duration<double>(t1 - t0).count();
}


// The same loop body for a list of pixels instead of whole rows: pixel k is (xy[2k], xy[2k+1]); a pixel listed twice is
// accumulated twice.  Lets a test look at blocks scattered over a large image at the price of a few rows.
ORC_API double orc_render_pixels(void* h, int frame, int n, const int* xy, float* accum, int nthreads) {
	Scene& s = *(Scene*)h;
	std::atomic<int> next(0);
	std::atomic<uint64_t> rays(0), tirs(0);
	auto worker = [&] {
		Ctx cx;
		cx.s = &s;
		cx.rng.mode = RNG_COUNTER;
		for (;;) {
			const int k0 = next.fetch_add(16);
			if (k0 >= n) break;
			for (int k = k0; k < std::min(n, k0 + 16); ++k) {
				const int col = xy[2 * k], row = xy[2 * k + 1];
				if (col < 0 || col >= s.width || row < 0 || row >= s.height) continue;
				V4 temp = mk(0, 0, 0, 0);
				for (int i = 0; i < 4; ++i) {
					uint32_t sample = ((uint32_t)row * (uint32_t)s.width + (uint32_t)col) * 4u + (uint32_t)i;
					temp = temp + color(cx, 0, get_ray(s, col, row, i), rt_sample_key(s.seed, (uint32_t)frame, sample));
				}
				float* px = accum + ((size_t)row * s.width + col) * 4;
				V4 add = temp * 0.25f;
				px[0] += add.x;
				px[1] += add.y;
				px[2] += add.z;
				px[3] += add.w;
			}
		}
		rays += cx.rays;
		tirs += cx.tir;
	};
	auto t0 = std::chrono::steady_clock::now();
	if (nthreads < 1) nthreads = 1;
	std::vector<std::thread> pool;
	for (int i = 0; i < nthreads; ++i) pool.emplace_back(worker);
	for (auto& th : pool) th.join();
	auto t1 = std::chrono::steady_clock::now();
	s.rayCount += rays.load();
	s.tirCount += tirs.load();
	return std::chrono::duration<double>(t1 - t0).count();
}
// Diagnostics: candidates, depth, origin, direction of the ray with the longest BVH::hit list in
// the last orc_render_rows call.
ORC_API void orc_worst_ray(void* h, float* out8) { std::memcpy(out8, ((Scene*)h)->worst, 32); }

ORC_API uint64_t orc_ray_count(void* h) { return ((Scene*)h)->rayCount.load(); }
ORC_API uint64_t orc_tir_count(void* h) { return ((Scene*)h)->tirCount.load(); }
ORC_API void orc_reset_counts(void* h) {
	((Scene*)h)->rayCount = 0;
	((Scene*)h)->tirCount = 0;
}

// Primary-ray closest hits of the pixel rectangle: ids[npix*4], tab[npix*4][3] = t, alpha, beta.
ORC_API void orc_primary_hits(void* h, int x0, int y0, int x1, int y1, int* ids, float* tab) {
	Scene& s = *(Scene*)h;
	std::vector<int> scratch;
	size_t k = 0;
	for (int y = y0; y < y1; ++y)
		for (int x = x0; x < x1; ++x)
			for (int i = 0; i < 4; ++i) {
				Closest c = closest_hit(s, get_ray(s, x, y, i), scratch);
				ids[k] = c.index;
				tab[3 * k + 0] = c.t;
				tab[3 * k + 1] = c.alpha;
				tab[3 * k + 2] = c.beta;
				++k;
			}
}

ORC_API void orc_tonemap(const float* accum, int frame, int npix, uint8_t* out8) {
	for (int p = 0; p < npix; ++p)
		for (int k = 0; k < 3; ++k) out8[(size_t)p * 3 + k] = tone(accum[(size_t)p * 4 + k], frame);
}

// Host-computed tables the HIP path uploads so that no device libm is involved:
// cos/sin (src/Triangle.cpp:8-21) and powf(c/255, 2.2f) (src/Texture.cpp:49-51).
ORC_API void orc_tables(float* cos512, float* sin512, float* gamma256) {
	for (int i = 0; i < 512; ++i) {
		cos512[i] = cosf(i * 3.1415926f * 2.0f / 512);
		sin512[i] = sinf(i * 3.1415926f * 2.0f / 512);
	}
	for (int c = 0; c < 256; ++c) gamma256[c] = powf(c / 255.0f, 2.2f);
}
