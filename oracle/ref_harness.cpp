// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// C-ABI driver around the *reference's own* hot-path translation units
// (/root/reference/src/{Camera,BVH,Triangle,Texture,Skybox,ImageIO,RayTracer}.cpp),
// which oracle/build_ref.sh compiles from where they lie.  This file contains no
// reference code: it only constructs the reference's objects and calls their methods,
// so that the CPU restatement in oracle/rt_oracle.cpp (and through it the HIP path) can
// be pinned against what the reference really computes.
//
// RayTracer::render / loadModel / parseConfigFile need TBB and assimp libraries that do
// not exist for Linux in the reference tree.  They are never referenced from here, and
// the link step garbage-collects them (-ffunction-sections + --gc-sections +
// -z defs), so no stand-in library is needed: RayTracer::color (src/RayTracer.cpp:136),
// RayTracer::addTriangle (:108) and everything they call link on their own.
//
// The reference keeps its RNGs as function-local thread_local std::mt19937 objects
// (src/Triangle.cpp:96,122) that cannot be reseeded from outside.  Every entry point that
// reaches them therefore runs its work on a fresh std::thread, whose generators start at
// the default seed.

#define private public
#include <RayTracer/RayTracer.h>
#undef private

#include <cstddef>
#include <cstring>
#include <thread>
#include <atomic>
#include <cmath>
#include <chrono>

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
inline Eigen::Vector4f v4(const float* p) { return Eigen::Vector4f(p[0], p[1], p[2], 0.0f); }
inline Ray mkray(const float* o, const float* d) { return Ray(v4(o), v4(d)); }
}  // namespace

// which: 0 Triangle, 1 LinearNode, 2 Ray, 3..13 offsets inside Triangle, 14..16 inside LinearNode
REF_API int ref_layout(int which) {
	switch (which) {
	case 0: return (int)sizeof(Triangle);
	case 1: return (int)sizeof(LinearNode);
	case 2: return (int)sizeof(Ray);
	case 3: return (int)offsetof(Triangle, vertexPosition);
	case 4: return (int)offsetof(Triangle, vertexNormal);
	case 5: return (int)offsetof(Triangle, planeNormal);
	case 6: return (int)offsetof(Triangle, color);
	case 7: return (int)offsetof(Triangle, uvCoordinate);
	case 8: return (int)offsetof(Triangle, specularRoughness);
	case 9: return (int)offsetof(Triangle, refractiveIndex);
	case 10: return (int)offsetof(Triangle, textureIndex);
	case 11: return (int)offsetof(Triangle, isLightEmitting);
	case 12: return (int)offsetof(Triangle, isMetal);
	case 13: return (int)offsetof(Triangle, isTransparent);
	case 14: return (int)offsetof(LinearNode, left);
	case 15: return (int)offsetof(LinearNode, right);
	case 16: return (int)offsetof(LinearNode, vertexIndex);
	}
	return -1;
}

REF_API void* ref_create() { return new RayTracer(); }
REF_API void ref_destroy(void* h) { delete (RayTracer*)h; }

REF_API void ref_set_params(void* h, int width, int height, int maxDepth, int diffuseNum, int specularNum,
							const float* bg) {
	auto* t = (RayTracer*)h;
	t->width = width;
	t->height = height;
	t->maxRecursionDepth = maxDepth;
	t->diffuseRayNum = diffuseNum;
	t->specualrRayNum = specularNum;
	t->backgroundColor = v4(bg);
}

// Calls Camera::setCamera (src/Camera.cpp:4) with the image size set by ref_set_params.
REF_API void ref_set_camera(void* h, const float* eye, const float* at, float focal, float rotateDeg) {
	auto* t = (RayTracer*)h;
	t->camera.setCamera(v4(eye), v4(at), focal, rotateDeg, t->width, t->height);
}

// out[16]: origin, leftUpCorner, rightStep, downStep (4 floats each)
REF_API void ref_get_camera(void* h, float* out) {
	auto* t = (RayTracer*)h;
	std::memcpy(out + 0, t->camera.origin.data(), 16);
	std::memcpy(out + 4, t->camera.leftUpCorner.data(), 16);
	std::memcpy(out + 8, t->camera.rightStep.data(), 16);
	std::memcpy(out + 12, t->camera.downStep.data(), 16);
}

REF_API void ref_put_camera(void* h, const float* in) {
	auto* t = (RayTracer*)h;
	std::memcpy(t->camera.origin.data(), in + 0, 16);
	std::memcpy(t->camera.leftUpCorner.data(), in + 4, 16);
	std::memcpy(t->camera.rightStep.data(), in + 8, 16);
	std::memcpy(t->camera.downStep.data(), in + 12, 16);
}

// Camera::getRay (src/Camera.cpp:40). out[4][8] = origin xyzw, direction xyzw per sub-sample.
REF_API void ref_get_ray(void* h, int x, int y, float* out) {
	auto* t = (RayTracer*)h;
	auto rays = t->camera.getRay(x, y);
	for (int i = 0; i < 4; ++i) {
		std::memcpy(out + i * 8, rays[i].origin.data(), 16);
		std::memcpy(out + i * 8 + 4, rays[i].direction.data(), 16);
	}
}

// Raw append of n triangles given in the reference's in-memory Triangle layout.
REF_API void ref_add_triangles(void* h, const void* tris, int n) {
	auto* t = (RayTracer*)h;
	size_t old = t->trianglesArray.size();
	t->trianglesArray.resize(old + n);
	std::memcpy((void*)(t->trianglesArray.data() + old), tris, sizeof(Triangle) * (size_t)n);
}

REF_API int ref_num_triangles(void* h) { return (int)((RayTracer*)h)->trianglesArray.size(); }

REF_API void ref_get_triangles(void* h, void* out) {
	auto* t = (RayTracer*)h;
	std::memcpy(out, (const void*)t->trianglesArray.data(), sizeof(Triangle) * t->trianglesArray.size());
}

// RayTracer::addTriangle (src/RayTracer.cpp:108)
REF_API void ref_add_triangle_cfg(void* h, const float* v0, const float* v1, const float* v2, const float* side,
								  const float* color, int metal, int emit, int transp, float rough, float ior) {
	auto* t = (RayTracer*)h;
	t->addTriangle(v4(v0), v4(v1), v4(v2), v4(side), v4(color), metal != 0, emit != 0, transp != 0, rough, ior);
}

// Texture::Texture(path) (src/Texture.cpp:4). Returns the texture index or -1.
REF_API int ref_add_texture_file(void* h, const char* path) {
	auto* t = (RayTracer*)h;
	Texture tex(path);
	if (!tex.hasTexture()) return -1;
	t->texturesArray.push_back(std::move(tex));
	return (int)t->texturesArray.size() - 1;
}

// The pixels Texture::Texture(path) (src/Texture.cpp:4-8) decoded through the reference's stb_image: writes
// width and height, copies width*height*3 bytes when they fit.  Returns the byte count, or -1.
REF_API long long ref_texture_pixels(const char* path, int* width, int* height, unsigned char* out, size_t capacity) {
	Texture tex(path);
	if (!tex.hasTexture()) return -1;
	const int w = tex.width, h = (int)tex.vMaxIndex + 1;
	*width = w;
	*height = h;
	const size_t n = (size_t)w * h * 3;
	if (out && capacity >= n) std::memcpy(out, tex.data, n);
	return (long long)n;
}

// Skybox::load (src/Skybox.cpp:4). Returns 1 when all six faces loaded.
REF_API int ref_load_skybox(void* h, float brightness, const char* front, const char* back, const char* left,
							const char* right, const char* top, const char* bottom) {
	auto* t = (RayTracer*)h;
	t->skybox.load(brightness, front, back, left, right, top, bottom);
	return t->skybox.hasSkybox() ? 1 : 0;
}

// Texture::sampleTexture (src/Texture.cpp:41)
REF_API void ref_sample_texture(void* h, int index, float u, float v, float* out) {
	auto* t = (RayTracer*)h;
	Eigen::Vector4f c = t->texturesArray[index].sampleTexture(Eigen::Vector2f(u, v));
	std::memcpy(out, c.data(), 16);
}

// Skybox::sampleBackground (src/Skybox.cpp:32)
REF_API void ref_sample_sky(void* h, const float* dir, float* out) {
	auto* t = (RayTracer*)h;
	float o[3] = {0, 0, 0};
	Eigen::Vector4f c = t->skybox.sampleBackground(mkray(o, dir));
	std::memcpy(out, c.data(), 16);
}

// BVH::buildTree (src/BVH.cpp:51). Returns the node count.
REF_API int ref_build_bvh(void* h) {
	auto* t = (RayTracer*)h;
	t->bvh.linearTree.clear();
	t->bvh.buildTree(t->trianglesArray);
	return (int)t->bvh.linearTree.size();
}

REF_API void ref_get_nodes(void* h, void* out) {
	auto* t = (RayTracer*)h;
	std::memcpy(out, (const void*)t->bvh.linearTree.data(), sizeof(LinearNode) * t->bvh.linearTree.size());
}

// AABB::hit (src/BVH.cpp:14) on an explicit box.
REF_API int ref_aabb_hit(const float* bmin, const float* bmax, const float* o, const float* d) {
	AABB box(v4(bmin), v4(bmax));
	return box.hit(mkray(o, d)) ? 1 : 0;
}

// BVH::hit (src/BVH.cpp:162): candidate list in the reference's order. Returns the count
// (which may exceed cap; only cap entries are written).
REF_API int ref_bvh_hit(void* h, const float* o, const float* d, int* out, int cap) {
	auto* t = (RayTracer*)h;
	const auto& list = t->bvh.hit(mkray(o, d));
	int n = (int)list.size();
	for (int i = 0; i < n && i < cap; ++i) out[i] = list[i];
	return n;
}

// Triangle::hit (src/Triangle.cpp:23) for n rays against triangle `tri`. out[n][4].
REF_API void ref_tri_hit(void* h, int tri, int n, const float* o, const float* d, float* out) {
	auto* t = (RayTracer*)h;
	for (int i = 0; i < n; ++i) {
		Eigen::Vector4f r = t->trianglesArray[tri].hit(mkray(o + 3 * i, d + 3 * i));
		std::memcpy(out + 4 * i, r.data(), 16);
	}
}

// Closest-hit selection exactly as the head of RayTracer::color does it
// (src/RayTracer.cpp:137-151), for n rays. ids[n], tab[n][3] = t, alpha, beta.
REF_API void ref_closest_hits(void* h, int n, const float* o, const float* d, int* ids, float* tab) {
	auto* t = (RayTracer*)h;
	for (int i = 0; i < n; ++i) {
		Ray r = mkray(o + 3 * i, d + 3 * i);
		const auto& list = t->bvh.hit(r);
		float best = FLT_MAX, alpha = 0.0f, beta = 0.0f;
		int index = -1;
		for (size_t k = 0; k < list.size(); ++k) {
			Eigen::Vector4f hc = t->trianglesArray[list[k]].hit(r);
			if (hc(2) < best) {
				index = list[k];
				best = hc(2);
				alpha = hc(0);
				beta = hc(1);
			}
		}
		ids[i] = index;
		tab[3 * i + 0] = best;
		tab[3 * i + 1] = alpha;
		tab[3 * i + 2] = beta;
	}
}

// Triangle::diffuse / specular (src/Triangle.cpp:95,121) on a fresh thread. out[num][4].
REF_API void ref_scatter(void* h, int tri, int lobe, const float* normal, const float* o, const float* d, int num,
						 float* out) {
	auto* t = (RayTracer*)h;
	std::thread th([&] {
		Ray r = mkray(o, d);
		Eigen::Vector4f n = v4(normal);
		const Triangle& T = t->trianglesArray[tri];
		std::vector<Eigen::Vector4f> res = lobe == 0 ? T.specular(n, r, num) : T.diffuse(n, r, num);
		for (int i = 0; i < num; ++i) std::memcpy(out + 4 * i, res[i].data(), 16);
	});
	th.join();
}

// Triangle::refract (src/Triangle.cpp:149). out[5] = proportion, direction xyzw.
REF_API void ref_refract(void* h, int tri, const float* normal, const float* o, const float* d, float* out) {
	auto* t = (RayTracer*)h;
	auto res = t->trianglesArray[tri].refract(v4(normal), mkray(o, d));
	out[0] = res.first;
	std::memcpy(out + 1, res.second.data(), 16);
}

// RayTracer::color (src/RayTracer.cpp:136) for n rays at recursion depth `depth`.
// fresh_each != 0: every ray on its own fresh thread (generators restart for each ray);
// otherwise all rays on one fresh thread, consuming the generators in sequence.
REF_API void ref_color(void* h, int depth, int n, const float* o, const float* d, float* out, int fresh_each) {
	auto* t = (RayTracer*)h;
	auto work = [&](int i0, int i1) {
		for (int i = i0; i < i1; ++i) {
			Eigen::Vector4f c = t->color(depth, mkray(o + 3 * i, d + 3 * i));
			std::memcpy(out + 4 * i, c.data(), 16);
		}
	};
	if (fresh_each) {
		for (int i = 0; i < n; ++i) {
			std::thread th(work, i, i + 1);
			th.join();
		}
	}
	else {
		std::thread th(work, 0, n);
		th.join();
	}
}

// One frame of the loop body of RayTracer::render (src/RayTracer.cpp:237-259) over rows
// [row0,row1), driven by std::thread workers in place of tbb::parallel_for. Rows are handed
// out through an atomic counter. accum is H*W*4 floats (in/out); out8 (H*W*3, may be null)
// receives the tone-mapped bytes. frame is the 1-based frame number `i` of the reference.
// Returns the wall seconds of the parallel region.
REF_API double ref_render_rows(void* h, int frame, int row0, int row1, float* accum, unsigned char* out8,
							   int nthreads) {
	auto* t = (RayTracer*)h;
	const int width = t->width;
	std::atomic<int> next(row0);
	auto worker = [&] {
		for (;;) {
			int row = next.fetch_add(1);
			if (row >= row1) break;
			for (int col = 0; col < width; ++col) {
				const auto& ray = t->camera.getRay(col, row);
				Eigen::Vector4f temp = Eigen::Vector4f::Zero();
				for (const auto& r : ray) temp += t->color(0, r);
				float* px = accum + ((size_t)row * width + col) * 4;
				Eigen::Map<Eigen::Vector4f> acc(px);
				acc += temp * 0.25f;
				if (out8) {
					for (int k = 0; k < 3; ++k) {
						float averaged = px[k] / frame;
						float gammaCorrected = powf(averaged, 1.0f / 2.2f);
						int clipNum = lroundf(gammaCorrected * 255.0f);
						if (clipNum > 255) clipNum = 255;
						if (clipNum < 0) clipNum = 0;
						out8[((size_t)row * width + col) * 3 + k] = (unsigned char)clipNum;
					}
				}
			}
		}
	};
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> pool;
	if (nthreads < 1) nthreads = 1;
	for (int i = 0; i < nthreads; ++i) pool.emplace_back(worker);
	for (auto& th : pool) th.join();
	auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}
