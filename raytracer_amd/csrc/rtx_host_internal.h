// Internal declarations shared by the host-side sources of librtx_host.so.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rtxh {

struct Vec4 {
	float x, y, z, w;
};

// class Triangle, /root/reference/include/RayTracer/Triangle.h:6-34 (176 bytes)
struct TriangleRec {
	Vec4 v[3];
	Vec4 n[3];
	Vec4 pn;
	Vec4 color;
	float uv[3][2];
	float rough, ior;
	int32_t tex;
	uint8_t emit, metal, transp, pad0;
	uint8_t pad1[8];
};
static_assert(sizeof(TriangleRec) == 176, "triangle record");

// struct LinearNode, /root/reference/include/RayTracer/BVH.h:35-43 (48 bytes)
struct NodeRec {
	Vec4 bmin, bmax;
	int32_t left, right, tri, pad;
};
static_assert(sizeof(NodeRec) == 48, "node record");

extern thread_local std::string g_err;
int fail(int code, const std::string& msg);

float dot(const Vec4& a, const Vec4& b);
Vec4 sub(const Vec4& a, const Vec4& b);
Vec4 scale(const Vec4& a, float s);
Vec4 divide(const Vec4& a, float s);
Vec4 normalized(const Vec4& a);
Vec4 cross3(const Vec4& a, const Vec4& b);

void camera(const Vec4& origin, const Vec4& viewPoint, float focal, float rotateDeg, int width, int height, Vec4 out[4]);
void make_triangle(const Vec4& v0, const Vec4& v1, const Vec4& v2, const Vec4& side, const Vec4& color, bool metal,
				   bool emit, bool transp, float rough, float ior, TriangleRec& T);
int build_bvh(const TriangleRec* tris, uint32_t n, std::vector<NodeRec>& out, int nthreads);
void tables(float* cos512, float* sin512, float* gamma256);
void tonemap(const float* accum, int frame, size_t npix, uint8_t* rgb8);
bool tonemap_table(uint32_t* bits, uint8_t* values, int cap, int* n_out, uint8_t special[2]);

}  // namespace rtxh
