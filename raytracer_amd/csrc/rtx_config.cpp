// The drop-in front end: RayTracer::parseConfigFile's grammar and error strings
// (src/RayTracer.cpp:284-518), loadModel's triangle construction (src/RayTracer.cpp:20-106) over a
// small OBJ/PLY reader (the reference imports through assimp, which is not available here), and
// the frame loop of RayTracer::render (src/RayTracer.cpp:228-274) with the parallel_for body
// replaced by calls into librtx_hip.so through the C ABI of include/rtx.h.
//
// librtx_hip.so is loaded at run time from the directory of this library, so that
// librtx_host.so itself never needs a GPU (the scene-preparation entry points work anywhere).
// Several GPUs: one scene per device, image tiles interleaved (tile t -> device t % n), one host
// thread per device per frame, accumulators merged tile by tile on the host before the tone map.

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <fstream>
#include <iostream>
#include <map>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_host_internal.h"
#include "rtx_image.h"

#define RTXH_API extern "C" __attribute__((visibility("default")))

namespace rtxh {

namespace {

// ---- mesh files -------------------------------------------------------------------------------

struct Mesh {
	std::vector<float> pos;      // xyz per vertex
	std::vector<float> nrm;      // xyz per vertex (empty: none in the file)
	std::vector<float> uv;       // uv per vertex (empty: none)
	std::vector<uint32_t> idx;   // 3 per triangle, indices into the unified vertex list
	float diffuse[3] = {0.6f, 0.6f, 0.6f};  // assimp's default material colour when the file names none
};

std::string dir_of(const std::string& path) {
	size_t p = path.find_last_of("/\\");
	return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

bool ends_with(const std::string& s, const char* suf) {
	size_t n = std::strlen(suf);
	if (s.size() < n) return false;
	for (size_t i = 0; i < n; ++i)
		if (std::tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
	return true;
}

// Wavefront OBJ: v / vn / vt / f (polygons fanned, negative indices allowed), Kd of the first
// material of an accompanying .mtl.  Position/normal/uv index triples are unified into one
// vertex list, which is what an importer hands the reference (mesh->mVertices[index] etc.).
bool load_obj(const std::string& path, Mesh& m) {
	std::ifstream in(path);
	if (!in) return false;
	std::vector<float> P, N, T;
	std::map<std::tuple<int, int, int>, uint32_t> remap;
	bool anyN = false, anyT = false, allN = true, allT = true;
	std::string line, mtl;
	struct Corner {
		int p, t, n;
	};
	std::vector<Corner> corners;
	std::vector<std::array<uint32_t, 3>> faces;
	auto vertex = [&](const Corner& c) {
		auto key = std::make_tuple(c.p, c.t, c.n);
		auto it = remap.find(key);
		if (it != remap.end()) return it->second;
		uint32_t id = (uint32_t)(m.pos.size() / 3);
		for (int k = 0; k < 3; ++k) m.pos.push_back(P[(size_t)c.p * 3 + k]);
		for (int k = 0; k < 3; ++k) m.nrm.push_back(c.n >= 0 ? N[(size_t)c.n * 3 + k] : 0.0f);
		for (int k = 0; k < 2; ++k) m.uv.push_back(c.t >= 0 ? T[(size_t)c.t * 2 + k] : 0.0f);
		remap.emplace(key, id);
		return id;
	};
	while (std::getline(in, line)) {
		std::istringstream ls(line);
		std::string tag;
		if (!(ls >> tag)) continue;
		if (tag == "v") {
			float x, y, z;
			if (ls >> x >> y >> z) P.insert(P.end(), {x, y, z});
		}
		else if (tag == "vn") {
			float x, y, z;
			if (ls >> x >> y >> z) N.insert(N.end(), {x, y, z});
		}
		else if (tag == "vt") {
			float u, v = 0.0f;
			if (ls >> u) {
				ls >> v;
				T.insert(T.end(), {u, v});
			}
		}
		else if (tag == "mtllib") ls >> mtl;
		else if (tag == "f") {
			corners.clear();
			std::string tok;
			while (ls >> tok) {
				Corner c{-1, -1, -1};
				int vals[3] = {0, 0, 0};
				int field = 0;
				size_t s = 0;
				while (field < 3 && s <= tok.size()) {
					size_t e = tok.find('/', s);
					if (e == std::string::npos) e = tok.size();
					if (e > s) vals[field] = std::atoi(tok.substr(s, e - s).c_str());
					++field;
					s = e + 1;
				}
				auto fix = [](int v, size_t count) { return v > 0 ? v - 1 : (v < 0 ? (int)count + v : -1); };
				c.p = fix(vals[0], P.size() / 3);
				c.t = fix(vals[1], T.size() / 2);
				c.n = fix(vals[2], N.size() / 3);
				if (c.p < 0 || (size_t)c.p >= P.size() / 3) return false;
				if (c.t >= 0 && (size_t)c.t >= T.size() / 2) c.t = -1;
				if (c.n >= 0 && (size_t)c.n >= N.size() / 3) c.n = -1;
				corners.push_back(c);
			}
			for (size_t k = 2; k < corners.size(); ++k) {
				for (const Corner* c : {&corners[0], &corners[k - 1], &corners[k]}) {
					anyN |= c->n >= 0;
					anyT |= c->t >= 0;
					allN &= c->n >= 0;
					allT &= c->t >= 0;
				}
				faces.push_back({vertex(corners[0]), vertex(corners[k - 1]), vertex(corners[k])});
			}
		}
	}
	for (auto& f : faces) m.idx.insert(m.idx.end(), f.begin(), f.end());
	if (!(anyN && allN)) m.nrm.clear();
	if (!(anyT && allT)) m.uv.clear();
	if (!mtl.empty()) {
		std::ifstream mi(dir_of(path) + mtl);
		std::string l2;
		while (std::getline(mi, l2)) {
			std::istringstream ls(l2);
			std::string tag;
			if ((ls >> tag) && tag == "Kd") {
				ls >> m.diffuse[0] >> m.diffuse[1] >> m.diffuse[2];
				break;
			}
		}
	}
	return !m.idx.empty();
}

// Stanford PLY, ascii or binary_little_endian: vertex x y z [nx ny nz] [s t | u v], face lists fanned.
bool load_ply(const std::string& path, Mesh& m) {
	std::ifstream in(path, std::ios::binary);
	if (!in) return false;
	std::string line;
	if (!std::getline(in, line) || line.substr(0, 3) != "ply") return false;
	bool binary = false;
	struct Prop {
		std::string type, name, listCount, listItem;
		bool isList = false;
	};
	struct Elem {
		std::string name;
		size_t count = 0;
		std::vector<Prop> props;
	};
	std::vector<Elem> elems;
	while (std::getline(in, line)) {
		if (!line.empty() && line.back() == '\r') line.pop_back();
		std::istringstream ls(line);
		std::string tag;
		ls >> tag;
		if (tag == "format") {
			std::string f;
			ls >> f;
			if (f == "binary_little_endian") binary = true;
			else if (f != "ascii") return false;
		}
		else if (tag == "element") {
			Elem e;
			ls >> e.name >> e.count;
			elems.push_back(e);
		}
		else if (tag == "property" && !elems.empty()) {
			Prop p;
			ls >> p.type;
			if (p.type == "list") {
				p.isList = true;
				ls >> p.listCount >> p.listItem >> p.name;
			}
			else
				ls >> p.name;
			elems.back().props.push_back(p);
		}
		else if (tag == "end_header") break;
	}
	auto size_of = [](const std::string& t) -> int {
		if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
		if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
		if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
		if (t == "double" || t == "float64") return 8;
		return 0;
	};
	auto read_num = [&](const std::string& t, double& out) -> bool {
		if (!binary) return (bool)(in >> out);
		unsigned char b[8];
		int n = size_of(t);
		if (!n || !in.read((char*)b, n)) return false;
		if (t == "float" || t == "float32") {
			float f;
			std::memcpy(&f, b, 4);
			out = f;
		}
		else if (t == "double" || t == "float64") std::memcpy(&out, b, 8);
		else if (t == "char" || t == "int8") out = (int8_t)b[0];
		else if (t == "uchar" || t == "uint8") out = b[0];
		else if (t == "short" || t == "int16") {
			int16_t v;
			std::memcpy(&v, b, 2);
			out = v;
		}
		else if (t == "ushort" || t == "uint16") {
			uint16_t v;
			std::memcpy(&v, b, 2);
			out = v;
		}
		else if (t == "int" || t == "int32") {
			int32_t v;
			std::memcpy(&v, b, 4);
			out = v;
		}
		else {
			uint32_t v;
			std::memcpy(&v, b, 4);
			out = v;
		}
		return true;
	};
	bool haveN = false, haveT = false;
	for (const Elem& e : elems) {
		if (e.name == "vertex") {
			int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
			for (size_t k = 0; k < e.props.size(); ++k) {
				const std::string& n = e.props[k].name;
				if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
				else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
				else if (n == "s" || n == "u" || n == "texture_u") iu = (int)k;
				else if (n == "t" || n == "v" || n == "texture_v") iv = (int)k;
			}
			if (ix < 0 || iy < 0 || iz < 0) return false;
			haveN = inx >= 0 && iny >= 0 && inz >= 0;
			haveT = iu >= 0 && iv >= 0;
			std::vector<double> vals(e.props.size());
			for (size_t i = 0; i < e.count; ++i) {
				for (size_t k = 0; k < e.props.size(); ++k) {
					if (e.props[k].isList) return false;
					if (!read_num(e.props[k].type, vals[k])) return false;
				}
				m.pos.insert(m.pos.end(), {(float)vals[ix], (float)vals[iy], (float)vals[iz]});
				if (haveN) m.nrm.insert(m.nrm.end(), {(float)vals[inx], (float)vals[iny], (float)vals[inz]});
				if (haveT) m.uv.insert(m.uv.end(), {(float)vals[iu], (float)vals[iv]});
			}
		}
		else {
			for (size_t i = 0; i < e.count; ++i) {
				for (const Prop& p : e.props) {
					if (p.isList) {
						double cnt;
						if (!read_num(p.listCount, cnt)) return false;
						std::vector<uint32_t> poly((size_t)cnt);
						for (auto& q : poly) {
							double v;
							if (!read_num(p.listItem, v)) return false;
							q = (uint32_t)v;
						}
						if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index"))
							for (size_t k = 2; k < poly.size(); ++k) m.idx.insert(m.idx.end(), {poly[0], poly[k - 1], poly[k]});
					}
					else {
						double v;
						if (!read_num(p.type, v)) return false;
					}
				}
			}
		}
	}
	const size_t nv = m.pos.size() / 3;
	for (uint32_t i : m.idx)
		if (i >= nv) return false;
	return !m.idx.empty();
}

// STL, binary or ASCII: three corners per facet, nothing shared.  The facet normal of the file goes to the three
// corners; if any facet's is zero (many writers leave it so) the mesh counts as having none and every face gets its
// geometric normal, like any other file without normals.
bool load_stl(const std::string& path, Mesh& m) {
	std::ifstream in(path, std::ios::binary);
	if (!in) return false;
	std::vector<char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	if (buf.size() < 15) return false;
	bool zeroNormal = false;
	auto facet = [&](const float n[3], const float v[9]) {
		const uint32_t first = (uint32_t)(m.pos.size() / 3);
		m.pos.insert(m.pos.end(), v, v + 9);
		for (int k = 0; k < 3; ++k) m.nrm.insert(m.nrm.end(), n, n + 3);
		m.idx.insert(m.idx.end(), {first, first + 1, first + 2});
		zeroNormal |= n[0] == 0.0f && n[1] == 0.0f && n[2] == 0.0f;
	};
	// binary: 80-byte header, facet count, 50 bytes per facet; the size settles it even if the header says "solid"
	uint32_t count = 0;
	if (buf.size() >= 84) std::memcpy(&count, &buf[80], 4);
	if (buf.size() >= 84 && buf.size() == 84 + (size_t)count * 50) {
		for (uint32_t f = 0; f < count; ++f) {
			float rec[12];
			std::memcpy(rec, &buf[84 + (size_t)f * 50], 48);
			facet(rec, rec + 3);
		}
	}
	else {
		std::istringstream text(std::string(buf.begin(), buf.end()));
		std::string tok;
		if (!(text >> tok) || tok != "solid") return false;
		float n[3] = {0, 0, 0}, v[9];
		int corners = 0;
		while (text >> tok) {
			if (tok == "facet") {
				text >> tok;  // "normal"
				if (!(text >> n[0] >> n[1] >> n[2])) return false;
				corners = 0;
			}
			else if (tok == "vertex") {
				if (corners >= 3 || !(text >> v[corners * 3] >> v[corners * 3 + 1] >> v[corners * 3 + 2])) return false;
				++corners;
			}
			else if (tok == "endfacet") {
				if (corners != 3) return false;
				facet(n, v);
			}
		}
	}
	if (zeroNormal) m.nrm.clear();
	return !m.idx.empty();
}

// Object File Format: "OFF", counts, vertices, polygons (fanned); "#" starts a comment.  Variants with per-vertex
// extras (COFF, NOFF, STOFF) are not read.
bool load_off(const std::string& path, Mesh& m) {
	std::ifstream in(path);
	if (!in) return false;
	std::string text, line;
	while (std::getline(in, line)) {
		const size_t hash = line.find('#');
		text += line.substr(0, hash);
		text += '\n';
	}
	std::istringstream ts(text);
	std::string tag;
	if (!(ts >> tag) || tag != "OFF") return false;
	size_t nv = 0, nf = 0, ne = 0;
	if (!(ts >> nv >> nf >> ne)) return false;
	m.pos.resize(nv * 3);
	for (float& x : m.pos)
		if (!(ts >> x)) return false;
	for (size_t f = 0; f < nf; ++f) {
		size_t k = 0;
		if (!(ts >> k)) return false;
		std::vector<uint32_t> poly(k);
		for (uint32_t& q : poly)
			if (!(ts >> q) || q >= nv) return false;
		for (size_t c = 2; c < k; ++c) m.idx.insert(m.idx.end(), {poly[0], poly[c - 1], poly[c]});
		std::getline(ts, line);  // a colour may follow the indices
	}
	return !m.idx.empty();
}

bool load_mesh(const std::string& path, Mesh& m) {
	if (ends_with(path, ".ply")) return load_ply(path, m);
	if (ends_with(path, ".obj")) return load_obj(path, m);
	if (ends_with(path, ".stl")) return load_stl(path, m);
	if (ends_with(path, ".off")) return load_off(path, m);
	for (auto reader : {load_obj, load_ply, load_off, load_stl}) {
		m = Mesh();
		if (reader(path, m)) return true;
	}
	return false;
}

// The triangle records of one placed mesh (src/RayTracer.cpp:78-105), appended to `out`.
// Done in bulk over the mesh arrays rather than corner by corner: every vertex is placed once (position * scale +
// origin, w lane 0 * scale + origin.w, as the reference's Vector4f expression gives), the fields all triangles of
// the model share are written into one template record, and a face then is the template plus three gathers.
void mesh_triangles(const Mesh& mesh, const Vec4& origin, float scale, bool isMetal, bool isLightEmitting, bool isTransparent,
					float specularRoughness, float refIndex, int32_t textureId, const std::optional<Vec4>& color,
					std::vector<TriangleRec>& out) {
	TriangleRec proto;
	std::memset(&proto, 0, sizeof(proto));
	proto.color = color ? *color : Vec4{mesh.diffuse[0], mesh.diffuse[1], mesh.diffuse[2], 0.0f};
	proto.metal = isMetal;
	proto.emit = isLightEmitting;
	proto.transp = isTransparent;
	proto.rough = specularRoughness;
	proto.ior = refIndex;
	proto.tex = textureId;

	const size_t nVerts = mesh.pos.size() / 3;
	std::vector<Vec4> placed(nVerts);
	for (size_t v = 0; v < nVerts; ++v)
		placed[v] = Vec4{mesh.pos[3 * v] * scale + origin.x, mesh.pos[3 * v + 1] * scale + origin.y, mesh.pos[3 * v + 2] * scale + origin.z,
						 0.0f * scale + origin.w};

	const bool smooth = !mesh.nrm.empty();
	const size_t base = out.size(), nFaces = mesh.idx.size() / 3;
	out.resize(base + nFaces, proto);
	for (size_t f = 0; f < nFaces; ++f) {
		TriangleRec& t = out[base + f];
		const uint32_t* corner = &mesh.idx[3 * f];
		for (int k = 0; k < 3; ++k) t.v[k] = placed[corner[k]];
		t.pn = normalized(cross3(sub(t.v[1], t.v[0]), sub(t.v[2], t.v[0])));
		for (int k = 0; k < 3; ++k) {
			// a file without normals gets the face normal at every corner (what the importer's GenNormals step gives
			// a flat face; for a positive scale that is the plane normal)
			if (smooth) t.n[k] = Vec4{mesh.nrm[3 * (size_t)corner[k]], mesh.nrm[3 * (size_t)corner[k] + 1], mesh.nrm[3 * (size_t)corner[k] + 2], 0.0f};
			else t.n[k] = t.pn;
			if (textureId >= 0) {
				t.uv[k][0] = mesh.uv[2 * (size_t)corner[k]];
				t.uv[k][1] = mesh.uv[2 * (size_t)corner[k] + 1];
			}
		}
	}
}

// ---- the device library, bound at run time ---------------------------------------------------

struct HipApi {
	void* lib = nullptr;
#define RTX_FN(name) decltype(&::name) name = nullptr;
	RTX_FN(rtx_last_error)
	RTX_FN(rtx_device_count)
	RTX_FN(rtx_scene_create)
	RTX_FN(rtx_scene_destroy)
	RTX_FN(rtx_scene_set_params)
	RTX_FN(rtx_scene_set_camera)
	RTX_FN(rtx_scene_set_triangles)
	RTX_FN(rtx_scene_set_bvh)
	RTX_FN(rtx_scene_add_texture)
	RTX_FN(rtx_scene_set_skybox)
	RTX_FN(rtx_scene_set_tables)
	RTX_FN(rtx_scene_set_tiles)
	RTX_FN(rtx_render_frames)
	RTX_FN(rtx_accum_read)
	RTX_FN(rtx_scene_set_tonemap_table)
	RTX_FN(rtx_tonemap_read)
	RTX_FN(rtx_render_frames_tonemapped)
	RTX_FN(rtx_get_stats)
	RTX_FN(rtx_gather_tiles)
	RTX_FN(rtx_exchange_tiles)
#undef RTX_FN
};

void bind_hip(HipApi& api) {
	Dl_info info;
	std::string dir;
	if (dladdr((void*)&bind_hip, &info) && info.dli_fname) dir = dir_of(info.dli_fname);
	const std::string path = dir + "librtx_hip.so";
	api.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
	if (!api.lib) throw std::runtime_error(std::string("Can't load ") + path + ": " + dlerror());
#define RTX_FN(name)                                                                    \
	api.name = (decltype(api.name))dlsym(api.lib, #name);                                \
	if (!api.name) throw std::runtime_error("librtx_hip.so lacks " #name);
	RTX_FN(rtx_last_error)
	RTX_FN(rtx_device_count)
	RTX_FN(rtx_scene_create)
	RTX_FN(rtx_scene_destroy)
	RTX_FN(rtx_scene_set_params)
	RTX_FN(rtx_scene_set_camera)
	RTX_FN(rtx_scene_set_triangles)
	RTX_FN(rtx_scene_set_bvh)
	RTX_FN(rtx_scene_add_texture)
	RTX_FN(rtx_scene_set_skybox)
	RTX_FN(rtx_scene_set_tables)
	RTX_FN(rtx_scene_set_tiles)
	RTX_FN(rtx_render_frames)
	RTX_FN(rtx_accum_read)
	RTX_FN(rtx_scene_set_tonemap_table)
	RTX_FN(rtx_tonemap_read)
	RTX_FN(rtx_render_frames_tonemapped)
	RTX_FN(rtx_get_stats)
	RTX_FN(rtx_gather_tiles)
	RTX_FN(rtx_exchange_tiles)
#undef RTX_FN
}

// ---- the host mirror of class RayTracer ---------------------------------------------------------

constexpr int TILE = 64;  // tile edge used by rtx_scene_set_tiles

class HostRayTracer {
public:
	void parseConfigFile(const std::string& path, const std::string& outDir, int nDevices);

private:
	int width = 0, height = 0, renderNum = 0;
	int diffuseRayNum = 0, specularRayNum = 0, maxRecursionDepth = 0;
	Vec4 backgroundColor{0, 0, 0, 0};
	Vec4 cam[4];
	std::vector<TriangleRec> trianglesArray;
	std::vector<Image> texturesArray;
	bool hasSkybox = false;
	float skyBrightness = 1.0f;
	int skyTex[6] = {-1, -1, -1, -1, -1, -1};
	std::string baseDir;

	std::string resolve(const std::string& p) const {
		if (p.empty() || p[0] == '/') return p;
		std::ifstream direct(p);
		if (direct) return p;
		return baseDir + p;
	}
	void loadModel(const std::string& modelPath, const Vec4& origin, float scale, bool isMetal, bool isLightEmitting,
				   bool isTransparent, float specularRoughness, float refIndex, const std::optional<std::string>& texturePath,
				   const std::optional<Vec4>& color);
	void loadSkybox(float brightness, const std::string paths[6]);
	void render(const std::string& outDir, int nDevices);
};

// The model keyword of the config file (src/RayTracer.cpp:20-106): one mesh becomes triangle records.
void HostRayTracer::loadModel(const std::string& modelPath, const Vec4& origin, float scale, bool isMetal,
							  bool isLightEmitting, bool isTransparent, float specularRoughness, float refIndex,
							  const std::optional<std::string>& texturePath, const std::optional<Vec4>& color) {
	Mesh mesh;
	if (!load_mesh(resolve(modelPath), mesh)) throw std::runtime_error("Can't load model in " + modelPath);

	// texture: only for a mesh that has coordinates, and only if the file decodes; both failures are reported on
	// stdout in the reference's words and the model is kept, untextured
	int32_t textureId = -1;
	if (texturePath) {
		Image img;
		if (mesh.uv.empty()) {
			std::cout << "No texture for model in " << modelPath << std::endl;
		}
		else if (!load_texture_file(resolve(*texturePath), img)) {
			std::cout << "Can't load texture in " << *texturePath << std::endl;
			std::cout << "No texture for model in " << modelPath << std::endl;
		}
		else {
			texturesArray.push_back(std::move(img));
			textureId = (int32_t)texturesArray.size() - 1;
		}
	}

	mesh_triangles(mesh, origin, scale, isMetal, isLightEmitting, isTransparent, specularRoughness, refIndex, textureId, color, trianglesArray);
}

// src/Skybox.cpp:4-26: all six faces or none
void HostRayTracer::loadSkybox(float brightness, const std::string paths[6]) {
	std::vector<Image> faces(6);
	for (int i = 0; i < 6; ++i)
		if (!load_texture_file(resolve(paths[i]), faces[i])) {
			std::cout << "Can't load skybox\n";
			return;
		}
	skyBrightness = brightness;
	for (int i = 0; i < 6; ++i) {
		texturesArray.push_back(std::move(faces[i]));
		skyTex[i] = (int)texturesArray.size() - 1;
	}
	hasSkybox = true;
}

// src/RayTracer.cpp:284-518, same keys in the same order, same messages
void HostRayTracer::parseConfigFile(const std::string& path, const std::string& outDir, int nDevices) {
	std::ifstream config(path);
	baseDir = dir_of(path);
	std::string key;
	auto expect = [&](const char* name) {
		if (!(config >> key && key == name)) throw std::runtime_error(std::string("Can't parse \"") + name + "\"");
	};
	auto vec3 = [&]() {
		float x, y, z;
		config >> x >> y >> z;
		return Vec4{x, y, z, 0.0f};
	};
	expect("frame");
	config >> width >> height;
	expect("camera");
	{
		float x, y, z, atX, atY, atZ, focal, rotate;
		config >> x >> y >> z >> atX >> atY >> atZ >> focal >> rotate;
		if (width <= 0 || height <= 0) throw std::runtime_error("Can't parse \"frame\"");
		camera(Vec4{x, y, z, 0.0f}, Vec4{atX, atY, atZ, 0.0f}, focal, rotate, width, height, cam);
	}
	expect("background_color");
	backgroundColor = vec3();
	expect("max_recursion_depth");
	config >> maxRecursionDepth;
	if (maxRecursionDepth <= 0) throw std::runtime_error("Expect: \"max_recursion_depth\" > 0");
	expect("diffuse_ray_number");
	config >> diffuseRayNum;
	if (diffuseRayNum <= 0) throw std::runtime_error("Expect: \"diffuse_ray_number\" > 0");
	expect("specular_ray_number");
	config >> specularRayNum;
	if (specularRayNum <= 0) throw std::runtime_error("Expect: \"specular_ray_number\" > 0");

	bool haveRenderNum = false;
	while (config >> key) {
		if (key == "skybox") {
			float brightness;
			std::string p[6];
			config >> brightness >> p[0] >> p[1] >> p[2] >> p[3] >> p[4] >> p[5];
			loadSkybox(brightness, p);
		}
		else if (key == "model_start") {
			std::string modelPath, texturePath;
			expect("model_path");
			config >> modelPath;
			expect("texture_path");
			config >> texturePath;
			std::optional<std::string> tex = texturePath;
			if (texturePath == "no") tex.reset();
			float scale;
			expect("scale");
			config >> scale;
			expect("position_offset");
			Vec4 origin = vec3();
			bool isMetal, isLightEmitting, isTransparent;
			float specularRoughness, refIndex;
			expect("is_metal");
			config >> isMetal;
			expect("is_light_emitting");
			config >> isLightEmitting;
			expect("is_transparent");
			config >> isTransparent;
			expect("specular_roughness");
			config >> specularRoughness;
			expect("refractive_index");
			config >> refIndex;
			std::optional<Vec4> color;
			config >> key;
			if (key == "override_color") {
				color = vec3();
				config >> key;
			}
			if (key != "model_end") throw std::runtime_error("Can't parse \"model_end\"");
			loadModel(modelPath, origin, scale, isMetal, isLightEmitting, isTransparent, specularRoughness, refIndex, tex, color);
		}
		else if (key == "triangle_start") {
			expect("vertex_0");
			Vec4 v0 = vec3();
			expect("vertex_1");
			Vec4 v1 = vec3();
			expect("vertex_2");
			Vec4 v2 = vec3();
			expect("normal_side");
			Vec4 side = vec3();
			expect("color");
			Vec4 color = vec3();
			bool isMetal, isLightEmitting, isTransparent;
			float specularRoughness, refIndex;
			expect("is_metal");
			config >> isMetal;
			expect("is_light_emitting");
			config >> isLightEmitting;
			expect("is_transparent");
			config >> isTransparent;
			expect("specular_roughness");
			config >> specularRoughness;
			expect("refractive_index");
			config >> refIndex;
			expect("triangle_end");
			TriangleRec t;
			make_triangle(v0, v1, v2, side, color, isMetal, isLightEmitting, isTransparent, specularRoughness, refIndex, t);
			trianglesArray.push_back(t);
		}
		else if (key == "render_num") {
			config >> renderNum;
			haveRenderNum = true;
			break;
		}
		else
			throw std::runtime_error("Expect: \"skybox\" or \"model_start\" or \"triangle_start\" or \"render_num\"");
	}
	// the reference leaves renderNum uninitialised when the key is missing; here that is zero frames
	if (!haveRenderNum) renderNum = 0;
	config.close();
	render(outDir, nDevices);
}

// src/RayTracer.cpp:228-274
void HostRayTracer::render(const std::string& outDir, int nDevices) {
	HipApi hip;
	bind_hip(hip);
	auto chk = [&](int rc) {
		if (rc < 0) throw std::runtime_error(std::string("GPU path: ") + hip.rtx_last_error());
		return rc;
	};
	const int avail = chk(hip.rtx_device_count());
	if (nDevices < 1) nDevices = 1;
	// Which GPU each of the nDevices workers drives: 0, 1, 2, ... unless RTX_DEVICE_MAP lists them ("0,0" = two
	// workers sharing GPU 0, whose tiles then meet by a copy inside that GPU instead of over RCCL).
	std::vector<int> deviceOf(nDevices);
	for (int d = 0; d < nDevices; ++d) deviceOf[d] = d;
	if (const char* e = std::getenv("RTX_DEVICE_MAP")) {
		std::vector<int> ids;
		for (const char* p = e; *p;) {
			char* end = nullptr;
			const long v = std::strtol(p, &end, 10);
			if (end == p) break;
			ids.push_back((int)v);
			p = end;
			while (*p == ',' || *p == ' ') ++p;
		}
		if ((int)ids.size() != nDevices) throw std::runtime_error("GPU path: RTX_DEVICE_MAP lists " + std::to_string(ids.size()) + " devices for " + std::to_string(nDevices) + " workers");
		deviceOf = ids;
	}
	for (int d : deviceOf)
		if (d < 0 || d >= avail) throw std::runtime_error("GPU path: device " + std::to_string(d) + " requested, " + std::to_string(avail) + " present");
	// How the tiles of a frame are shared out among several workers.  static: worker d renders the tiles d, d + n, ...
	// of every frame.  dynamic (the default): the first part of the tiles is interleaved the same way, the rest
	// (RTX_DYNAMIC_PERCENT of them, default 25) is handed out in small batches from one atomic counter to whichever
	// worker comes first, so that a GPU that is slower on its share -- or busy with something else -- takes fewer.
	bool dynamicBalance = nDevices > 1;
	if (const char* e = std::getenv("RTX_BALANCE")) dynamicBalance = nDevices > 1 && std::string(e) != "static";
	int dynamicPercent = 25;
	if (const char* e = std::getenv("RTX_DYNAMIC_PERCENT")) dynamicPercent = std::max(1, std::min(100, std::atoi(e)));

	std::vector<NodeRec> nodes;
	unsigned hw = std::thread::hardware_concurrency();
	build_bvh(trianglesArray.data(), (uint32_t)trianglesArray.size(), nodes, (int)std::min<unsigned>(hw ? hw : 1, 16));
	std::vector<float> ct(512), st(512), gt(256);
	tables(ct.data(), st.data(), gt.data());

	struct Dev {
		rtx_scene* scene = nullptr;
	};
	std::vector<Dev> devs(nDevices);
	struct Cleanup {
		HipApi& hip;
		std::vector<Dev>& devs;
		~Cleanup() {
			for (auto& d : devs)
				if (d.scene) hip.rtx_scene_destroy(d.scene);
		}
	} cleanup{hip, devs};

	// The 8-bit conversion runs on the device from a table of this host's powf/lroundf (same bytes, a quarter
	// of the transfer, none of the 6 M powf calls per 1080p frame); should the table not be constructible
	// the accumulator is read back and converted here.
	std::vector<uint32_t> toneBits(1024);
	std::vector<uint8_t> toneVal(1024);
	int toneN = 0;
	uint8_t toneSpecial[2] = {0, 0};
	const bool deviceToneMap = tonemap_table(toneBits.data(), toneVal.data(), 1024, &toneN, toneSpecial);

	const float bg[3] = {backgroundColor.x, backgroundColor.y, backgroundColor.z};
	for (int d = 0; d < nDevices; ++d) {
		chk(hip.rtx_scene_create(&devs[d].scene, deviceOf[d]));
		rtx_scene* s = devs[d].scene;
		chk(hip.rtx_scene_set_params(s, width, height, maxRecursionDepth, diffuseRayNum, specularRayNum, bg, 0x5EEDull));
		chk(hip.rtx_scene_set_camera(s, &cam[0].x, &cam[1].x, &cam[2].x, &cam[3].x));
		for (const Image& t : texturesArray) chk(hip.rtx_scene_add_texture(s, t.rgb.data(), t.w, t.h));
		if (hasSkybox) chk(hip.rtx_scene_set_skybox(s, skyBrightness, skyTex));
		chk(hip.rtx_scene_set_triangles(s, trianglesArray.data(), (uint32_t)trianglesArray.size()));
		chk(hip.rtx_scene_set_bvh(s, nodes.data(), (uint32_t)nodes.size()));
		chk(hip.rtx_scene_set_tables(s, ct.data(), st.data(), gt.data()));
		chk(hip.rtx_scene_set_tiles(s, d, nDevices, -1));
		if (deviceToneMap && d == 0) chk(hip.rtx_scene_set_tonemap_table(s, toneBits.data(), toneVal.data(), toneN, toneSpecial[0], toneSpecial[1]));
	}

	// Images are encoded and written by a second thread while the next frame renders; the frame's line on
	// stdout follows its file, in frame order, as in the reference (src/RayTracer.cpp:263-271).
	struct Job {
		int frame;
		float seconds;
		std::vector<uint8_t> pixels;
	};
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Job> jobs;
	bool done = false;
	std::string writeError;
	std::thread writer([&] {
		for (;;) {
			Job job;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return done || !jobs.empty(); });
				if (jobs.empty()) return;
				job = std::move(jobs.front());
			}
			char name[32];
			std::snprintf(name, sizeof(name), "out_%03d.png", job.frame);
			const std::string file = outDir.empty() ? std::string(name) : (outDir + "/" + name);
			const bool ok = write_png(file, width, height, job.pixels.data());
			if (ok) std::cout << "Output frame " << job.frame << ", use " << job.seconds << "s\n";
			{
				std::lock_guard<std::mutex> lk(mu);
				if (!ok && writeError.empty()) writeError = "Can't write " + file;
				jobs.pop_front();  // only now: the renderer may run at most two frames ahead of the disk
			}
			cv.notify_all();
		}
	});
	struct JoinWriter {
		std::mutex& mu;
		std::condition_variable& cv;
		bool& done;
		std::thread& t;
		~JoinWriter() {
			{
				std::lock_guard<std::mutex> lk(mu);
				done = true;
			}
			cv.notify_all();
			if (t.joinable()) t.join();
		}
	} joinWriter{mu, cv, done, writer};

	// Every worker renders its tiles of frame i on a thread of its own; the tiles then travel to worker 0's GPU (one
	// message per peer, each over its own xGMI link, all at once) and worker 0 alone hands the image back: 8-bit from
	// its tone map, or the accumulator if the table could not be built.
	//   static   worker d keeps the accumulators of its share over the frames; rtx_gather_tiles copies them over.
	//   dynamic  a worker's image collects one frame's contributions, whichever tiles it happened to render;
	//            rtx_exchange_tiles adds them to worker 0's accumulator and leaves the worker's image empty again.
	std::vector<float> accumulateImg;
	if (!deviceToneMap) accumulateImg.assign((size_t)width * height * 4, 0.0f);
	std::vector<rtx_scene*> scenes(nDevices);
	for (int d = 0; d < nDevices; ++d) scenes[d] = devs[d].scene;
	const int numTiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
	const int headTiles = dynamicBalance ? (int)((long long)numTiles * (100 - dynamicPercent) / 100) / nDevices * nDevices : numTiles;
	const int batchTiles = std::max(1, (numTiles - headTiles) / (nDevices * 4));
	// One GPU rendering the whole image: the frames go to the device in groups (RTX_FRAMES_PER_BATCH, default 4), whose rays
	// share the queues of every recursion level -- a level's launch ends with its waves draining their last rays, about a
	// quarter of a millisecond whatever the queue holds, so four frames together cost about 4 % less than one after the
	// other -- while every frame still gets its own image: its contribution is added in frame order and tone-mapped
	// before the next one's is (rtx_render_frames_tonemapped).  The frame's line on stdout carries its share of the
	// group's time.
	int framesPerBatch = 1;
	if (nDevices == 1 && deviceToneMap && !std::getenv("RTX_GATHER_SELF")) {
		framesPerBatch = 4;
		if (const char* e = std::getenv("RTX_FRAMES_PER_BATCH")) framesPerBatch = std::max(1, std::min(64, std::atoi(e)));
	}
	const size_t queueRoom = (size_t)std::max(2, 2 * framesPerBatch);  // images the renderer may be ahead of the disk
	for (int i = 1; framesPerBatch > 1 && i <= renderNum; i += framesPerBatch) {
		const int nb = std::min(framesPerBatch, renderNum - i + 1);
		std::vector<std::vector<uint8_t>> images((size_t)nb, std::vector<uint8_t>((size_t)width * height * 3));
		std::vector<uint8_t*> ptrs((size_t)nb);
		for (int k = 0; k < nb; ++k) ptrs[(size_t)k] = images[(size_t)k].data();
		auto time1 = std::chrono::system_clock::now();
		chk(hip.rtx_render_frames_tonemapped(scenes[0], (uint32_t)i, (uint32_t)nb, ptrs.data()));
		auto time2 = std::chrono::system_clock::now();
		const float seconds = std::chrono::duration<float>(time2 - time1).count() / (float)nb;
		for (int k = 0; k < nb; ++k) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return jobs.size() < queueRoom || !writeError.empty(); });
				if (!writeError.empty()) throw std::runtime_error(writeError);
				jobs.push_back(Job{i + k, seconds, std::move(images[(size_t)k])});
			}
			cv.notify_all();
		}
	}
	for (int i = 1; framesPerBatch == 1 && i <= renderNum; ++i) {
		auto time1 = std::chrono::system_clock::now();
		std::vector<std::string> errors(nDevices);
		if (nDevices == 1) {
			if (hip.rtx_render_frames(scenes[0], (uint32_t)i, 1) < 0) errors[0] = hip.rtx_last_error();
		}
		else {
			std::atomic<int> nextTile(headTiles);  // the seam of tbb::parallel_for's own work stealing (src/RayTracer.cpp:237)
			std::vector<std::thread> workers;
			for (int d = 0; d < nDevices; ++d)
				workers.emplace_back([&, d] {
					// rtx_last_error is per thread: fetch it on the thread that made the call
					auto run = [&](int first, int stride, int count) {
						if (hip.rtx_scene_set_tiles(scenes[d], first, stride, count) < 0 || hip.rtx_render_frames(scenes[d], (uint32_t)i, 1) < 0) {
							errors[d] = hip.rtx_last_error();
							return false;
						}
						return true;
					};
					if (!dynamicBalance) {
						run(d, nDevices, -1);
						return;
					}
					if (headTiles > 0 && !run(d, nDevices, headTiles / nDevices)) return;
					for (;;) {
						const int first = nextTile.fetch_add(batchTiles);
						if (first >= numTiles) break;
						if (!run(first, 1, std::min(batchTiles, numTiles - first))) return;
					}
				});
			for (auto& w : workers) w.join();
		}
		for (const auto& e : errors)
			if (!e.empty()) throw std::runtime_error("GPU path: " + e);
		// RTX_GATHER_SELF=1 (tests on one GPU): the single share goes through the exchange as well, sent to itself
		if (dynamicBalance) chk(hip.rtx_exchange_tiles(scenes.data(), nDevices));
		else if (nDevices > 1 || std::getenv("RTX_GATHER_SELF")) chk(hip.rtx_gather_tiles(scenes.data(), nDevices));
		std::vector<uint8_t> outputBuffer((size_t)width * height * 3);
		if (deviceToneMap) chk(hip.rtx_tonemap_read(scenes[0], i, outputBuffer.data()));
		else {
			chk(hip.rtx_accum_read(scenes[0], accumulateImg.data()));
			tonemap(accumulateImg.data(), i, (size_t)width * height, outputBuffer.data());
		}
		auto time2 = std::chrono::system_clock::now();  // the reference's interval covers the tone map too (:235-261)
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return jobs.size() < 2 || !writeError.empty(); });
			if (!writeError.empty()) throw std::runtime_error(writeError);
			jobs.push_back(Job{i, std::chrono::duration<float>(time2 - time1).count(), std::move(outputBuffer)});
		}
		cv.notify_all();
	}
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return jobs.empty() || !writeError.empty(); });
		if (!writeError.empty()) throw std::runtime_error(writeError);
	}
	std::cout << "Render finished" << std::endl;
}

}  // namespace

}  // namespace rtxh

RTXH_API long long rtxh_load_model(const char* model_path, const float origin[3], float scale, int is_metal, int is_light_emitting,
								   int is_transparent, float specular_roughness, float refractive_index, int texture_index,
								   const float* override_color, void* triangles_out, size_t capacity, int* textured) {
	if (!model_path || !origin) return rtxh::fail(RTX_ERR_ARG, "null argument");
	rtxh::Mesh mesh;
	if (!rtxh::load_mesh(model_path, mesh)) return rtxh::fail(RTX_ERR_STATE, std::string("Can't load model in ") + model_path);
	const bool useTexture = texture_index >= 0 && !mesh.uv.empty();
	if (textured) *textured = useTexture ? 1 : 0;
	const size_t n = mesh.idx.size() / 3;
	if (!triangles_out) return (long long)n;
	if (capacity < n) return rtxh::fail(RTX_ERR_ARG, "triangle buffer too small");
	std::optional<rtxh::Vec4> color;
	if (override_color) color = rtxh::Vec4{override_color[0], override_color[1], override_color[2], 0.0f};
	std::vector<rtxh::TriangleRec> out;
	rtxh::mesh_triangles(mesh, rtxh::Vec4{origin[0], origin[1], origin[2], 0.0f}, scale, is_metal != 0, is_light_emitting != 0,
						 is_transparent != 0, specular_roughness, refractive_index, useTexture ? texture_index : -1, color, out);
	std::memcpy(triangles_out, out.data(), n * sizeof(rtxh::TriangleRec));
	return (long long)n;
}

RTXH_API int rtxh_run_config(const char* config_path, const char* out_dir, int n_devices) {
	if (!config_path) return rtxh::fail(RTX_ERR_ARG, "null argument");
	try {
		rtxh::HostRayTracer tracer;
		tracer.parseConfigFile(config_path, out_dir ? out_dir : "", n_devices);
	}
	catch (const std::exception& e) {
		// the reference's main prints what() and carries on (src/main.cpp:13-15); the message travels
		// back through rtxh_last_error so that the caller can do the same
		return rtxh::fail(RTX_ERR_STATE, e.what());
	}
	return RTX_OK;
}
