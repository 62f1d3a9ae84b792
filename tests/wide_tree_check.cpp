// Host-side check of raytracer_amd/csrc/rtx_wide.h (header only, no GPU): pairing of triangles into two-triangle leaves,
// the surface-area tree over the leaves and its four-wide quantised collapse.  Built and run by tests/test_host.py.
// Prints "ok" and returns 0, or says what is wrong and returns 1.
#include "../raytracer_amd/csrc/rtx_wide.h"

#include <cstdio>
#include <random>
#include <set>

static int fail(const char* what, long a = 0, long b = 0) {
	std::printf("FAILED: %s (%ld, %ld)\n", what, a, b);
	return 1;
}

static void boxes_of(const std::vector<float>& verts, std::vector<float>& box) {
	const size_t n = verts.size() / 9;
	box.resize(n * 6);
	for (size_t t = 0; t < n; ++t)
		for (int c = 0; c < 3; ++c) {
			float lo = FLT_MAX, hi = -FLT_MAX;
			for (int k = 0; k < 3; ++k) {
				lo = std::min(lo, verts[t * 9 + k * 3 + c]);
				hi = std::max(hi, verts[t * 9 + k * 3 + c]);
			}
			box[t * 6 + c] = lo;
			box[t * 6 + 3 + c] = hi;
		}
}

// mates are mutual, share two vertices bit for bit, and their common box is at most `growth` times the larger own box
static int check_mates(const std::vector<float>& verts, const std::vector<float>& box, const std::vector<int>& mate, double growth) {
	const int n = (int)mate.size();
	for (int i = 0; i < n; ++i) {
		const int j = mate[i];
		if (j < 0) continue;
		if (j >= n || j == i || mate[j] != i) return fail("mates are not mutual", i, j);
		int shared = 0;
		for (int a = 0; a < 3; ++a)
			for (int b = 0; b < 3; ++b)
				if (!std::memcmp(&verts[(size_t)i * 9 + a * 3], &verts[(size_t)j * 9 + b * 3], 12)) ++shared;
		if (shared < 2) return fail("mates do not share an edge", i, j);
		float mn[3], mx[3];
		for (int c = 0; c < 3; ++c) {
			mn[c] = std::min(box[(size_t)i * 6 + c], box[(size_t)j * 6 + c]);
			mx[c] = std::max(box[(size_t)i * 6 + 3 + c], box[(size_t)j * 6 + 3 + c]);
		}
		const double own = std::max(rtxwide::half_area(&box[(size_t)i * 6], &box[(size_t)i * 6 + 3]), rtxwide::half_area(&box[(size_t)j * 6], &box[(size_t)j * 6 + 3]));
		if (rtxwide::half_area(mn, mx) > growth * own * 1.0001) return fail("common box too large", i, j);
	}
	return 0;
}

// leaves, tree and collapse as rtx_scene_set_bvh makes them; then every record must be reachable exactly once through the
// wide nodes, every child box must lie inside the decoded (quantised) box of its slot, and no node may be visited twice
static int check_tree(const std::vector<float>& box, const std::vector<int>& mate, int threads) {
	const int n = (int)mate.size();
	std::vector<float> primBox;
	std::vector<int> leafValue, recTri;
	for (int i = 0; i < n; ++i) {
		const int m = mate[i];
		if (m >= 0 && m < i) continue;
		const int rec = (int)recTri.size();
		float b6[6];
		std::memcpy(b6, &box[(size_t)i * 6], sizeof(b6));
		recTri.push_back(i);
		if (m >= 0) {
			for (int c = 0; c < 3; ++c) {
				b6[c] = std::min(b6[c], box[(size_t)m * 6 + c]);
				b6[3 + c] = std::max(b6[3 + c], box[(size_t)m * 6 + 3 + c]);
			}
			recTri.push_back(m);
		}
		primBox.insert(primBox.end(), b6, b6 + 6);
		leafValue.push_back(rec * 2 + (m >= 0 ? 1 : 0));
	}
	if ((int)recTri.size() != n) return fail("records do not cover the triangles", (long)recTri.size(), n);
	std::vector<rtxwide::BinNode> tree;
	rtxwide::build_sah(primBox, (int)leafValue.size(), tree, threads);
	std::vector<float> wide;
	int stackNeed = 0;
	if (!rtxwide::collapse_wide(tree, leafValue, wide, stackNeed, threads)) return fail("collapse_wide refused the tree");
	const int nw = (int)(wide.size() / 16);
	std::vector<char> seenNode(nw, 0), seenRec(n, 0);
	std::vector<int> st{0};
	struct Bound { float lo[3], hi[3]; };
	std::vector<Bound> bound(nw);
	for (int c = 0; c < 3; ++c) { bound[0].lo[c] = -FLT_MAX; bound[0].hi[c] = FLT_MAX; }
	while (!st.empty()) {
		const int w = st.back();
		st.pop_back();
		if (w < 0 || w >= nw) return fail("node index out of range", w, nw);
		if (seenNode[w]) return fail("wide node reachable twice", w);
		seenNode[w] = 1;
		const float* W = &wide[(size_t)w * 16];
		const float step[3] = {W[3], W[14], W[15]};
		uint32_t q[6];
		std::memcpy(q, W + 4, 24);
		int32_t code[4];
		std::memcpy(code, W + 10, 16);
		for (int k = 0; k < 4; ++k) {
			float lo[3], hi[3];
			for (int a = 0; a < 3; ++a) {
				lo[a] = W[a] + (float)((q[a * 2] >> (8 * k)) & 255u) * step[a];
				hi[a] = W[a] + (float)((q[a * 2 + 1] >> (8 * k)) & 255u) * step[a];
			}
			if (code[k] == 0) continue;
			if (code[k] > 0) {
				if (code[k] <= w) return fail("child numbered at or below its parent", w, code[k]);
				for (int a = 0; a < 3; ++a) { bound[code[k]].lo[a] = lo[a]; bound[code[k]].hi[a] = hi[a]; }
				st.push_back(code[k]);
				continue;
			}
			const int v = ~code[k], rec = v >> 1, cnt = 1 + (v & 1);
			for (int r = rec; r < rec + cnt; ++r) {
				if (r < 0 || r >= n) return fail("leaf record out of range", r, n);
				if (seenRec[r]) return fail("record in two leaves", r);
				seenRec[r] = 1;
				const int t = recTri[r];
				for (int a = 0; a < 3; ++a)
					if (box[(size_t)t * 6 + a] < lo[a] || box[(size_t)t * 6 + 3 + a] > hi[a]) return fail("triangle sticks out of its quantised leaf box", t, a);
			}
		}
		// a node's decoded child boxes may not stick out of what its parent decoded for it by more than rounding
		// (checked through the triangles above: every triangle lies inside the box of every slot on its path)
	}
	for (int r = 0; r < n; ++r)
		if (!seenRec[r]) return fail("record not reachable", r);
	if (stackNeed < 1 || stackNeed > 3 * 40) return fail("implausible stack need", stackNeed);
	return 0;
}

int main() {
	// 1. a displaced grid in the scene generator's order (all lower triangles, then all upper ones): every quad pairs up
	{
		const int nu = 37, nv = 23, nq = nu * nv;
		std::vector<float> P((size_t)(nu + 1) * (nv + 1) * 3);
		for (int i = 0; i <= nu; ++i)
			for (int j = 0; j <= nv; ++j) {
				float* p = &P[((size_t)i * (nv + 1) + j) * 3];
				p[0] = i * 0.37f; p[1] = 0.15f * std::sin(0.4f * i) * std::sin(0.4f * j); p[2] = j * 0.41f;
			}
		std::vector<float> verts((size_t)2 * nq * 9), box;
		auto put = [&](int t, int a, int b, int c) { const int id[3] = {a, b, c}; for (int k = 0; k < 3; ++k) std::memcpy(&verts[(size_t)t * 9 + k * 3], &P[(size_t)id[k] * 3], 12); };
		for (int i = 0; i < nu; ++i)
			for (int j = 0; j < nv; ++j) {
				const int a = i * (nv + 1) + j, b = a + nv + 1, c = b + 1, d = a + 1;
				put(i * nv + j, a, b, c);
				put(nq + i * nv + j, a, c, d);
			}
		boxes_of(verts, box);
		for (int threads : {1, 4}) {
			std::vector<int> mate;
			rtxwide::pair_triangles(verts.data(), box, 2 * nq, mate, threads);
			if (check_mates(verts, box, mate, 1.3)) return 1;
			for (int t = 0; t < nq; ++t)
				if (mate[t] != t + nq) return fail("grid quad not paired with its other half", t, mate[t]);
			if (check_tree(box, mate, threads)) return 1;
		}
	}
	// 2. a soup of unconnected triangles: nothing pairs, the tree still holds everything; duplicates and a single triangle
	{
		std::mt19937 rng(5);
		std::uniform_real_distribution<float> U(-10.0f, 10.0f), S(-0.5f, 0.5f);
		for (int n : {1, 2, 3, 5, 64, 1000}) {
			std::vector<float> verts((size_t)n * 9), box;
			for (int t = 0; t < n; ++t) {
				const float c[3] = {U(rng), U(rng), U(rng)};
				for (int k = 0; k < 9; ++k) verts[(size_t)t * 9 + k] = c[k % 3] + S(rng);
			}
			if (n >= 5) std::memcpy(&verts[9], &verts[0], 36);  // an exact duplicate: shares all three edges
			boxes_of(verts, box);
			std::vector<int> mate;
			rtxwide::pair_triangles(verts.data(), box, n, mate, 2);
			if (check_mates(verts, box, mate, 1.3)) return 1;
			int paired = 0;
			for (int m : mate) paired += m >= 0;
			if (n >= 5 ? paired != 2 : paired != 0) return fail("unexpected number of paired triangles in a soup", n, paired);
			if (check_tree(box, mate, 2)) return 1;
		}
	}
	// 3. a fan around one vertex with very different triangle sizes: pairs only where the common box stays small
	{
		std::vector<float> verts, box;
		const float r[6] = {1.0f, 1.0f, 8.0f, 8.0f, 1.0f, 1.0f};
		for (int k = 0; k < 5; ++k) {
			const float a0 = 0.9f * k, a1 = 0.9f * (k + 1);
			const float tri[9] = {0, 0, 0, r[k] * std::cos(a0), 0.0f, r[k] * std::sin(a0), r[k + 1] * std::cos(a1), 0.0f, r[k + 1] * std::sin(a1)};
			verts.insert(verts.end(), tri, tri + 9);
		}
		boxes_of(verts, box);
		std::vector<int> mate;
		rtxwide::pair_triangles(verts.data(), box, 5, mate, 1);
		if (check_mates(verts, box, mate, 1.3)) return 1;
		if (check_tree(box, mate, 1)) return 1;
	}
	// 4. no triangles at all, and trees whose interior nodes lack a child (as a caller's tree may): the collapse must end
	{
		std::vector<float> box;
		std::vector<int> mate;
		if (check_tree(box, mate, 1)) return 1;
		using rtxwide::BinNode;
		std::vector<BinNode> t(4);
		for (BinNode& b : t) { for (int c = 0; c < 3; ++c) { b.bmin[c] = 0.0f; b.bmax[c] = 1.0f; } }
		t[0].left = 1;               // root with a left child only
		t[1].left = 2; t[1].right = 3;
		t[2].tri = 0;
		t[3].left = -1; t[3].right = -1;  // an interior node without children: not usable
		std::vector<int> leafValue{0};
		std::vector<float> wide;
		int need = 0;
		if (!rtxwide::collapse_wide(t, leafValue, wide, need, 1)) return fail("collapse_wide refused a lopsided tree");
		if (wide.size() != 16) return fail("lopsided tree: expected one wide node", (long)wide.size() / 16);
		int32_t code[4];
		std::memcpy(code, &wide[10], 16);
		if (code[0] != ~0 || code[1] || code[2] || code[3]) return fail("lopsided tree: expected the one leaf in slot 0", code[0], code[1]);
	}
	std::printf("ok\n");
	return 0;
}
