// Host-side construction of the 4-wide, 8-bit quantised tree that k_trace_wide walks.
//
// For a ray whose reciprocals are finite, the candidate list of the reference's BVH::hit
// (src/BVH.cpp:170-189) is exactly the set of triangles whose own bounding box passes AABB::hit, in the
// reference's leaf order.  The order is carried by each triangle's rank, the membership by the exact
// leaf test in the kernel; the interior of the tree only prunes.  The pruning tree therefore need not be
// the reference's (an object-median split, src/BVH.cpp:93-111): it is built here with a binned
// surface-area heuristic, which visits far fewer nodes per ray, and collapsed four wide.
#pragma once

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace rtxwide {

struct BinNode {
	float bmin[3], bmax[3];
	int left = -1, right = -1;  // children (index into the node array) or -1
	int tri = -1;               // >= 0: leaf of that triangle
};

struct Prim {
	float bmin[3], bmax[3], c[3];
	int tri;
};

inline double half_area(const float* mn, const float* mx) {
	double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
	return dx * dy + dy * dz + dz * dx;
}

inline void grow(float* mn, float* mx, const float* omn, const float* omx) {
	for (int a = 0; a < 3; ++a) {
		mn[a] = std::min(mn[a], omn[a]);
		mx[a] = std::max(mx[a], omx[a]);
	}
}

inline void reset(float* mn, float* mx) {
	for (int a = 0; a < 3; ++a) {
		mn[a] = FLT_MAX;
		mx[a] = -FLT_MAX;
	}
}

// Binned SAH split of prims[start,end) (8 to 64 bins per axis, one primitive per leaf).  Returns the split
// position; both sides are non-empty.
inline int sah_split(std::vector<Prim>& prims, int start, int end) {
	constexpr int MAXBINS = 64;
	const int BINS = std::min(MAXBINS, std::max(8, (end - start) / 2));  // the large nodes near the root gain from fine planes
	float cmin[3], cmax[3];
	reset(cmin, cmax);
	for (int i = start; i < end; ++i)
		for (int a = 0; a < 3; ++a) {
			cmin[a] = std::min(cmin[a], prims[i].c[a]);
			cmax[a] = std::max(cmax[a], prims[i].c[a]);
		}
	double bestCost = DBL_MAX;
	int bestAxis = -1, bestPlane = 0;
	for (int axis = 0; axis < 3; ++axis) {
		const float ext = cmax[axis] - cmin[axis];
		if (!(ext > 0.0f)) continue;
		const float scale = (float)BINS / ext;
		int cnt[MAXBINS] = {0};
		float bmn[MAXBINS][3], bmx[MAXBINS][3];
		for (int b = 0; b < BINS; ++b) reset(bmn[b], bmx[b]);
		for (int i = start; i < end; ++i) {
			int b = std::min(BINS - 1, std::max(0, (int)((prims[i].c[axis] - cmin[axis]) * scale)));
			cnt[b]++;
			grow(bmn[b], bmx[b], prims[i].bmin, prims[i].bmax);
		}
		double rightArea[MAXBINS];
		int rightCnt[MAXBINS];
		float rmn[3], rmx[3];
		reset(rmn, rmx);
		int rc = 0;
		for (int b = BINS - 1; b > 0; --b) {
			if (cnt[b]) grow(rmn, rmx, bmn[b], bmx[b]);
			rc += cnt[b];
			rightArea[b] = rc ? half_area(rmn, rmx) : 0.0;
			rightCnt[b] = rc;
		}
		float lmn[3], lmx[3];
		reset(lmn, lmx);
		int lc = 0;
		for (int b = 0; b < BINS - 1; ++b) {  // plane b: bins 0..b on the left
			if (cnt[b]) grow(lmn, lmx, bmn[b], bmx[b]);
			lc += cnt[b];
			if (lc == 0 || rightCnt[b + 1] == 0) continue;
			double cost = half_area(lmn, lmx) * lc + rightArea[b + 1] * rightCnt[b + 1];
			if (cost < bestCost) {
				bestCost = cost;
				bestAxis = axis;
				bestPlane = b;
			}
		}
	}
	if (bestAxis < 0) return (start + end) / 2;  // all centroids coincide
	const float scale = (float)BINS / (cmax[bestAxis] - cmin[bestAxis]);
	const float base = cmin[bestAxis];
	auto mid = std::partition(prims.begin() + start, prims.begin() + end, [&](const Prim& p) {
		int b = std::min(BINS - 1, std::max(0, (int)((p.c[bestAxis] - base) * scale)));
		return b <= bestPlane;
	});
	int split = (int)(mid - prims.begin());
	if (split == start || split == end) split = (start + end) / 2;  // cannot happen; keeps the recursion finite
	return split;
}

struct Range {
	int node, start, end;
};

// Subtree under pool[first.node]; ranges of at most `handoff` primitives are deferred (0: none).
inline void build_range(std::vector<BinNode>& pool, std::vector<Prim>& prims, Range first, int handoff,
						std::vector<Range>* deferred) {
	std::vector<Range> st;
	st.push_back(first);
	while (!st.empty()) {
		Range w = st.back();
		st.pop_back();
		const int count = w.end - w.start;
		if (count == 1) {
			pool[w.node].tri = prims[w.start].tri;
			continue;
		}
		if (deferred && count <= handoff && w.node != first.node) {
			deferred->push_back(w);
			continue;
		}
		const int split = sah_split(prims, w.start, w.end);
		BinNode l, r;
		reset(l.bmin, l.bmax);
		for (int i = w.start; i < split; ++i) grow(l.bmin, l.bmax, prims[i].bmin, prims[i].bmax);
		reset(r.bmin, r.bmax);
		for (int i = split; i < w.end; ++i) grow(r.bmin, r.bmax, prims[i].bmin, prims[i].bmax);
		pool.push_back(l);
		pool.push_back(r);
		const int li = (int)pool.size() - 2;
		pool[w.node].left = li;
		pool[w.node].right = li + 1;
		st.push_back(Range{li, w.start, split});
		st.push_back(Range{li + 1, split, w.end});
	}
}

// Binary SAH tree over triangle boxes (box[i*6 .. i*6+5] = min xyz, max xyz).  Node 0 is the root.
inline void build_sah(const std::vector<float>& box, int ntri, std::vector<BinNode>& pool, int nthreads) {
	std::vector<Prim> prims((size_t)ntri);
	BinNode root;
	reset(root.bmin, root.bmax);
	for (int i = 0; i < ntri; ++i) {
		Prim& p = prims[i];
		for (int a = 0; a < 3; ++a) {
			p.bmin[a] = box[(size_t)i * 6 + a];
			p.bmax[a] = box[(size_t)i * 6 + 3 + a];
			p.c[a] = 0.5f * p.bmin[a] + 0.5f * p.bmax[a];
		}
		p.tri = i;
		grow(root.bmin, root.bmax, p.bmin, p.bmax);
	}
	pool.clear();
	pool.reserve(2 * (size_t)ntri + 2);
	pool.push_back(root);
	if (ntri == 0) return;
	const bool parallel = nthreads > 1 && ntri >= 16384;
	std::vector<Range> deferred;
	const int handoff = parallel ? std::max(2048, ntri / (nthreads * 8)) : 0;
	build_range(pool, prims, Range{0, 0, ntri}, handoff, parallel ? &deferred : nullptr);
	if (!parallel || deferred.empty()) return;
	struct Job {
		Range r;
		std::vector<BinNode> local;  // local[0] mirrors the subtree's root
	};
	std::vector<Job> jobs(deferred.size());
	for (size_t k = 0; k < deferred.size(); ++k) jobs[k].r = deferred[k];
	std::atomic<size_t> next(0);
	auto worker = [&] {
		for (;;) {
			size_t k = next.fetch_add(1);
			if (k >= jobs.size()) break;
			Job& jb = jobs[k];
			jb.local.reserve(2 * (size_t)(jb.r.end - jb.r.start) + 2);
			jb.local.push_back(pool[jb.r.node]);
			build_range(jb.local, prims, Range{0, jb.r.start, jb.r.end}, 0, nullptr);  // disjoint slices of prims
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
	for (auto& t : th) t.join();
	for (Job& jb : jobs) {
		const int base = (int)pool.size() - 1;  // local index i > 0 -> pool index base + i
		auto fix = [&](int c) { return c > 0 ? base + c : c; };
		pool[jb.r.node].left = fix(jb.local[0].left);
		pool[jb.r.node].right = fix(jb.local[0].right);
		pool[jb.r.node].tri = jb.local[0].tri;
		for (size_t i = 1; i < jb.local.size(); ++i) {
			BinNode b = jb.local[i];
			b.left = fix(b.left);
			b.right = fix(b.right);
			pool.push_back(b);
		}
	}
}

// Leaves of two triangles.  Two triangles that share an edge and whose common box is hardly larger than the larger
// of their own boxes (the two halves of a quad of a tessellated surface: for an axis-aligned quad both boxes ARE the
// common box) are walked as one leaf: the tree over n/2 such leaves is about a level shallower, and a ray that
// reaches the quad parks one leaf instead of two.  The kernels still test each triangle of a leaf on its own (own
// box, own Triangle::hit), so what is found is unchanged.
// mate[i] = the triangle paired with i, or -1.  Shared edges are found by sorting the 3n edges by a hash of their end
// points (order-independent) and comparing the end points of equal hashes bit for bit.  Pairs are taken greedily in
// index order, the tightest class first (common box within 2 % of the larger own box), then up to maxGrowth.
inline void pair_triangles(const float* verts9, const std::vector<float>& box, int ntri, std::vector<int>& mate, int nthreads,
						   double maxGrowth = 1.3) {
	mate.assign((size_t)ntri, -1);
	if (ntri < 2) return;
	struct Edge {
		uint64_t key;
		int tri, e;
	};
	auto mix64 = [](uint64_t h, uint32_t v) {
		h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
		h *= 0xFF51AFD7ED558CCDull;
		h ^= h >> 33;
		return h;
	};
	auto vhash = [&](const float* p) {
		uint32_t u[3];
		std::memcpy(u, p, 12);
		for (uint32_t& w : u)
			if (w == 0x80000000u) w = 0;  // -0 and +0 are the same point
		return mix64(mix64(mix64(0x1234567ull, u[0]), u[1]), u[2]);
	};
	std::vector<Edge> edges((size_t)ntri * 3);
	auto fill = [&](int t0, int t1) {
		for (int t = t0; t < t1; ++t) {
			const float* v = verts9 + (size_t)t * 9;
			const uint64_t h[3] = {vhash(v), vhash(v + 3), vhash(v + 6)};
			for (int e = 0; e < 3; ++e) {
				const uint64_t a = h[e], b = h[(e + 1) % 3];
				edges[(size_t)t * 3 + e] = Edge{std::min(a, b) * 0x9E3779B97F4A7C15ull + std::max(a, b), t, e};
			}
		}
	};
	const int T = std::max(1, std::min(nthreads, ntri / 65536 + 1));
	{
		std::vector<std::thread> th;
		for (int k = 0; k < T; ++k) th.emplace_back(fill, (int)((long long)ntri * k / T), (int)((long long)ntri * (k + 1) / T));
		for (auto& t : th) t.join();
	}
	// sort by key: buckets by the top bits (serial counting pass), each bucket sorted on a thread of its own
	constexpr int BUCKET_BITS = 6, NB = 1 << BUCKET_BITS;
	std::vector<size_t> start(NB + 1, 0);
	for (const Edge& e : edges) start[(e.key >> (64 - BUCKET_BITS)) + 1]++;
	for (int b = 0; b < NB; ++b) start[b + 1] += start[b];
	std::vector<Edge> sorted(edges.size());
	{
		std::vector<size_t> pos(start.begin(), start.end() - 1);
		for (const Edge& e : edges) sorted[pos[e.key >> (64 - BUCKET_BITS)]++] = e;
	}
	edges.clear();
	edges.shrink_to_fit();
	auto by_key = [](const Edge& a, const Edge& b) { return a.key != b.key ? a.key < b.key : a.tri != b.tri ? a.tri < b.tri : a.e < b.e; };
	{
		std::atomic<int> next(0);
		auto work = [&] {
			for (;;) {
				const int b = next.fetch_add(1);
				if (b >= NB) break;
				std::sort(sorted.begin() + start[b], sorted.begin() + start[b + 1], by_key);
			}
		};
		std::vector<std::thread> th;
		for (int k = 0; k < T; ++k) th.emplace_back(work);
		for (auto& t : th) t.join();
	}
	auto same_point = [](const float* a, const float* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
	auto same_edge = [&](const Edge& x, const Edge& y) {
		const float* vx = verts9 + (size_t)x.tri * 9;
		const float* vy = verts9 + (size_t)y.tri * 9;
		const float *a0 = vx + 3 * x.e, *a1 = vx + 3 * ((x.e + 1) % 3), *b0 = vy + 3 * y.e, *b1 = vy + 3 * ((y.e + 1) % 3);
		return (same_point(a0, b0) && same_point(a1, b1)) || (same_point(a0, b1) && same_point(a1, b0));
	};
	auto growth = [&](int i, int j) {
		float mn[3], mx[3];
		for (int a = 0; a < 3; ++a) {
			mn[a] = std::min(box[(size_t)i * 6 + a], box[(size_t)j * 6 + a]);
			mx[a] = std::max(box[(size_t)i * 6 + 3 + a], box[(size_t)j * 6 + 3 + a]);
		}
		const double own = std::max(half_area(&box[(size_t)i * 6], &box[(size_t)i * 6 + 3]), half_area(&box[(size_t)j * 6], &box[(size_t)j * 6 + 3]));
		const double both = half_area(mn, mx);
		return own > 0.0 ? both / own : (both > 0.0 ? DBL_MAX : 1.0);
	};
	// every edge gets the triangle on its other side (the tightest one if there are several), bucket by bucket on the
	// threads: the slots written belong to the edges of the bucket
	std::vector<int> other((size_t)ntri * 3, -1);
	std::vector<float> otherGrowth((size_t)ntri * 3, FLT_MAX);
	{
		std::atomic<int> next(0);
		auto work = [&] {
			for (;;) {
				const int bkt = next.fetch_add(1);
				if (bkt >= NB) break;
				for (size_t a = start[bkt]; a < start[bkt + 1];) {
					size_t b = a + 1;
					while (b < start[bkt + 1] && sorted[b].key == sorted[a].key) ++b;
					if (b - a <= 8)  // an edge shared by more triangles than that is no surface
						for (size_t x = a; x < b; ++x)
							for (size_t y = a; y < b; ++y)
								if (sorted[x].tri != sorted[y].tri && same_edge(sorted[x], sorted[y])) {
									const double g = growth(sorted[x].tri, sorted[y].tri);
									const size_t slot = (size_t)sorted[x].tri * 3 + sorted[x].e;
									if (g <= maxGrowth && ((float)g < otherGrowth[slot] || ((float)g == otherGrowth[slot] && sorted[y].tri < other[slot]))) {
										otherGrowth[slot] = (float)g;
										other[slot] = sorted[y].tri;
									}
								}
					a = b;
				}
			}
		};
		std::vector<std::thread> th;
		for (int k = 0; k < T; ++k) th.emplace_back(work);
		for (auto& t : th) t.join();
	}
	// greedy, in index order: a free triangle takes the free neighbour with which it shares the tightest box
	for (int pass = 0; pass < 2; ++pass) {
		const float limit = pass == 0 ? 1.02f : (float)maxGrowth;
		for (int i = 0; i < ntri; ++i) {
			if (mate[i] >= 0) continue;
			int pick = -1;
			float g = FLT_MAX;
			for (int e = 0; e < 3; ++e) {
				const int o = other[(size_t)i * 3 + e];
				const float og = otherGrowth[(size_t)i * 3 + e];
				if (o < 0 || mate[o] >= 0 || og > limit) continue;
				if (og < g || (og == g && o < pick)) {
					g = og;
					pick = o;
				}
			}
			if (pick >= 0) {
				mate[i] = pick;
				mate[pick] = i;
			}
		}
	}
}

// Collapses a binary tree four wide and quantises it.  out: 16 floats per wide node,
//   px py pz sx                          origin; sx, sy, sz: quantisation steps (powers of two)
//   lo.x[4] hi.x[4] lo.y[4] hi.y[4]      one byte per child
//   lo.z[4] hi.z[4] code0 code1
//   code2 code3 sy sz                      code > 0 wide node, < 0 ~leafValue[N[.].tri] (a leaf), 0 absent
// Which binary nodes become wide nodes is decided by a table over the subtrees (pass 0): the choice with the
// smallest summed surface of wide nodes, about a sixth fewer nodes than opening the largest child first gives.
// stackNeed: entries a traversal may have waiting when it takes the nearest
// child directly and stacks the others.  Returns false when a box cannot be represented.
inline bool collapse_wide(const std::vector<BinNode>& N, const std::vector<int>& leafValue, std::vector<float>& out, int& stackNeed,
						  int nthreads = 1) {
	auto interior = [&](int i) { return N[i].tri < 0 && (N[i].left >= 0 || N[i].right >= 0); };
	auto usable = [&](int i) { return i >= 0 && (N[i].tri >= 0 || interior(i)); };
	// Pass 0 (serial, children before parents): the cheapest way to cover every subtree with at most j = 1..4 wide-tree
	// entries, cost = summed surface of the wide nodes it takes (the expected number of node visits of a random ray).
	//   F(leaf, j) = 0;  F(n, 1) = area(n) + G(n, 4);  F(n, j) = min(F(n, 1), G(n, j));
	//   G(n, j) = min over k of F(left, k) + F(right, j - k): n's content without a node for n itself
	struct Dp {
		double F[4];       // F[j-1]
		uint8_t split[3];  // split[j-2]: k of G(n, j)
	};
	std::vector<Dp> dp;
	{
		dp.resize(N.size());
		std::vector<int> order;
		order.reserve(N.size());
		order.push_back(0);
		for (size_t h = 0; h < order.size(); ++h)
			if (interior(order[h]))
				for (int ch : {N[order[h]].left, N[order[h]].right})
					if (usable(ch)) order.push_back(ch);
		for (size_t h = order.size(); h-- > 0;) {
			const int n = order[h];
			Dp& d = dp[n];
			const int l = N[n].left, r = N[n].right;
			if (!interior(n)) {
				for (double& f : d.F) f = 0.0;
				continue;
			}
			if (!usable(l) || !usable(r)) {  // a single child: its covers stand for n
				for (int j = 0; j < 4; ++j) d.F[j] = usable(l) ? dp[l].F[j] : usable(r) ? dp[r].F[j] : 0.0;
				d.split[0] = d.split[1] = d.split[2] = 0;
				continue;
			}
			double G[5] = {0, 0, 0, 0, 0};
			for (int j = 2; j <= 4; ++j) {
				double bestG = DBL_MAX;
				int bestK = 1;
				for (int k = 1; k < j; ++k) {
					const double g = dp[l].F[k - 1] + dp[r].F[j - k - 1];
					if (g < bestG) {
						bestG = g;
						bestK = k;
					}
				}
				G[j] = bestG;
				d.split[j - 2] = (uint8_t)bestK;
			}
			d.F[0] = half_area(N[n].bmin, N[n].bmax) + G[4];
			for (int j = 2; j <= 4; ++j) d.F[j - 1] = std::min(d.F[0], G[j]);
		}
	}
	// members of wide node x by the table: x's content in at most four entries
	auto members_dp = [&](int x, int m[4]) {
		int k = 0;
		if (N[x].tri >= 0) {
			m[k++] = x;
			return k;
		}
		struct Item { int n, j; bool open; };
		Item st[16];
		int sp = 0;
		st[sp++] = Item{x, 4, true};
		while (sp > 0) {
			const Item it = st[--sp];
			const int n = it.n, l = N[n].left, r = N[n].right;
			if (!interior(n)) {
				if (N[n].tri >= 0) m[k++] = n;  // (the root of an empty tree is neither leaf nor interior)
				continue;
			}
			if (!usable(l) || !usable(r)) {
				if (usable(l) || usable(r)) st[sp++] = Item{usable(l) ? l : r, it.j, it.open};
				continue;
			}
			if (!it.open) {
				// n itself as one entry, unless splitting it into at most j entries is cheaper
				const int kk = it.j >= 2 ? dp[n].split[it.j - 2] : 0;
				const double g = it.j >= 2 ? dp[l].F[kk - 1] + dp[r].F[it.j - kk - 1] : DBL_MAX;
				if (it.j < 2 || dp[n].F[0] <= g) {
					m[k++] = n;
					continue;
				}
			}
			const int kk = dp[n].split[it.j - 2];
			st[sp++] = Item{r, it.j - kk, false};
			st[sp++] = Item{l, kk, false};
		}
		return k;
	};
	// Pass 1 (serial, breadth first): which binary nodes become wide nodes, and each one's members.
	std::vector<int> wideOf(N.size(), 0), worder, mem;
	std::vector<uint8_t> cnt;
	worder.push_back(0);
	for (size_t h = 0; h < worder.size(); ++h) {
		int m[4] = {-1, -1, -1, -1};
		const int k = members_dp(worder[h], m);
		if (worder.size() > N.size()) return false;  // every binary node becomes a wide node at most once
		for (int c = 0; c < k; ++c)
			if (N[m[c]].tri < 0) {
				wideOf[m[c]] = (int)worder.size();
				worder.push_back(m[c]);
			}
		mem.insert(mem.end(), m, m + 4);
		cnt.push_back((uint8_t)k);
	}
	const size_t nw = worder.size();
	out.assign(nw * 16, 0.0f);
	auto bits = [](uint32_t u) {
		float f;
		std::memcpy(&f, &u, 4);
		return f;
	};
	// Pass 2 (independent per node, threads over ranges): boxes, steps, quantised bounds, codes.
	std::vector<uint8_t> bad((size_t)std::max(1, nthreads), 0);
	auto quantise = [&](size_t h0, size_t h1, int tid) {
		for (size_t h = h0; h < h1; ++h) {
			const int* m = &mem[h * 4];
			const int k = cnt[h];
			double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
			for (int c = 0; c < k; ++c)
				for (int a = 0; a < 3; ++a) {
					lo[a] = c ? std::min(lo[a], (double)N[m[c]].bmin[a]) : (double)N[m[c]].bmin[a];
					hi[a] = c ? std::max(hi[a], (double)N[m[c]].bmax[a]) : (double)N[m[c]].bmax[a];
				}
			uint32_t q[6] = {0, 0, 0, 0, 0, 0};  // lo.x hi.x lo.y hi.y lo.z hi.z
			float steps[3];
			for (int a = 0; a < 3; ++a) {
				// smallest power-of-two step with 254 steps >= extent (one step of slack for the rounding of
				// the double arithmetic below), with a biased float exponent in 1..254
				const double ext = hi[a] - lo[a];
				int e = 1;
				if (ext > 0.0) {
					int x;
					(void)std::frexp(ext / 254.0, &x);  // ext/254 = f * 2^x, f in [0.5, 1)
					e = std::min(254, std::max(1, x - 1 + 127));
				}
				while (e > 1 && 254.0 * std::ldexp(1.0, e - 1 - 127) >= ext) --e;
				while (e < 254 && 254.0 * std::ldexp(1.0, e - 127) < ext) ++e;
				const double step = std::ldexp(1.0, e - 127);
				if (254.0 * step < ext) bad[tid] = 1;
				steps[a] = (float)step;
				for (int c = 0; c < 4; ++c) {
					if (c >= k) {  // absent child: an empty interval, which no ray enters
						q[a * 2 + 0] |= 255u << (8 * c);
						continue;
					}
					const double cmin = N[m[c]].bmin[a], cmax = N[m[c]].bmax[a];
					uint32_t l = (uint32_t)std::min(255.0, std::max(0.0, std::floor((cmin - lo[a]) / step)));
					uint32_t u = (uint32_t)std::min(255.0, std::max(0.0, std::ceil((cmax - lo[a]) / step)));
					// A rounded double sum that compares strictly beyond a float proves the exact sum is at or
					// beyond it; move outwards until that holds (the ends of the range hold by construction).
					while (l > 0 && !(lo[a] + l * step < cmin)) --l;
					while (u < 255 && !(lo[a] + u * step > cmax)) ++u;
					q[a * 2 + 0] |= l << (8 * c);
					q[a * 2 + 1] |= u << (8 * c);
				}
			}
			int code[4] = {0, 0, 0, 0};
			for (int c = 0; c < k; ++c) code[c] = N[m[c]].tri >= 0 ? ~leafValue[N[m[c]].tri] : wideOf[m[c]];
			float* w = &out[h * 16];
			w[0] = (float)lo[0]; w[1] = (float)lo[1]; w[2] = (float)lo[2]; w[3] = steps[0];
			w[4] = bits(q[0]); w[5] = bits(q[1]); w[6] = bits(q[2]); w[7] = bits(q[3]);
			w[8] = bits(q[4]); w[9] = bits(q[5]); w[10] = bits((uint32_t)code[0]); w[11] = bits((uint32_t)code[1]);
			w[12] = bits((uint32_t)code[2]); w[13] = bits((uint32_t)code[3]); w[14] = steps[1]; w[15] = steps[2];
		}
	};
	if (nthreads > 1 && nw >= 65536) {
		std::vector<std::thread> th;
		const size_t per = (nw + (size_t)nthreads - 1) / (size_t)nthreads;
		for (int t = 0; t < nthreads; ++t) {
			const size_t h0 = std::min(nw, per * (size_t)t), h1 = std::min(nw, h0 + per);
			if (h0 < h1) th.emplace_back(quantise, h0, h1, t);
		}
		for (auto& t : th) t.join();
	}
	else
		quantise(0, nw, 0);
	bool ok = true;
	for (uint8_t b : bad) ok = ok && !b;
	// Pass 3 (serial, children before parents: a child's number is larger than its parent's): entries a
	// traversal may have waiting.
	std::vector<int> need(nw, 0);
	for (size_t h = nw; h-- > 0;) {
		const int* m = &mem[h * 4];
		const int k = cnt[h];
		int deepest = 0;
		for (int c = 0; c < k; ++c)
			if (N[m[c]].tri < 0) deepest = std::max(deepest, need[wideOf[m[c]]]);
		need[h] = std::max(0, k - 1) + deepest;
	}
	stackNeed = std::max(1, need[0]);
	return ok;
}

}  // namespace rtxwide
