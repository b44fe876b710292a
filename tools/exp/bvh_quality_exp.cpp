// Offline experiment (CPU, not shipped): how much traversal work would a better binary tree save?
// Builds over the same triangle soup: (1) LBVH (Morton, Karras-equivalent split on the highest differing bit),
// (2) binned-SAH top-down, (3) PLOC (radius r); collapses each to 4-ary (largest-area opening, leaves <= 4 tris)
// and counts node visits / triangle tests for short AO-like rays (closest hit, LIFO, cull by tbest).
// input: binary file  [u32 n][n * 9 floats]   usage: exp tris.bin
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <array>
#include <functional>
struct V3 { float x, y, z; };
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct Box { V3 lo{1e30f, 1e30f, 1e30f}, hi{-1e30f, -1e30f, -1e30f};
    void grow(V3 p) { lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)}; hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)}; }
    void grow(const Box& b) { grow(b.lo); grow(b.hi); }
    float area() const { V3 d = hi - lo; return 2.f * (d.x * d.y + d.y * d.z + d.z * d.x); } };
struct Tri { V3 a, b, c; };
struct BNode { Box box; int l = -1, r = -1; int first = 0, count = 0; };   // leaf if l < 0 (count == 1 for leaves at build)
static std::vector<Tri> T; static std::vector<Box> TB; static std::vector<V3> TC;

static uint64_t expand21(uint64_t v) { v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; }

// a binary tree = nodes + order (triangle ids in leaf order)
struct Tree { std::vector<BNode> n; std::vector<int> order; int root = 0; };

static int lbvh_rec(Tree& t, const std::vector<uint64_t>& key, int lo, int hi)   // [lo, hi]
{
    int id = (int)t.n.size(); t.n.emplace_back();
    if (lo == hi) { t.n[id].first = lo; t.n[id].count = 1; t.n[id].box = TB[t.order[lo]]; return id; }
    int split;
    if (key[lo] == key[hi]) split = (lo + hi) >> 1;
    else { int pre = __builtin_clzll(key[lo] ^ key[hi]); int s = lo, step = hi - lo;
        do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi && __builtin_clzll(key[lo] ^ key[ns]) > pre) s = ns; } while (step > 1);
        split = s; }
    int l = lbvh_rec(t, key, lo, split), r = lbvh_rec(t, key, split + 1, hi);
    t.n[id].l = l; t.n[id].r = r; t.n[id].first = lo; t.n[id].count = hi - lo + 1;
    t.n[id].box = t.n[l].box; t.n[id].box.grow(t.n[r].box); return id;
}
static std::vector<uint64_t> morton_sorted(std::vector<int>& order)
{
    Box cb; for (auto& c : TC) cb.grow(c);
    int n = (int)T.size(); std::vector<std::pair<uint64_t, int>> kv(n);
    V3 ext = cb.hi - cb.lo;
    for (int i = 0; i < n; ++i) {
        uint64_t x = (uint64_t)std::min(2097151.0, (TC[i].x - cb.lo.x) / std::max(ext.x, 1e-20f) * 2097152.0);
        uint64_t y = (uint64_t)std::min(2097151.0, (TC[i].y - cb.lo.y) / std::max(ext.y, 1e-20f) * 2097152.0);
        uint64_t z = (uint64_t)std::min(2097151.0, (TC[i].z - cb.lo.z) / std::max(ext.z, 1e-20f) * 2097152.0);
        kv[i] = {expand21(x) << 2 | expand21(y) << 1 | expand21(z), i}; }
    std::sort(kv.begin(), kv.end());
    order.resize(n); std::vector<uint64_t> key(n);
    for (int i = 0; i < n; ++i) { key[i] = kv[i].first; order[i] = kv[i].second; }
    return key;
}
static Tree build_lbvh() { Tree t; auto key = morton_sorted(t.order); t.n.reserve(2 * T.size()); t.root = lbvh_rec(t, key, 0, (int)T.size() - 1); return t; }

static int sah_rec(Tree& t, int lo, int hi)   // [lo, hi)
{
    int id = (int)t.n.size(); t.n.emplace_back();
    Box b, cb; for (int i = lo; i < hi; ++i) { b.grow(TB[t.order[i]]); cb.grow(TC[t.order[i]]); }
    t.n[id].box = b; t.n[id].first = lo; t.n[id].count = hi - lo;
    if (hi - lo == 1) return id;
    const int NB = 16; float best = 1e30f; int bax = -1, bsp = 0;
    for (int ax = 0; ax < 3; ++ax) {
        float c0 = (&cb.lo.x)[ax], c1 = (&cb.hi.x)[ax]; if (c1 - c0 < 1e-9f) continue;
        Box bb[NB]; int bc[NB] = {0};
        for (int i = lo; i < hi; ++i) { int k = std::min(NB - 1, (int)(((&TC[t.order[i]].x)[ax] - c0) / (c1 - c0) * NB)); bb[k].grow(TB[t.order[i]]); bc[k]++; }
        float ra[NB]; Box acc; int cnt = 0; int rc[NB];
        for (int k = NB - 1; k > 0; --k) { if (bc[k]) acc.grow(bb[k]); cnt += bc[k]; ra[k] = cnt ? acc.area() : 0.f; rc[k] = cnt; }
        Box la; int lc = 0;
        for (int k = 0; k < NB - 1; ++k) { if (bc[k]) la.grow(bb[k]); lc += bc[k]; if (lc == 0 || rc[k + 1] == 0) continue; float c = la.area() * lc + ra[k + 1] * rc[k + 1]; if (c < best) { best = c; bax = ax; bsp = k; } }
    }
    int mid;
    if (bax < 0) mid = (lo + hi) >> 1;
    else { float c0 = (&cb.lo.x)[bax], c1 = (&cb.hi.x)[bax];
        mid = (int)(std::partition(t.order.begin() + lo, t.order.begin() + hi, [&](int i) { return std::min(NB - 1, (int)(((&TC[i].x)[bax] - c0) / (c1 - c0) * NB)) <= bsp; }) - t.order.begin());
        if (mid == lo || mid == hi) mid = (lo + hi) >> 1; }
    int l = sah_rec(t, lo, mid), r = sah_rec(t, mid, hi); t.n[id].l = l; t.n[id].r = r; return id;
}
// hybrid: LBVH on top, every maximal subtree with <= K triangles rebuilt top-down by (binned) SAH over its own
// contiguous Morton range -- ranges stay contiguous, so the rest of an LBVH pipeline is untouched
static int hybrid_rec(Tree& out, const Tree& lb, int n, int K)
{
    const BNode& b = lb.n[n];
    if (b.count <= K) return sah_rec(out, b.first, b.first + b.count);
    int id = (int)out.n.size(); out.n.emplace_back();
    int l = hybrid_rec(out, lb, b.l, K), r = hybrid_rec(out, lb, b.r, K);
    out.n[id].l = l; out.n[id].r = r; out.n[id].first = b.first; out.n[id].count = b.count;
    out.n[id].box = out.n[l].box; out.n[id].box.grow(out.n[r].box);
    return id;
}
static Tree build_hybrid(int K) { Tree lb = build_lbvh(); Tree t; t.order = lb.order; t.n.reserve(2 * T.size()); t.root = hybrid_rec(t, lb, lb.root, K); return t; }

// hybrid 2: LBVH on top, every maximal subtree with <= K triangles rebuilt by PLOC over its own Morton range (the
// block-local form a GPU would run in LDS); leaf order inside the range is the DFS order of the rebuilt subtree
static int ploc_range(Tree& out, int lo, int hi, int radius)     // [lo, hi) of out.order; returns root, reorders the range
{
    std::vector<int> cl, tri_of; int base = (int)out.n.size();
    for (int i = lo; i < hi; ++i) { out.n.emplace_back(); out.n.back().box = TB[out.order[i]]; out.n.back().count = 1; cl.push_back((int)out.n.size() - 1); }
    std::vector<int> leaf_tri(out.order.begin() + lo, out.order.begin() + hi);
    while (cl.size() > 1) { int m = (int)cl.size(); std::vector<int> nn(m);
        for (int i = 0; i < m; ++i) { float best = 1e30f; int bj = -1;
            for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); ++j) { if (j == i) continue; Box b = out.n[cl[i]].box; b.grow(out.n[cl[j]].box); float a = b.area(); if (a < best) { best = a; bj = j; } }
            nn[i] = bj; }
        std::vector<int> next;
        for (int i = 0; i < m; ++i) { int j = nn[i];
            if (nn[j] == i) { if (i < j) { int id = (int)out.n.size(); out.n.emplace_back(); out.n[id].l = cl[i]; out.n[id].r = cl[j]; out.n[id].box = out.n[cl[i]].box; out.n[id].box.grow(out.n[cl[j]].box); out.n[id].count = out.n[cl[i]].count + out.n[cl[j]].count; next.push_back(id); } }
            else next.push_back(cl[i]); }
        cl.swap(next); }
    // DFS order -> first/count, reorder the range
    int root = cl[0], pos = lo; std::vector<std::pair<int, int>> st{{root, 0}};
    while (!st.empty()) { auto [n, ph] = st.back(); st.pop_back();
        if (out.n[n].l < 0) { out.n[n].first = pos; out.order[pos++] = leaf_tri[n - base]; continue; }
        if (ph == 0) { st.push_back({n, 1}); st.push_back({out.n[n].r, 0}); st.push_back({out.n[n].l, 0}); }
        else out.n[n].first = out.n[out.n[n].l].first; }
    return root;
}
static int hybrid2_rec(Tree& out, const Tree& lb, int n, int K, int radius)
{
    const BNode& b = lb.n[n];
    if (b.count <= K) return ploc_range(out, b.first, b.first + b.count, radius);
    int id = (int)out.n.size(); out.n.emplace_back();
    int l = hybrid2_rec(out, lb, b.l, K, radius), r = hybrid2_rec(out, lb, b.r, K, radius);
    out.n[id].l = l; out.n[id].r = r; out.n[id].first = b.first; out.n[id].count = b.count;
    out.n[id].box = out.n[l].box; out.n[id].box.grow(out.n[r].box);
    return id;
}
static Tree build_hybrid2(int K, int radius) { Tree lb = build_lbvh(); Tree t; t.order = lb.order; t.n.reserve(3 * T.size()); t.root = hybrid2_rec(t, lb, lb.root, K, radius); return t; }

// two-level: treelets (<= K, local PLOC) + the upper tree rebuilt by PLOC over the treelet roots in Morton order
static void collect_cut(const Tree& lb, int n, int K, std::vector<int>& cut) { if (lb.n[n].count <= K) { cut.push_back(n); return; } collect_cut(lb, lb.n[n].l, K, cut); collect_cut(lb, lb.n[n].r, K, cut); }
static Tree build_two_level(int K, int radius, int radius_top)
{
    Tree lb = build_lbvh(); Tree t; t.order = lb.order; t.n.reserve(3 * T.size());
    std::vector<int> cut; collect_cut(lb, lb.root, K, cut);
    std::vector<int> cl;
    for (int c : cut) cl.push_back(ploc_range(t, lb.n[c].first, lb.n[c].first + lb.n[c].count, radius));
    for (int id : cl) if (t.n[id].l < 0) { /* single leaf: ploc_range set first */ }
    while (cl.size() > 1) { int m = (int)cl.size(); std::vector<int> nn(m);
        for (int i = 0; i < m; ++i) { float best = 1e30f; int bj = -1;
            for (int j = std::max(0, i - radius_top); j <= std::min(m - 1, i + radius_top); ++j) { if (j == i) continue; Box b = t.n[cl[i]].box; b.grow(t.n[cl[j]].box); float a = b.area(); if (a < best) { best = a; bj = j; } }
            nn[i] = bj; }
        std::vector<int> next;
        for (int i = 0; i < m; ++i) { int j = nn[i];
            if (nn[j] == i) { if (i < j) { int id = (int)t.n.size(); t.n.emplace_back(); t.n[id].l = cl[i]; t.n[id].r = cl[j]; t.n[id].box = t.n[cl[i]].box; t.n[id].box.grow(t.n[cl[j]].box); t.n[id].count = t.n[cl[i]].count + t.n[cl[j]].count; next.push_back(id); } }
            else next.push_back(cl[i]); }
        cl.swap(next); }
    t.root = cl[0];
    // global DFS re-order (ranges of whole treelets move)
    std::vector<int> old = t.order, neworder; neworder.reserve(old.size());
    std::vector<std::pair<int, int>> st{{t.root, 0}};
    while (!st.empty()) { auto [n, ph] = st.back(); st.pop_back();
        if (t.n[n].l < 0) { int tri = old[t.n[n].first]; t.n[n].first = (int)neworder.size(); neworder.push_back(tri); continue; }
        if (ph == 0) { st.push_back({n, 1}); st.push_back({t.n[n].r, 0}); st.push_back({t.n[n].l, 0}); }
        else t.n[n].first = t.n[t.n[n].l].first; }
    t.order = neworder;
    return t;
}


// hybrid 3: binned SAH on top down to ranges of <= K triangles, each range then put into Morton order and rebuilt by
// local PLOC (what a "fast trace" GPU build would do: SAH level passes over the whole array, then the LDS treelets)
static std::vector<uint64_t> g_mkey;   // per triangle
static int g_one_axis = 0;
static int sahtop_rec(Tree& t, int lo, int hi, int K, int radius)
{
    if (hi - lo <= K) { std::sort(t.order.begin() + lo, t.order.begin() + hi, [&](int a, int b) { return g_mkey[a] < g_mkey[b]; }); return ploc_range(t, lo, hi, radius); }
    Box b, cb; for (int i = lo; i < hi; ++i) { b.grow(TB[t.order[i]]); cb.grow(TC[t.order[i]]); }
    const int NB = 16; int only = -1; if (g_one_axis) { V3 e = cb.hi - cb.lo; only = e.x >= e.y && e.x >= e.z ? 0 : (e.y >= e.z ? 1 : 2); } float best = 1e30f; int bax = -1, bsp = 0;
    for (int ax = 0; ax < 3; ++ax) { if (only >= 0 && ax != only) continue;
        float c0 = (&cb.lo.x)[ax], c1 = (&cb.hi.x)[ax]; if (c1 - c0 < 1e-9f) continue;
        Box bb[NB]; int bc[NB] = {0};
        for (int i = lo; i < hi; ++i) { int k = std::min(NB - 1, (int)(((&TC[t.order[i]].x)[ax] - c0) / (c1 - c0) * NB)); bb[k].grow(TB[t.order[i]]); bc[k]++; }
        float ra[NB]; Box acc; int cnt = 0; int rc[NB];
        for (int k = NB - 1; k > 0; --k) { if (bc[k]) acc.grow(bb[k]); cnt += bc[k]; ra[k] = cnt ? acc.area() : 0.f; rc[k] = cnt; }
        Box la; int lc = 0;
        for (int k = 0; k < NB - 1; ++k) { if (bc[k]) la.grow(bb[k]); lc += bc[k]; if (lc == 0 || rc[k + 1] == 0) continue; float c = la.area() * lc + ra[k + 1] * rc[k + 1]; if (c < best) { best = c; bax = ax; bsp = k; } }
    }
    int mid;
    if (bax < 0) mid = (lo + hi) >> 1;
    else { float c0 = (&cb.lo.x)[bax], c1 = (&cb.hi.x)[bax];
        mid = (int)(std::partition(t.order.begin() + lo, t.order.begin() + hi, [&](int i) { return std::min(NB - 1, (int)(((&TC[i].x)[bax] - c0) / (c1 - c0) * NB)) <= bsp; }) - t.order.begin());
        if (mid == lo || mid == hi) mid = (lo + hi) >> 1; }
    int l = sahtop_rec(t, lo, mid, K, radius), r = sahtop_rec(t, mid, hi, K, radius);
    int id = (int)t.n.size(); t.n.emplace_back(); t.n[id].l = l; t.n[id].r = r; t.n[id].first = lo; t.n[id].count = hi - lo; t.n[id].box = b; return id;
}
static Tree build_sahtop(int K, int radius)
{
    Tree t; std::vector<int> ord; auto key = morton_sorted(ord); g_mkey.resize(T.size()); for (size_t i = 0; i < T.size(); ++i) g_mkey[ord[i]] = key[i];
    t.order = ord; t.n.reserve(3 * T.size()); t.root = sahtop_rec(t, 0, (int)T.size(), K, radius); return t;
}
static Tree build_sah() { Tree t; t.order.resize(T.size()); for (size_t i = 0; i < T.size(); ++i) t.order[i] = (int)i; t.n.reserve(2 * T.size()); t.root = sah_rec(t, 0, (int)T.size()); return t; }

// PLOC over Morton-sorted leaves; leaf order is rebuilt by DFS afterwards
static void dfs_order(Tree& t, int id, std::vector<int>& out, const std::vector<int>& leaf_tri)
{
    std::vector<int> st{id};
    // iterative post-fix of first/count
    std::vector<std::pair<int, int>> stack{{id, 0}};
    while (!stack.empty()) { auto [n, ph] = stack.back(); stack.pop_back();
        if (t.n[n].l < 0) { t.n[n].first = (int)out.size(); t.n[n].count = 1; out.push_back(leaf_tri[n]); continue; }
        if (ph == 0) { stack.push_back({n, 1}); stack.push_back({t.n[n].r, 0}); stack.push_back({t.n[n].l, 0}); }
        else { t.n[n].first = t.n[t.n[n].l].first; t.n[n].count = t.n[t.n[n].l].count + t.n[t.n[n].r].count; } }
}
static Tree build_ploc(int radius)
{
    Tree t; std::vector<int> order; morton_sorted(order);
    int n = (int)T.size(); t.n.reserve(2 * n); std::vector<int> leaf_tri(2 * n, -1);
    std::vector<int> cl(n);
    for (int i = 0; i < n; ++i) { t.n.emplace_back(); t.n[i].box = TB[order[i]]; leaf_tri[i] = order[i]; cl[i] = i; }
    int iters = 0;
    while (cl.size() > 1) { ++iters; int m = (int)cl.size(); std::vector<int> nn(m);
        for (int i = 0; i < m; ++i) { float best = 1e30f; int bj = -1;
            for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); ++j) { if (j == i) continue; Box b = t.n[cl[i]].box; b.grow(t.n[cl[j]].box); float a = b.area(); if (a < best) { best = a; bj = j; } }
            nn[i] = bj; }
        std::vector<int> next; next.reserve(m);
        for (int i = 0; i < m; ++i) { int j = nn[i];
            if (nn[j] == i) { if (i < j) { int id = (int)t.n.size(); t.n.emplace_back(); t.n[id].l = cl[i]; t.n[id].r = cl[j]; t.n[id].box = t.n[cl[i]].box; t.n[id].box.grow(t.n[cl[j]].box); next.push_back(id); } }
            else next.push_back(cl[i]); }
        cl.swap(next); }
    t.root = cl[0]; dfs_order(t, t.root, t.order, leaf_tri);
    fprintf(stderr, "ploc r=%d iterations %d\n", radius, iters);
    return t;
}

static int LEAFMAX = 4;
// 4-ary collapse: open largest-area child with > LEAFMAX tris, up to twice
static int WIDTH = 4;      // env WIDTH=8: 8-ary collapse (open the largest child up to 6 times)
struct WNode { Box cb[8]; int ref[8]; int nch; };   // ref >= 0 wide node, < 0 leaf ~(binary node)
struct Wide { std::vector<WNode> w; const Tree* t; };
static Wide collapse(const Tree& t)
{
    Wide W; W.t = &t; std::vector<int> bin{t.root}; W.w.emplace_back();
    for (size_t wi = 0; wi < bin.size(); ++wi) {
        int b = bin[wi]; std::vector<int> ch;
        if (t.n[b].count <= LEAFMAX || t.n[b].l < 0) ch = {b};
        else { ch = {t.n[b].l, t.n[b].r};
            for (int k = 0; k < WIDTH - 2; ++k) { int bi = -1; float ba = -1;
                for (size_t c = 0; c < ch.size(); ++c) if (t.n[ch[c]].count > LEAFMAX && t.n[ch[c]].l >= 0 && t.n[ch[c]].box.area() > ba) { ba = t.n[ch[c]].box.area(); bi = (int)c; }
                if (bi < 0) break; int o = ch[bi]; ch[bi] = t.n[o].l; ch.push_back(t.n[o].r); } }
        WNode wn; wn.nch = (int)ch.size();
        for (int c = 0; c < wn.nch; ++c) { wn.cb[c] = t.n[ch[c]].box;
            if (t.n[ch[c]].count <= LEAFMAX || t.n[ch[c]].l < 0) wn.ref[c] = ~ch[c];
            else { wn.ref[c] = (int)bin.size(); bin.push_back(ch[c]); W.w.emplace_back(); } }
        W.w[wi] = wn; }
    return W;
}
// SAH-optimal 4-ary collapse by dynamic programming over the binary tree (Ylitie, Karras, Laine 2017, for width 4):
// D[n][j] = cheapest way to represent subtree n as at most j+1 roots (wide nodes or leaves).
static float CT = 0.6f;   // cost of a triangle test relative to a node step
static Wide collapse_dp(const Tree& t)
{
    const int N = (int)t.n.size();
    std::vector<std::array<float, 4>> D(N); std::vector<std::array<signed char, 4>> split(N);   // split[n][j]: roots given to the left child (0 = keep n as one root)
    std::vector<float> cint(N, 1e30f); std::vector<signed char> isplit(N, 0), isleaf(N, 0);
    // post-order
    std::vector<int> order; order.reserve(N); { std::vector<int> st{t.root}; while (!st.empty()) { int n = st.back(); st.pop_back(); order.push_back(n); if (t.n[n].l >= 0) { st.push_back(t.n[n].l); st.push_back(t.n[n].r); } } }
    for (int oi = N - 1; oi >= 0; --oi) { int n = order[oi]; const BNode& b = t.n[n]; float A = b.box.area();
        float cleaf = b.count <= LEAFMAX ? A * CT * b.count : 1e30f;
        if (b.l < 0) { for (int j = 0; j < 4; ++j) { D[n][j] = A * CT * b.count; split[n][j] = 0; } isleaf[n] = 1; continue; }
        int l = b.l, r = b.r;
        float best = 1e30f; int bi = 1;
        for (int i = 1; i <= 3; ++i) { float c = D[l][i - 1] + D[r][4 - i - 1]; if (c < best) { best = c; bi = i; } }
        cint[n] = A + best; isplit[n] = (signed char)bi;
        float c1 = std::min(cleaf, cint[n]); isleaf[n] = cleaf <= cint[n];
        D[n][0] = c1; split[n][0] = 0;
        for (int j = 1; j < 4; ++j) { float bestj = D[n][j - 1]; int bs = split[n][j - 1] ? split[n][j - 1] : 0; bool own = true;
            for (int i = 1; i <= j; ++i) { float c = D[l][i - 1] + D[r][j + 1 - i - 1]; if (c < bestj) { bestj = c; bs = i; own = false; } }
            D[n][j] = bestj; split[n][j] = own ? split[n][j - 1] : (signed char)bs; }
    }
    // materialise
    Wide W; W.t = &t; std::vector<int> bin{t.root}; W.w.emplace_back();
    // gather roots of subtree n when allowed j+1 roots
    std::function<void(int, int, std::vector<int>&)> roots = [&](int n, int j, std::vector<int>& out) {
        int sp = split[n][j]; if (sp == 0 || t.n[n].l < 0) { out.push_back(n); return; }
        roots(t.n[n].l, sp - 1, out); roots(t.n[n].r, j + 1 - sp - 1, out); };
    for (size_t wi = 0; wi < bin.size(); ++wi) { int b = bin[wi]; std::vector<int> ch;
        if (t.n[b].l < 0 || (isleaf[b] && t.n[b].count <= LEAFMAX)) ch = {b};
        else { int sp = isplit[b]; roots(t.n[b].l, sp - 1, ch); roots(t.n[b].r, 4 - sp - 1, ch); }
        WNode wn; wn.nch = (int)ch.size();
        for (int c = 0; c < wn.nch; ++c) { wn.cb[c] = t.n[ch[c]].box;
            if (t.n[ch[c]].l < 0 || (isleaf[ch[c]] && t.n[ch[c]].count <= LEAFMAX)) wn.ref[c] = ~ch[c];
            else { wn.ref[c] = (int)bin.size(); bin.push_back(ch[c]); W.w.emplace_back(); } }
        W.w[wi] = wn; }
    return W;
}
static bool slab(const Box& b, V3 o, V3 inv, float tmin, float tmax)
{
    float t0 = (b.lo.x - o.x) * inv.x, t1 = (b.hi.x - o.x) * inv.x; if (t0 > t1) std::swap(t0, t1); tmin = std::max(tmin, t0); tmax = std::min(tmax, t1);
    t0 = (b.lo.y - o.y) * inv.y; t1 = (b.hi.y - o.y) * inv.y; if (t0 > t1) std::swap(t0, t1); tmin = std::max(tmin, t0); tmax = std::min(tmax, t1);
    t0 = (b.lo.z - o.z) * inv.z; t1 = (b.hi.z - o.z) * inv.z; if (t0 > t1) std::swap(t0, t1); tmin = std::max(tmin, t0); tmax = std::min(tmax, t1);
    return tmin <= tmax;
}
static bool tri_hit(const Tri& tr, V3 o, V3 d, float tmin, float& tbest)
{
    V3 e1 = tr.b - tr.a, e2 = tr.c - tr.a, p = cross(d, e2); float det = dot(e1, p); if (std::fabs(det) < 1e-12f) return false;
    float id = 1.f / det; V3 tv = o - tr.a; float u = dot(tv, p) * id; if (u < 0 || u > 1) return false;
    V3 q = cross(tv, e1); float v = dot(d, q) * id; if (v < 0 || u + v > 1) return false; float tt = dot(e2, q) * id;
    if (tt > tmin && tt < tbest) { tbest = tt; return true; } return false;
}
static void trace(const Wide& W, V3 o, V3 d, float tmax, uint64_t& nodes, uint64_t& tris)
{
    V3 inv{1.f / d.x, 1.f / d.y, 1.f / d.z}; float tbest = tmax; std::vector<int> st{0};
    while (!st.empty()) { int r = st.back(); st.pop_back();
        if (r >= 0) { ++nodes; const WNode& w = W.w[r]; for (int c = 0; c < w.nch; ++c) if (slab(w.cb[c], o, inv, 1e-3f, tbest)) st.push_back(w.ref[c]); }
        else { const BNode& b = W.t->n[~r]; for (int k = 0; k < b.count; ++k) { ++tris; tri_hit(T[W.t->order[b.first + k]], o, d, 1e-3f, tbest); } } }
}
static double sah_cost(const Tree& t) { double c = 0; float ra = t.n[t.root].box.area(); for (auto& n : t.n) c += (n.l < 0 ? 1.0 : 1.2) * n.box.area() / ra; return c; }

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); uint32_t n; if (fread(&n, 4, 1, f) != 1) return 1; T.resize(n); if (fread(T.data(), 36, n, f) != n) return 1; fclose(f);
    TB.resize(n); TC.resize(n);
    for (uint32_t i = 0; i < n; ++i) { TB[i].grow(T[i].a); TB[i].grow(T[i].b); TB[i].grow(T[i].c); TC[i] = (TB[i].lo + TB[i].hi) * 0.5f; }
    // rays: origin on a random triangle, cosine-ish hemisphere about its normal, both sides
    std::mt19937 rng(7); std::uniform_real_distribution<float> U(0, 1); int NR = 200000; std::vector<V3> ro(NR), rd(NR);
    for (int i = 0; i < NR; ++i) { const Tri& tr = T[rng() % n]; float u = U(rng), v = U(rng); if (u + v > 1) { u = 1 - u; v = 1 - v; }
        V3 p = tr.a + (tr.b - tr.a) * u + (tr.c - tr.a) * v; V3 nn = cross(tr.b - tr.a, tr.c - tr.a); float l = std::sqrt(dot(nn, nn)); if (l < 1e-12f) { --i; continue; } nn = nn * (1.f / l);
        if (nn.y < 0) nn = nn * -1.f; V3 dd; do { dd = {U(rng) * 2 - 1, U(rng) * 2 - 1, U(rng) * 2 - 1}; } while (dot(dd, dd) > 1 || dot(dd, dd) < 1e-4f); dd = dd * (1.f / std::sqrt(dot(dd, dd))); if (dot(dd, nn) < 0) dd = dd * -1.f;
        ro[i] = p + nn * 1e-3f; rd[i] = dd; }
    float radius = argc > 2 ? atof(argv[2]) : 1.2f;
    if (getenv("WIDTH")) WIDTH = atoi(getenv("WIDTH"));
    if (argc > 3) LEAFMAX = atoi(argv[3]);
    auto eval = [&](const char* name, const Tree& t) { if (getenv("DP")) { CT = atof(getenv("DP")); Wide W = collapse_dp(t); uint64_t nv = 0, tt = 0; for (int i = 0; i < NR; ++i) trace(W, ro[i], rd[i], radius, nv, tt);
        printf("%-10s DP collapse (ct %.2f): wide nodes %zu  nodes/ray %.2f  tris/ray %.2f\n", name, CT, W.w.size(), (double)nv / NR, (double)tt / NR); fflush(stdout); }
        Wide W = collapse(t); uint64_t nv = 0, tt = 0; for (int i = 0; i < NR; ++i) trace(W, ro[i], rd[i], radius, nv, tt);
        printf("%-10s binary SAH %.1f  wide nodes %zu  nodes/ray %.2f  tris/ray %.2f\n", name, sah_cost(t), W.w.size(), (double)nv / NR, (double)tt / NR); fflush(stdout); };
    bool all = argc > 4;
    if (all) {
    { Tree t = build_lbvh(); eval("lbvh", t); }
    { Tree t = build_hybrid(1024); eval("hyb1024", t); }
    { Tree t = build_hybrid(4096); eval("hyb4096", t); }
    { Tree t = build_hybrid2(256, 8); eval("lploc256", t); }
    { Tree t = build_two_level(1024, 8, 8); eval("2lvl r8", t); }
    { Tree t = build_ploc(8); eval("ploc r8", t); }
    }
    { Tree t = build_hybrid2(1024, 8); eval("lploc1024", t); }
    { Tree t = build_sahtop(1024, 8); eval("sahtop1024", t); }
    { g_one_axis = 1; Tree t = build_sahtop(1024, 8); eval("sahtop1024-1ax", t); g_one_axis = 0; }
    { Tree t = build_sahtop(512, 8); eval("sahtop512", t); }
    { Tree t = build_sahtop(256, 8); eval("sahtop256", t); }
    { Tree t = build_sah(); eval("sah16", t); }
    return 0;
}
