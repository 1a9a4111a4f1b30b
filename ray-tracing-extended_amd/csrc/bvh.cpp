// bvh.cpp — binned-SAH binary build, collapsed to 4-wide nodes in breadth-first order (hot upper levels
// end up contiguous).  See bvh.hpp for why a hierarchy is legal behind the reference's flat chunk loop.
#include "bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>

namespace rtbvh {
namespace {

struct Box {
    float mn[3], mx[3];
    void reset()
    {
        for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
    }
    void grow(const float* p)
    {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    void grow(const Box& b)
    {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
    }
    float half_area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BNode {            // binary build node
    Box      box;
    int32_t  left = -1, right = -1;
    uint32_t first = 0, count = 0;   // leaf when count > 0
};

struct Builder {
    const float* pos; size_t stride; uint32_t n;
    std::vector<Box>      tbox;
    std::vector<float>    cent;      // 3 per triangle
    std::vector<uint32_t> idx;
    std::vector<BNode>    bn;
    int depth = 0;
    int max_leaf = kMaxLeaf;
    int g_bins = 32;            // SAH bins per axis (Tuning::bins)
    float g_cost_exp = 1.0f;    // the SAH's subtree-cost model: area * count^g_cost_exp

    int build_range(uint32_t first, uint32_t count, int level)
    {
        depth = std::max(depth, level);
        int me = (int)bn.size();
        bn.emplace_back();
        Box b; b.reset();
        Box cb; cb.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            b.grow(tbox[idx[i]]);
            cb.grow(&cent[3 * (size_t)idx[i]]);
        }
        bn[me].box = b;
        if (count <= (uint32_t)max_leaf) {
            bn[me].first = first; bn[me].count = count;
            return me;
        }
        constexpr int NBMAX = 128;
        const int NB = g_bins;
        int best_axis = -1, best_split = -1; float best_cost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
            float lo = cb.mn[a], ext = cb.mx[a] - cb.mn[a];
            if (!(ext > 0.f)) continue;
            Box bb[NBMAX]; uint32_t bc[NBMAX];
            for (int k = 0; k < NB; ++k) { bb[k].reset(); bc[k] = 0; }
            float scale = (float)NB / ext;
            for (uint32_t i = first; i < first + count; ++i) {
                uint32_t t = idx[i];
                int k = (int)((cent[3 * (size_t)t + a] - lo) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                bb[k].grow(tbox[t]); bc[k]++;
            }
            float la[NBMAX]; uint32_t lc[NBMAX];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int k = 0; k < NB - 1; ++k) { acc.grow(bb[k]); c += bc[k]; la[k] = acc.half_area(); lc[k] = c; }
            acc.reset(); c = 0;
            for (int k = NB - 1; k > 0; --k) {
                acc.grow(bb[k]); c += bc[k];
                if (lc[k - 1] == 0 || c == 0) continue;
                float cost = g_cost_exp == 1.0f ? la[k - 1] * (float)lc[k - 1] + acc.half_area() * (float)c
                                                : la[k - 1] * std::pow((float)lc[k - 1], g_cost_exp) + acc.half_area() * std::pow((float)c, g_cost_exp);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
        }
        // Large-triangle isolation: centroid binning can never separate a triangle whose box spans the node (a floor quad
        // under 100k small triangles) from the rest, so it would sink to the bottom of the tree and inflate every box on
        // its way.  Candidate split: the one or two largest triangles of the node against everything else.
        {
            uint32_t big[2] = { 0, 0 }; float bigA[2] = { -1.f, -1.f };
            for (uint32_t i = first; i < first + count; ++i) {
                const float ar = tbox[idx[i]].half_area();
                if (ar > bigA[0]) { bigA[1] = bigA[0]; big[1] = big[0]; bigA[0] = ar; big[0] = i; }
                else if (ar > bigA[1]) { bigA[1] = ar; big[1] = i; }
            }
            const float nodeA = b.half_area();
            if (bigA[0] > 0.25f * nodeA && count > 2) {
                for (int k = 1; k <= 2; ++k) {
                    if (k == 2 && !(bigA[1] > 0.25f * nodeA && count > 3)) break;
                    Box in; in.reset(); Box rest; rest.reset();
                    for (uint32_t i = first; i < first + count; ++i) {
                        const bool isBig = i == big[0] || (k == 2 && i == big[1]);
                        (isBig ? in : rest).grow(tbox[idx[i]]);
                    }
                    const float ce = g_cost_exp;
                    const float cost = in.half_area() * std::pow((float)k, ce) + rest.half_area() * std::pow((float)(count - k), ce);
                    if (cost < best_cost) { best_cost = cost; best_axis = 3 + k; }
                }
                if (best_axis >= 4) {
                    const int k = best_axis - 3;
                    // move the k big triangles to the front of the range (order inside a range is free)
                    uint32_t a0 = big[0], a1 = big[1];
                    std::swap(idx[first], idx[a0]);
                    if (k == 2) { if (a1 == first) a1 = a0; std::swap(idx[first + 1], idx[a1]); }
                    int l = build_range(first, (uint32_t)k, level + 1);
                    int r = build_range(first + k, count - k, level + 1);
                    bn[me].left = l; bn[me].right = r;
                    return me;
                }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            float lo = cb.mn[best_axis], scale = (float)NB / (cb.mx[best_axis] - cb.mn[best_axis]);
            int a = best_axis, s = best_split;
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t t) {
                int k = (int)((cent[3 * (size_t)t + a] - lo) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                return k < s;
            });
            mid = (uint32_t)(it - idx.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        } else {
            mid = first + count / 2;     // all centroids coincide
        }
        int l = build_range(first, mid - first, level + 1);
        int r = build_range(mid, first + count - mid, level + 1);
        bn[me].left = l; bn[me].right = r;
        return me;
    }
};

// Insertion-based optimisation of the binary tree (after Bittner, Hapala, Havran, "Fast insertion-based optimization of
// bounding volume hierarchies", CGF 2013): every subtree in turn is cut out (its parent node goes with it, the sibling
// moves up) and put back where it adds the least surface area to the tree — found by a branch-and-bound search over
// (area the ancestors grow by) + (area of the new common parent).  Only links and boxes change; leaves keep their triangle
// ranges, so the traversal still returns the same closest hit.
struct Reinserter {
    std::vector<BNode>& bn;
    std::vector<int32_t> parent;
    int root;
    struct Cand { float induced; int node; bool operator<(const Cand& o) const { return induced > o.induced; } };
    std::vector<Cand> heap;

    Reinserter(std::vector<BNode>& nodes, int r) : bn(nodes), parent(nodes.size(), -1), root(r)
    {
        for (int i = 0; i < (int)bn.size(); ++i)
            if (bn[i].count == 0) { parent[bn[i].left] = i; parent[bn[i].right] = i; }
    }
    double cost() const            // sum of the internal nodes' areas (the part of the SAH the topology decides)
    {
        double c = 0;
        for (const BNode& n : bn) if (n.count == 0) c += n.box.half_area();
        return c;
    }
    void refit_up(int i)
    {
        while (i >= 0) {
            Box b = bn[bn[i].left].box; b.grow(bn[bn[i].right].box);
            if (std::memcmp(&b, &bn[i].box, sizeof b) == 0) break;
            bn[i].box = b;
            i = parent[i];
        }
    }
    void reinsert(int x)
    {
        const int p = parent[x];
        if (p < 0) return;
        const int g = parent[p];
        if (g < 0) return;                                   // children of the root stay
        const int s = bn[p].left == x ? bn[p].right : bn[p].left;
        (bn[g].left == p ? bn[g].left : bn[g].right) = s;    // the sibling takes the parent's place
        parent[s] = g;
        refit_up(g);
        const Box bx = bn[x].box;
        const float ax = bx.half_area();
        float bestCost = std::numeric_limits<float>::infinity(); int best = s;
        heap.clear();
        heap.push_back({ 0.f, root });
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const Cand c = heap.back(); heap.pop_back();
            if (c.induced + ax >= bestCost) break;           // nothing below can beat the best position
            Box m = bn[c.node].box; m.grow(bx);
            const float total = c.induced + m.half_area();
            if (total < bestCost) { bestCost = total; best = c.node; }
            const float below = total - bn[c.node].box.half_area();
            if (bn[c.node].count == 0 && below + ax < bestCost) {
                heap.push_back({ below, bn[c.node].left });  std::push_heap(heap.begin(), heap.end());
                heap.push_back({ below, bn[c.node].right }); std::push_heap(heap.begin(), heap.end());
            }
        }
        const int y = best, py = parent[y];
        bn[p].left = y; bn[p].right = x; bn[p].count = 0;
        parent[y] = p; parent[x] = p; parent[p] = py;
        if (py >= 0) (bn[py].left == y ? bn[py].left : bn[py].right) = p; else root = p;
        Box b = bn[y].box; b.grow(bx); bn[p].box = b;
        refit_up(py);
    }
    void pass()
    {
        std::vector<int> order;
        order.reserve(bn.size());
        for (int i = 0; i < (int)bn.size(); ++i) if (parent[i] >= 0 && parent[parent[i]] >= 0) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bn[a].box.half_area() > bn[b].box.half_area(); });
        for (int x : order) reinsert(x);
    }
};

// Widen a child box so the slab test stays conservative under float rounding.  The kernels evaluate
// t = plane*inv - (o*inv) with one FMA: the absolute error is about (2|o| + |plane|) * 2^-24 in space, where the ray
// origin o is the camera or a surface point.  G bounds both (scene coordinates and camera position), so a floor of
// 2e-6*G is >10x that error; the relative term covers coordinates larger than the floor's scale.
void pad_box(const Box& b, float G, float* mn, float* mx)
{
    for (int a = 0; a < 3; ++a) {
        float m = std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a]));
        float e = 3e-5f * m + 2e-6f * G + 1e-30f;
        mn[a] = b.mn[a] - e;
        mx[a] = b.mx[a] + e;
    }
}

} // namespace

// Binned-SAH binary tree over n boxes (lo.xyz, hi.xyz), every box a leaf of its own — the top of the device builder's tree over the
// clusters its bottom-up rounds have formed (rt_bvh_gpu.hpp): out[k] = (left, right), a child >= 0 is box index, a child < 0 is ~(index
// into out); out[0] is the root (n >= 2).  Same split search as `build`, large-box isolation included; weights = one per box.
void build_over_boxes(const float* boxes, uint32_t n, const Tuning& tuning, std::vector<int32_t>& out)
{
    out.clear();
    if (n < 2) return;
    Builder B; B.pos = nullptr; B.stride = 0; B.n = n;
    B.max_leaf = 1;
    B.g_bins = std::min(std::max(tuning.bins, 2), 128);
    B.g_cost_exp = (float)std::min(std::max(tuning.cost_exp_percent, 10), 300) / 100.0f;
    B.tbox.resize(n); B.cent.resize(3 * (size_t)n); B.idx.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        Box b;
        for (int a = 0; a < 3; ++a) { b.mn[a] = boxes[6 * (size_t)i + a]; b.mx[a] = boxes[6 * (size_t)i + 3 + a]; }
        B.tbox[i] = b;
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * b.mn[a] + 0.5f * b.mx[a];
            B.cent[3 * (size_t)i + a] = (c == c && std::fabs(c) < 3.0e38f) ? c : 0.0f;      // (an empty or NaN box still gets a place)
        }
        B.idx[i] = i;
    }
    B.bn.reserve(2 * (size_t)n);
    const int root = B.build_range(0, n, 0);
    // internal nodes in pre-order: bn[root] first
    std::vector<int32_t> slot(B.bn.size(), -1);
    int32_t next = 0;
    for (size_t i = 0; i < B.bn.size(); ++i) if (B.bn[i].count == 0) slot[i] = next++;
    out.assign(2 * (size_t)next, 0);
    auto ref = [&](int node) -> int32_t { return B.bn[node].count ? (int32_t)B.idx[B.bn[node].first] : ~slot[node]; };
    for (size_t i = 0; i < B.bn.size(); ++i)
        if (B.bn[i].count == 0) { out[2 * (size_t)slot[i]] = ref(B.bn[i].left); out[2 * (size_t)slot[i] + 1] = ref(B.bn[i].right); }
    (void)root;          // (build_range numbers the root 0, so slot[root] = 0)
}

void build(const float* tri_pos, size_t stride_floats, uint32_t n_tris, float origin_magnitude, const Tuning& tuning, Bvh& out)
{
    const int g_reinsert_passes = std::min(std::max(tuning.reinsert_passes, 0), 16);
    const int max_leaf = tuning.max_leaf;
    out.nodes.clear(); out.order.clear(); out.maxStack = 0; out.depth = 0; out.levelStart.clear();
    if (n_tris == 0) return;

    Builder B; B.pos = tri_pos; B.stride = stride_floats; B.n = n_tris;
    const int B_max_leaf = std::min(std::max(max_leaf, 1), kMaxLeaf);
    B.max_leaf = tuning.collapse_dp == 1 ? 1 : B_max_leaf;          // (cost-driven collapse 1: the dynamic programme forms the leaves; 2: the split search does, as for the greedy collapse)
    B.g_bins = std::min(std::max(tuning.bins, 2), 128);
    B.g_cost_exp = (float)std::min(std::max(tuning.cost_exp_percent, 10), 300) / 100.0f;
    B.tbox.resize(n_tris); B.cent.resize(3 * (size_t)n_tris); B.idx.resize(n_tris);
    float G = origin_magnitude;
    for (uint32_t t = 0; t < n_tris; ++t) {
        const float* p = tri_pos + (size_t)t * stride_floats;
        Box b; b.reset(); b.grow(p); b.grow(p + 3); b.grow(p + 6);
        B.tbox[t] = b;
        for (int a = 0; a < 3; ++a) {
            B.cent[3 * (size_t)t + a] = 0.5f * (b.mn[a] + b.mx[a]);
            G = std::max(G, std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a])));
        }
        B.idx[t] = t;
    }
    B.bn.reserve(2 * (size_t)n_tris + 16);
    int root = B.build_range(0, n_tris, 0);
    if (g_reinsert_passes > 0 && B.bn.size() > 7) {
        Reinserter R(B.bn, root);
        for (int it = 0; it < g_reinsert_passes; ++it) {
            const double before = R.cost();
            R.pass();
            if (!(R.cost() < 0.995 * before)) break;
        }
        root = R.root;
        // depth of the rewired tree
        std::vector<std::pair<int, int>> st; st.push_back({ root, 0 }); B.depth = 0;
        while (!st.empty()) {
            auto [n, d] = st.back(); st.pop_back();
            B.depth = std::max(B.depth, d);
            if (B.bn[n].count == 0) { st.push_back({ B.bn[n].left, d + 1 }); st.push_back({ B.bn[n].right, d + 1 }); }
        }
    }
    if (g_reinsert_passes > 0 && tuning.collapse_dp != 0 && !B.bn.empty()) {
        // the cost-driven collapse makes leaves of whole binary subtrees and takes their triangles as ONE range of the index array —
        // true for the tree the split search built, not after the insertion passes have moved subtrees around: put the index array
        // back into the tree's depth-first order (found by the randomised differential test: three triangles in no leaf, three twice)
        std::vector<uint32_t> nidx; nidx.reserve(B.idx.size());
        std::vector<int> st; st.push_back(root);
        while (!st.empty()) {
            const int n = st.back(); st.pop_back();
            BNode& N = B.bn[n];
            if (N.count) {
                const uint32_t nf = (uint32_t)nidx.size();
                for (uint32_t i = N.first; i < N.first + N.count; ++i) nidx.push_back(B.idx[i]);
                N.first = nf;
            } else { st.push_back(N.right); st.push_back(N.left); }
        }
        B.idx.swap(nidx);
    }
    out.order = B.idx;
    out.magnitude = G;
    out.depth = B.depth;

    // ---- which binary nodes become children of which 4-wide node: cost-driven collapse (after Ylitie, Karras, Laine, "Efficient
    // incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017, section 4.1; width 4, leaves of <= max_leaf triangles).
    // A node step tests four slots whether they are used or not, so an empty slot is wasted work: the greedy "open the largest child"
    // collapse left 3.1 of 4 slots used on the 100k-triangle workload.  Here the binary tree goes down to single triangles and a
    // dynamic programme picks, per binary node n and i = 1..4,
    //     C(n, i) = the cheapest way to represent n's subtree by at most i children of a wide node
    //     C(n, 1) = min( leaf: area(n) * triangles * Ct  [triangles <= max_leaf],   wide node: area(n) * Cn + D(n, 4) )
    //     C(n, i) = min( C(n, i - 1),  D(n, i) ),    D(n, j) = min over k of C(left, k) + C(right, j - k)
    // (expected cost of a random ray: every visited wide node costs a node step Cn, every visited leaf its triangle tests Ct, both
    // in proportion to their box area).  Leaves may split into smaller ones where a free slot makes that cheaper.
    const bool dp = tuning.collapse_dp != 0;
    const float Cn = (float)std::min(std::max(tuning.node_cost_percent, 1), 10000) / 100.0f, Ct = 1.0f;
    std::vector<double> C;              // [node][4]
    std::vector<uint8_t> pick;          // [node][4]: [0] 1 = leaf / 0 = wide node; [i >= 1] for at most i + 1 children: 0 = same as i, else k = children given to the left
    std::vector<uint8_t> pick4;         // [node]: k of D(n, 4) (children of the wide node rooted at n that come from the left subtree)
    std::vector<uint32_t> ntri;
    if (dp) {
        const size_t nb = B.bn.size();
        C.assign(4 * nb, 0.0); pick.assign(4 * nb, 0); pick4.assign(nb, 0); ntri.assign(nb, 0);
        // children were created after their parent unless the tree was rewired: explicit post-order
        std::vector<int> postorder; postorder.reserve(nb);
        { std::vector<int> st; st.push_back(root);
          while (!st.empty()) { int n = st.back(); st.pop_back(); postorder.push_back(n);
                                if (B.bn[n].count == 0) { st.push_back(B.bn[n].left); st.push_back(B.bn[n].right); } } }
        const double INF = std::numeric_limits<double>::infinity();
        for (size_t q = postorder.size(); q-- > 0;) {
            const int n = postorder[q];
            const BNode& N = B.bn[n];
            // (a box with a NaN or an infinity in it — degenerate input — must not poison the comparisons: its area counts as 0 or as huge)
            const float Af = N.box.half_area();
            const double A = Af == Af ? std::min((double)Af, 1.0e30) : 0.0;
            double* c = &C[4 * (size_t)n]; uint8_t* pk = &pick[4 * (size_t)n];
            if (N.count > 0) {
                ntri[n] = N.count;
                for (int i = 0; i < 4; ++i) c[i] = A * (double)N.count * Ct;
                pk[0] = 1;
                continue;
            }
            const int l = N.left, r = N.right;
            ntri[n] = ntri[l] + ntri[r];
            const double* cl = &C[4 * (size_t)l]; const double* cr = &C[4 * (size_t)r];
            double D[5]; uint8_t Dk[5];
            for (int j = 2; j <= 4; ++j) {
                D[j] = INF; Dk[j] = 1;
                for (int k = 1; k < j; ++k) { const double v = cl[k - 1] + cr[j - k - 1]; if (v < D[j]) { D[j] = v; Dk[j] = (uint8_t)k; } }
            }
            const double c_wide = A * Cn + D[4];
            const bool may_leaf = ntri[n] <= (uint32_t)B_max_leaf;
            const double c_leaf = may_leaf ? A * (double)ntri[n] * Ct : INF;
            pick4[n] = Dk[4];
            if (may_leaf && c_leaf <= c_wide) { c[0] = c_leaf; pk[0] = 1; } else { c[0] = c_wide; pk[0] = 0; }
            for (int i = 2; i <= 4; ++i) {
                if (D[i] < c[i - 2]) { c[i - 1] = D[i]; pk[i - 1] = Dk[i]; } else { c[i - 1] = c[i - 2]; pk[i - 1] = 0; }
            }
        }
    }
    // the children (binary nodes) of the wide node rooted at binary node n
    struct Kid { int bnode; bool leaf; };
    auto kids_of = [&](int n, Kid* out_k) -> int {
        int nk = 0;
        struct Item { int node, i; };           // represent `node` by at most i children
        Item st[16]; int sp = 0;
        const int k4 = pick4[n];
        st[sp++] = { B.bn[n].right, 4 - k4 }; st[sp++] = { B.bn[n].left, k4 };
        while (sp > 0) {
            Item it = st[--sp];
            while (it.i > 1 && pick[4 * (size_t)it.node + it.i - 1] == 0) --it.i;
            if (it.i == 1) { out_k[nk++] = { it.node, pick[4 * (size_t)it.node] != 0 }; continue; }
            const int k = pick[4 * (size_t)it.node + it.i - 1];
            st[sp++] = { B.bn[it.node].right, it.i - k }; st[sp++] = { B.bn[it.node].left, k };
        }
        return nk;
    };
    // first triangle (BVH order) of a binary subtree: its leftmost leaf's
    auto first_tri = [&](int n) -> uint32_t { while (B.bn[n].count == 0) n = B.bn[n].left; return B.bn[n].first; };

    // ---- collapse to 4-wide, breadth-first ----
    struct Pending { int bnode; uint32_t slot; int level; };
    std::vector<Node4>& N = out.nodes;
    std::queue<Pending> q;
    auto new_node = [&]() -> uint32_t {
        Node4 z; std::memset(&z, 0, sizeof z);
        for (int k = 0; k < 4; ++k) {
            z.child[k] = kEmpty;
            // an empty slot can never be entered: near planes at +inf, far planes at -inf for either ray direction
            z.minx[k] = z.miny[k] = z.minz[k] = std::numeric_limits<float>::infinity();
            z.maxx[k] = z.maxy[k] = z.maxz[k] = -std::numeric_limits<float>::infinity();
        }
        N.push_back(z);
        return (uint32_t)N.size() - 1u;
    };
    q.push({ root, new_node(), 0 });
    out.levelStart.push_back(0);
    while (!q.empty()) {
        Pending pd = q.front(); q.pop();
        if ((int)out.levelStart.size() <= pd.level) out.levelStart.push_back(pd.slot);
        int kids[4]; bool kid_leaf[4]; int nk = 0;
        const BNode& r = B.bn[pd.bnode];
        if (dp) {
            if (r.count > 0 || pick[4 * (size_t)pd.bnode] != 0) { kids[nk] = pd.bnode; kid_leaf[nk++] = true; }     // the whole mesh fits one leaf
            else { Kid kk[4]; nk = kids_of(pd.bnode, kk); for (int k = 0; k < nk; ++k) { kids[k] = kk[k].bnode; kid_leaf[k] = kk[k].leaf; } }
        } else {
        if (r.count > 0) { kids[nk++] = pd.bnode; }         // the whole mesh fits one leaf
        else { kids[nk++] = r.left; kids[nk++] = r.right; }
        while (nk < 4) {                                    // open the internal child with the largest area
            int pick = -1; float pa = -1.f;
            for (int k = 0; k < nk; ++k)
                if (B.bn[kids[k]].count == 0) {
                    float ar = B.bn[kids[k]].box.half_area();
                    if (ar > pa) { pa = ar; pick = k; }
                }
            if (pick < 0) break;
            int c = kids[pick];
            kids[pick] = B.bn[c].left;
            kids[nk++] = B.bn[c].right;
        }
        for (int k = 0; k < nk; ++k) kid_leaf[k] = B.bn[kids[k]].count > 0;
        }
        for (int k = 0; k < nk; ++k) {
            const BNode& c = B.bn[kids[k]];
            float mn[3], mx[3];
            pad_box(c.box, G, mn, mx);
            uint32_t ref;
            if (kid_leaf[k]) ref = dp ? make_leaf(first_tri(kids[k]), ntri[kids[k]]) : make_leaf(c.first, c.count);
            else { ref = new_node(); q.push({ kids[k], ref, pd.level + 1 }); }
            Node4& me = N[pd.slot];                          // (re-fetch: new_node may reallocate)
            me.minx[k] = mn[0]; me.miny[k] = mn[1]; me.minz[k] = mn[2];
            me.maxx[k] = mx[0]; me.maxy[k] = mx[1]; me.maxz[k] = mx[2];
            me.child[k] = ref;
        }
        N[pd.slot].meta[0] = (uint32_t)nk;
    }

    out.levelStart.push_back((uint32_t)N.size());

    // ---- worst-case traversal stack: at a node with k used slots the nearest child becomes current and up
    // to k-1 are pushed; nodes are in BFS order so children have larger indices -> sweep backwards.
    std::vector<int> need(N.size(), 0);
    for (size_t i = N.size(); i-- > 0;) {
        int k = (int)N[i].meta[0], deepest = 0;
        for (int s = 0; s < k; ++s) {
            uint32_t c = N[i].child[s];
            if (!(c & kLeafBit)) deepest = std::max(deepest, need[c]);
        }
        need[i] = (k - 1) + deepest;
    }
    out.maxStack = std::max(1, need[0]);
}

} // namespace rtbvh
