// tests/bvh_check.cpp — structural check of the host builder's BVH4 (tests/test_bvh_host_cpu.py compiles it with csrc/bvh.cpp):
// every triangle in exactly one leaf, every finite vertex inside its leaf's box and every box on the way down to it.
//   bvh_check <tris.f32: 9 floats per triangle> <collapse_dp> <max_leaf> <reinsert_passes>
#include "bvh.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); std::vector<float> v; float x; while (fread(&x, 4, 1, f) == 1) v.push_back(x); fclose(f);
    const uint32_t n = (uint32_t)(v.size() / 9);
    rtbvh::Bvh b; rtbvh::Tuning t;
    t.collapse_dp = atoi(argv[2]); t.max_leaf = atoi(argv[3]); t.reinsert_passes = atoi(argv[4]);
    rtbvh::build(v.data(), 9, n, 10.0f, t, b);
    std::vector<int> seen(n, 0); int bad = 0;
    std::function<void(uint32_t, const float*, const float*)> walk = [&](uint32_t ni, const float* pmn, const float* pmx) {
        const rtbvh::Node4& N = b.nodes[ni];
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = N.child[k]; if (c == rtbvh::kEmpty) continue;
            const float mn[3] = { N.minx[k], N.miny[k], N.minz[k] }, mx[3] = { N.maxx[k], N.maxy[k], N.maxz[k] };
            if (c & rtbvh::kLeafBit) {
                const uint32_t first = (c & 0x7FFFFFFFu) >> 2, cnt = (c & 3u) + 1;
                for (uint32_t i = first; i < first + cnt; ++i) {
                    if (i >= b.order.size()) { printf("leaf index %u out of range\n", i); ++bad; continue; }
                    const uint32_t tri = b.order[i]; seen[tri]++;
                    for (int vtx = 0; vtx < 3; ++vtx) for (int a = 0; a < 3; ++a) {
                        const float p = v[9 * (size_t)tri + 3 * vtx + a];
                        if (p == p && std::fabs(p) < 1e30f && !(p >= mn[a] && p <= mx[a])) { if (bad < 10) printf("tri %u vertex outside its leaf box (node %u slot %d axis %d: %g not in [%g,%g])\n", tri, ni, k, a, p, mn[a], mx[a]); ++bad; }
                    }
                }
            } else walk(c, mn, mx);
        }
    };
    if (!b.nodes.empty()) walk(0, nullptr, nullptr);
    int miss = 0, dup = 0; for (uint32_t i = 0; i < n; ++i) { if (!seen[i]) { if (miss < 10) printf("triangle %u in no leaf\n", i); ++miss; } if (seen[i] > 1) ++dup; }
    printf("%u triangles, %zu nodes: %d missing, %d duplicated, %d containment errors\n", n, b.nodes.size(), miss, dup, bad);
    return 0;
}
