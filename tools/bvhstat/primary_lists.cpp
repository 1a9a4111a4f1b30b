// tools/bvhstat/primary_lists.cpp — how many pixels of a workload could trace their camera rays from a short per-pixel list of leaves?
// (Round 4, verdict item 1(c): an estimate on the CPU before anything is built into the tracer.)
//   input: 26 floats (camera position 3, camLocalToWorld 16 row-major, viewParams 3, width, height, divergeStrength, defocusStrength) + 9 floats per triangle
//   g++ -O2 -std=c++17 tools/bvhstat/primary_lists.cpp ray-tracing-extended_amd/csrc/bvh.cpp -o tools/bvhstat/primary_lists
// For every 4th pixel in x and y: the four corner rays of the pixel's jitter footprint (widened by 5 %) are traced through the host builder's
// BVH4.  "single" = all four hit the same front-facing triangle with barycentrics >= 0.01: every camera ray of the pixel then hits that
// triangle or something nearer, so its closest hit lies in a leaf whose box meets the footprint's frustum before t_max = the farthest
// corner hit; K = number of such leaves.  "sky" = all four miss and no leaf box meets the frustum at all.  Everything else = "mixed".
#include "../../ray-tracing-extended_amd/csrc/bvh.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct V { double x, y, z; };
static V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V operator*(V a, double s) { return { a.x * s, a.y * s, a.z * s }; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static rtbvh::Bvh bvh; static const float* T;
static bool tri(V o, V d, uint32_t ti, double& t, double& u, double& v)
{
    const float* p = T + 9 * (size_t)bvh.order[ti];
    V A{ p[0], p[1], p[2] }, B{ p[3], p[4], p[5] }, C{ p[6], p[7], p[8] };
    V e1 = B - A, e2 = C - A, n = cross(e1, e2), ao = o - A, dao = cross(ao, d);
    double det = -dot(d, n); if (!(det >= 1e-6)) return false;
    double inv = 1.0 / det; t = dot(ao, n) * inv; u = dot(e2, dao) * inv; v = -dot(e1, dao) * inv;
    return t >= 0 && u >= 0 && v >= 0 && 1 - u - v >= 0;
}
static bool box(const rtbvh::Node4& N, int k, V o, V inv, double tmax, double& tn)
{
    double t0 = (N.minx[k] - o.x) * inv.x, t1 = (N.maxx[k] - o.x) * inv.x; double a = std::min(t0, t1), b = std::max(t0, t1);
    t0 = (N.miny[k] - o.y) * inv.y; t1 = (N.maxy[k] - o.y) * inv.y; a = std::max(a, std::min(t0, t1)); b = std::min(b, std::max(t0, t1));
    t0 = (N.minz[k] - o.z) * inv.z; t1 = (N.maxz[k] - o.z) * inv.z; a = std::max(a, std::min(t0, t1)); b = std::min(b, std::max(t0, t1));
    tn = std::max(a, 0.0); return tn <= std::min(b, tmax);
}
static int trace(V o, V d, double& tbest, double& ub, double& vb)
{
    V inv{ 1 / d.x, 1 / d.y, 1 / d.z }; int best = -1; tbest = 1e300;
    std::vector<uint32_t> st{ 0 };
    while (!st.empty()) {
        uint32_t c = st.back(); st.pop_back();
        if (c & rtbvh::kLeafBit) {
            uint32_t first = (c & 0x7FFFFFFFu) >> 2, cnt = (c & 3) + 1;
            for (uint32_t j = 0; j < cnt; ++j) { double t, u, v; if (tri(o, d, first + j, t, u, v) && t < tbest) { tbest = t; best = (int)(first + j); ub = u; vb = v; } }
            continue;
        }
        const rtbvh::Node4& N = bvh.nodes[c];
        for (int k = 0; k < 4; ++k) { double tn; if (N.child[k] != rtbvh::kEmpty && box(N, k, o, inv, tbest, tn)) st.push_back(N.child[k]); }
    }
    return best;
}
// leaves (and their triangles) whose boxes meet the frustum of the four planes through o (normals pointing out) within distance tmax of o
static void frustum_leaves(V o, const V* n, double tmax, int& leaves, int& tris, int& nodes_visited)
{
    std::vector<uint32_t> st{ 0 }; leaves = tris = nodes_visited = 0;
    while (!st.empty()) {
        uint32_t c = st.back(); st.pop_back();
        if (c & rtbvh::kLeafBit) { ++leaves; tris += (c & 3) + 1; continue; }
        const rtbvh::Node4& N = bvh.nodes[c]; ++nodes_visited;
        for (int k = 0; k < 4; ++k) {
            if (N.child[k] == rtbvh::kEmpty) continue;
            bool out = false;
            for (int q = 0; q < 4 && !out; ++q) {           // the box corner farthest against the plane normal must be inside
                V p{ n[q].x > 0 ? N.minx[k] : N.maxx[k], n[q].y > 0 ? N.miny[k] : N.maxy[k], n[q].z > 0 ? N.minz[k] : N.maxz[k] };
                if (dot(n[q], p - o) > 0) out = true;
            }
            if (!out) {                                      // nearest point of the box to o farther than tmax?
                double dx = std::max({ N.minx[k] - o.x, 0.0, o.x - N.maxx[k] }), dy = std::max({ N.miny[k] - o.y, 0.0, o.y - N.maxy[k] }), dz = std::max({ N.minz[k] - o.z, 0.0, o.z - N.maxz[k] });
                if (std::sqrt(dx * dx + dy * dy + dz * dz) > tmax) out = true;
            }
            if (!out) st.push_back(N.child[k]);
        }
    }
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    std::vector<float> v; float x; while (fread(&x, 4, 1, f) == 1) v.push_back(x); fclose(f);
    const float* h = v.data(); T = v.data() + 26; const uint32_t nt = (uint32_t)((v.size() - 26) / 9);
    const int step = argc > 2 ? atoi(argv[2]) : 4;
    rtbvh::Tuning tu; rtbvh::build(T, 9, nt, 10.0f, tu, bvh);
    V pos{ h[0], h[1], h[2] }; const float* M = h + 3; const double W = h[22], H = h[23], r = h[24] / W * 1.05;
    V right{ M[0], M[4], M[8] }, up{ M[1], M[5], M[9] };
    long n = 0, single = 0, sky = 0, mixed = 0, miss_with_boxes = 0; std::vector<int> hist(34, 0), hist_t(66, 0), hist_sky(34, 0); double nodes_sum = 0;
    for (int y = 0; y < (int)H; y += step) for (int xx = 0; xx < (int)W; xx += step) {
        ++n;
        double lx = ((xx + 0.5) / W - 0.5) * h[19], ly = ((y + 0.5) / H - 0.5) * h[20], lz = h[21];
        V fp{ M[0] * lx + M[1] * ly + M[2] * lz + M[3], M[4] * lx + M[5] * ly + M[6] * lz + M[7], M[8] * lx + M[9] * ly + M[10] * lz + M[11] };
        V dir[4]; int id[4]; double t[4]; bool margin = true;
        for (int c = 0; c < 4; ++c) {
            V tgt = fp + right * ((c & 1) ? r : -r) + up * ((c & 2) ? r : -r);
            dir[c] = tgt - pos; double u, vv; id[c] = trace(pos, dir[c], t[c], u, vv);
            if (id[c] >= 0 && !(u >= 0.01 && vv >= 0.01 && 1 - u - vv >= 0.01)) margin = false;
        }
        // planes through pos: corners in the order (-,-), (+,-), (+,+), (-,+); normal of the side between consecutive corners, pointing out
        const int ord[4] = { 0, 1, 3, 2 }; V nrm[4]; V centre = fp - pos;
        for (int q = 0; q < 4; ++q) { V a = dir[ord[q]], b = dir[ord[(q + 1) & 3]]; V nn = cross(a, b); if (dot(nn, centre) > 0) nn = nn * -1.0; nrm[q] = nn; }
        int leaves, tris, nv;
        if (id[0] >= 0 && id[0] == id[1] && id[1] == id[2] && id[2] == id[3] && margin) {
            double tmax = std::max({ t[0], t[1], t[2], t[3] }) * std::sqrt(std::max({ dot(dir[0], dir[0]), dot(dir[1], dir[1]), dot(dir[2], dir[2]), dot(dir[3], dir[3]) })) * 1.0001;
            frustum_leaves(pos, nrm, tmax, leaves, tris, nv);
            ++single; hist[std::min(leaves, 33)]++; hist_t[std::min(tris, 65)]++; nodes_sum += nv;
        } else if (id[0] < 0 && id[1] < 0 && id[2] < 0 && id[3] < 0) {
            frustum_leaves(pos, nrm, 1e300, leaves, tris, nv);
            if (leaves == 0) ++sky; else { ++miss_with_boxes; ++mixed; hist_sky[std::min(leaves, 33)]++; }
        } else ++mixed;
    }
    printf("%ld pixels sampled (every %d-th): single-triangle footprints %.1f %%, certain sky %.1f %%, the rest %.1f %% (of which all-miss but boxes in the frustum %.1f %%)\n",
           n, step, 100.0 * single / n, 100.0 * sky / n, 100.0 * mixed / n, 100.0 * miss_with_boxes / n);
    printf("single-triangle footprints: leaves in the truncated frustum (K): "); long cum = 0;
    for (int k = 0; k < 34; ++k) { cum += hist[k]; if (hist[k]) printf("%d:%.1f%% ", k, 100.0 * hist[k] / std::max(single, 1L)); } printf("\n");
    printf("all four corners miss, leaves in the (unbounded) frustum: "); { long c = 0; for (int k = 1; k < 34; ++k) { c += hist_sky[k]; if (k <= 8 || k == 33) printf("<=%d:%.1f%% ", k, 100.0 * c / n); } printf("(of all pixels)\n"); }
    long c8 = 0, c4 = 0; for (int k = 0; k <= 8; ++k) { c8 += hist[k]; if (k <= 4) c4 += hist[k]; }
    double mt = 0; for (int k = 0; k < 66; ++k) mt += (double)k * hist_t[k];
    printf("K <= 4: %.1f %% of all pixels, K <= 8: %.1f %%; mean triangles in the list %.1f; frustum traversal visits %.1f nodes per pixel\n",
           100.0 * c4 / n, 100.0 * c8 / n, mt / std::max(single, 1L), nodes_sum / std::max(single, 1L));
    return 0;
}
