#include "../../ray-tracing-extended_amd/csrc/bvh.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<float> v; float x; while (fread(&x, 4, 1, f) == 1) v.push_back(x); fclose(f);
    uint32_t n = v.size() / 9;
    // the API filters triangles with non-finite positions? emulate: pass all
    rtbvh::Bvh b; rtbvh::Tuning t; t.reinsert_passes = argc > 2 ? atoi(argv[2]) : 0;
    rtbvh::build(v.data(), 9, n, 10.0f, t, b);
    printf("tris %u nodes %zu maxStack %d depth %d\n", n, b.nodes.size(), b.maxStack, b.depth);
    return 0;
}
