#include "../../ray-tracing-extended_amd/host_cpp/rt_host.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        try {
            rthost::RayTracingManager m = rthost::LoadUnityScene(argv[i], 320, 180);
            m.OnValidate();
            rthost::SceneBuffers b = m.BuildBuffers();
            printf("%s: %zu spheres %zu triangles %zu chunks\n", argv[i], b.spheres.size(), b.triangles.size(), b.meshInfo.size());
        } catch (const std::exception& e) { printf("%s: exception %s\n", argv[i], e.what()); }
    }
    return 0;
}
