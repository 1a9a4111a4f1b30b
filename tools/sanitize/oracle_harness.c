#include "../../oracle/rt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static void* slurp(const char* fmt, int seed, size_t* n) { char p[256]; snprintf(p, sizeof p, fmt, seed); FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); *n = ftell(f); rewind(f); void* b = malloc(*n ? *n : 1); if (fread(b, 1, *n, f) != *n) abort(); fclose(f); return b; }
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        int seed = atoi(argv[i]); size_t np, ns, nt, nm;
        rt_params* p = slurp("/tmp/rtx_sanitize/orc_p%d.bin", seed, &np); rt_sphere* s = slurp("/tmp/rtx_sanitize/orc_s%d.bin", seed, &ns);
        rt_triangle* t = slurp("/tmp/rtx_sanitize/orc_t%d.bin", seed, &nt); rt_meshinfo* m = slurp("/tmp/rtx_sanitize/orc_m%d.bin", seed, &nm);
        size_t px = (size_t)p->width * p->height; float* a = malloc(px * 16 + 16); float* b = malloc(px * 16 + 16); orc_counts ca, cb;
        orc_set_accel(0); orc_render_frame(p, s, ns / sizeof *s, t, nt / sizeof *t, m, nm / sizeof *m, 1, 0, 0, p->width, p->height, a, 4, &ca);
        orc_set_accel(1); orc_render_frame(p, s, ns / sizeof *s, t, nt / sizeof *t, m, nm / sizeof *m, 1, 0, 0, p->width, p->height, b, 4, &cb);
        printf("seed %d: %zu tris, same image %d, rays %llu %llu\n", seed, nt / sizeof *t, memcmp(a, b, px * 16) == 0, (unsigned long long)ca.rays, (unsigned long long)cb.rays);
        free(p); free(s); free(t); free(m); free(a); free(b);
    }
    return 0;
}
