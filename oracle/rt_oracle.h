/*
 * rt_oracle.h — API of the CPU oracle (TEST INFRASTRUCTURE: see the header of rt_oracle.c).
 * Uses the product's public POD layouts from include/rt.h so that oracle and HIP path are fed the very
 * same bytes; nothing in the product includes or links this.
 */
#ifndef RT_ORACLE_H_
#define RT_ORACLE_H_
#include "../include/rt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_counts {
    uint64_t rays;          /* CalculateRayCollision invocations (RayTracing.shader:256)   */
    uint64_t sphereTests;   /* RaySphere calls (:266)                                      */
    uint64_t boxTests;      /* RayBoundingBox calls (:279), FLAT_CHUNKS mode only          */
    uint64_t triTests;      /* RayTriangle calls (:286)                                    */
    uint64_t hits;
    int32_t  threads;       /* threads actually used                                       */
    int32_t  _pad;
} orc_counts;

/* Render pixels [x0,x1) x [y0,y1) of frame `frame` of the full params->width x params->height image into
 * out_rgba (dense crop, row-major, row y0 first, 4 floats per pixel).  params->intersectMode selects
 * RT_INTERSECT_FLAT_CHUNKS (the literal reference loop) or RT_INTERSECT_BRUTE (no chunk cull).
 * nthreads <= 0: all OpenMP threads.  Returns 0 on success. */
int orc_render_frame(const rt_params* params,
                     const rt_sphere* spheres, int ns,
                     const rt_triangle* tris, int nt,
                     const rt_meshinfo* meshinfo, int nm,
                     int frame, int x0, int y0, int x1, int y1,
                     float* out_rgba, int nthreads, orc_counts* counts);

/* on != 0: later orc_render_frame calls find triangles through the oracle's own search tree instead of the linear
 * loop — same result bit for bit (tests/test_oracle_cpu.py), fast enough for full-size images.  In that mode
 * boxTests / triTests count the tree's work, not the reference loop's. */
void orc_set_accel(int on);

/* Accumulate.shader:43-54 applied in place to `accum` (n_floats = pixels*4). */
void orc_accumulate(float* accum, const float* cur, size_t n_floats, int frame);

/* linear RGBA32F -> sRGB RGBA8 (the display blit after the path, RayTracingManager.cs:84) */
void orc_display_srgb8(const float* rgba, uint32_t* out, size_t n_pixels);

/* pieces exported for known-answer tests */
uint32_t orc_next_random(uint32_t* state);
float    orc_random_value(uint32_t* state);
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int      orc_philox_substreams(int rays_per_pixel);      /* sub-streams of the Philox mode's estimator sum: 16 / 4 / 1 */
float om_sin(float), om_cos(float), om_log(float), om_exp2(float), om_pow(float, float);
float om_min(float, float), om_max(float, float);
int   orc_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
