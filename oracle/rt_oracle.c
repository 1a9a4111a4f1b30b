/*
 * rt_oracle.c — CPU restatement of the reference's per-pixel path tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle and the timed CPU baseline.  It is never linked into, imported by or
 * called from the product (ray-tracing-extended_amd/, include/): only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * What it restates (all citations relative to /root/reference):
 *   Assets/Scripts/Shaders/RayTracing.shader:35,38-58,120-389   the tracer (a Unity HLSL fragment shader)
 *   Assets/Scripts/Shaders/Accumulate.shader:43-54              the progressive running average
 * The reference has no CPU tracer and cannot be built here (HLSL inside Unity ShaderLab, driven by C# that
 * needs UnityEngine; no dotnet/mono/Unity/dxc in the image) — DESIGN.md "Oracle".
 *
 * PINNING STATUS.  Integer PCG stream: pinned by known-answer vectors (tests/golden/pcg_kat.json, derived
 * from RayTracing.shader:193-199 by exact integer arithmetic).  Buffer layouts: pinned by the reference's
 * struct definitions (64/80/72/96 bytes).  Scene inputs: pinned by the reference's .unity files.
 * Float level: PARITY UNPINNED — the HLSL intrinsics (normalize, lerp, reflect, smoothstep, pow, log, cos,
 * sin, min/max NaN rules) are evaluated by Unity's shader compiler and a GPU driver, neither of which
 * exists here, and the reference holds no golden images.  The frozen choices are listed below and in
 * DESIGN.md; the HIP kernels must reproduce THIS file bit for bit.
 *
 * Frozen float semantics (float32 everywhere, no FMA contraction: build with -ffp-contract=off):
 *   dot(a,b)        = (ax*bx + ay*by) + az*bz                (left to right)
 *   cross(a,b)      = (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
 *   normalize(v)    = v / sqrt(dot(v,v))                     (IEEE sqrt, three IEEE divisions)
 *   lerp(a,b,t)     = a + t*(b-a)
 *   reflect(i,n)    = i - (2*dot(i,n))*n
 *   min/max         = return the non-NaN operand (formulas om_min/om_max below)
 *   saturate(x)     = min(max(x,0),1)
 *   smoothstep      = t = saturate((x-e0)/(e1-e0)); t*t*(3-2*t)
 *   pow(x,y)        = x==0 ? 0 : exp2(y*log2(x));  log2(x) = log(x)*1.44269504
 *   sin/cos/log/exp2= the polynomial kernels om_* below (Cody-Waite reduction + Cephes single-precision
 *                     minimax coefficients); they use only + - * floor and integer bit operations, so
 *                     g++ and hipcc produce identical bits.
 *   uint -> float   = round-to-nearest-even;  RandomValue = (float)r * 2^-32   (shader :203: the literal
 *                     4294967295.0 is a float32 constant, i.e. 2^32)
 *   pixel centre uv = ((x+0.5)/W, (y+0.5)/H), row y=0 at the bottom (Unity uv origin)
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* bit casts                                                                                         */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------------ */
/* frozen intrinsics                                                                                 */

float om_min(float a, float b) { return (a < b) ? a : ((b != b) ? a : b); }
float om_max(float a, float b) { return (a > b) ? a : ((b != b) ? a : b); }
static inline float om_saturate(float x) { return om_min(om_max(x, 0.0f), 1.0f); }

/* Quadrant reduction shared by sin and cos.  pi/2 = C1 + C2 + C3 (Cody-Waite; C1 has 8 significant bits
 * so q*C1 is exact for |q| < 2^16). */
static inline float om_reduce(float x, int* quadrant)
{
    const float TWO_OVER_PI = 0.636619772f;
    const float C1 = 1.5703125f;
    const float C2 = 4.837512969970703125e-4f;
    const float C3 = 7.54978995489188216e-8f;
    float q = floorf(x * TWO_OVER_PI + 0.5f);
    float r = ((x - q * C1) - q * C2) - q * C3;
    *quadrant = (int)q & 3;
    return r;
}
static inline float om_sin_poly(float r)
{
    float z = r * r;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}
static inline float om_cos_poly(float r)
{
    float z = r * r;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
           - 0.5f * z + 1.0f;
}
/* ORC_INTRINSICS_LIBM (a second build, oracle/Makefile target librt_oracle_libm.so — never the parity oracle): the transcendental
 * intrinsics come from the host's libm instead of the frozen polynomial kernels.  Unity's shader compiler and the GPU driver evaluate
 * sin / cos / log / pow with their own (unknown) approximations; rendering the reference's scenes with two different, both accurate,
 * implementations shows how much of a converged image depends on that choice (tools/oracle_sensitivity.py). */
#ifdef ORC_INTRINSICS_LIBM
float om_sin(float x) { return sinf(x); }
float om_cos(float x) { return cosf(x); }
float om_log(float x) { return logf(x); }
float om_exp2(float x) { return exp2f(x); }
float om_pow(float x, float y) { return x == 0.0f ? 0.0f : powf(x, y); }
#define om_sin om_sin_frozen_unused
#define om_cos om_cos_frozen_unused
#define om_log om_log_frozen_unused
#define om_exp2 om_exp2_frozen_unused
#define om_pow om_pow_frozen_unused
#endif
float om_sin(float x)
{
    int n; float r = om_reduce(x, &n);
    float s = om_sin_poly(r), c = om_cos_poly(r);
    float v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}
float om_cos(float x)
{
    int n; float r = om_reduce(x, &n);
    float s = om_sin_poly(r), c = om_cos_poly(r);
    float v = (n & 1) ? s : c;
    return ((n + 1) & 2) ? -v : v;
}

/* natural log, x >= 0 (Cephes logf scheme).  log(0) = -inf, log(inf) = inf, log(NaN) = NaN, log(x<0) = NaN. */
float om_log(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0xff800000u);
    uint32_t bits = f2u(x);
    if (bits == 0x7f800000u) return x;
    int e = 0;
    if (bits < 0x00800000u) { x = x * 8388608.0f; bits = f2u(x); e = -23; }   /* denormal: scale by 2^23 (exact) */
    e += (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);                         /* [1,2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f
                   - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f - 1.6668057665e-1f) * f
                + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = f + y;
    r = r + 0.693359375f * fe;
    return r;
}
static inline float om_log2(float x) { return om_log(x) * 1.44269504f; }

/* 2^x */
float om_exp2(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (x < -150.0f) return 0.0f;
    float k = floorf(x + 0.5f);
    float f = x - k;                                                           /* [-0.5, 0.5] */
    float p = ((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
                + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f;
    float r = p * f + 1.0f;
    int ki = (int)k;
    int k1 = ki >> 1, k2 = ki - k1;                                            /* both within [-75, 64] */
    r = r * u2f((uint32_t)(k1 + 127) << 23);
    r = r * u2f((uint32_t)(k2 + 127) << 23);
    return r;
}
float om_pow(float x, float y)
{
    if (x == 0.0f) return 0.0f;                                                /* pow(0, y>0) = 0 */
    return om_exp2(y * om_log2(x));
}
#ifdef ORC_INTRINSICS_LIBM
#undef om_sin
#undef om_cos
#undef om_log
#undef om_exp2
#undef om_pow
#endif
static inline float om_smoothstep(float e0, float e1, float x)
{
    float t = om_saturate((x - e0) / (e1 - e0));
    return t * t * (3.0f - 2.0f * t);
}

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v_cross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v_normalize(v3 a)
{
    float len = sqrtf(v_dot(a, a));
    return V(a.x / len, a.y / len, a.z / len);
}
static inline v3 v_lerp(v3 a, v3 b, float t)
{
    return V(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
static inline v3 v_reflect(v3 i, v3 n)
{
    float k = 2.0f * v_dot(i, n);
    return V(i.x - k * n.x, i.y - k * n.y, i.z - k * n.z);
}
static inline v3 v_load(const float* p) { return V(p[0], p[1], p[2]); }

/* ------------------------------------------------------------------------------------------------ */
/* RNG — RayTracing.shader:193-204                                                                  */

uint32_t orc_next_random(uint32_t* state)
{
    *state = *state * 747796405u + 2891336453u;
    uint32_t s = *state;
    uint32_t result = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
float orc_random_value(uint32_t* state)
{
    return (float)orc_next_random(state) * 2.3283064365386963e-10f;           /* r / 2^32 (shader :203) */
}
/* Counter-based alternative (rngMode = RT_RNG_PHILOX; not a reference mode — the north-star's perf mode): Philox4x32-10
 * (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), key = (pixelIndex, Frame), counter = (block, sample, 0, 0).
 * The reference chains one PCG state through every sample and bounce of a pixel (RayTracing.shader:362,374-385), which forbids
 * spreading a pixel's samples over lanes; here every draw is addressed by what it is for, so samples (and bounces) are independent:
 *   block 0                 the sample's camera ray: words 0..3 = the four draws of frag :377,380 in the shader's order
 *                           (defocus angle, defocus radius, diverge angle, diverge radius)
 *   blocks 1 + 2b, 2 + 2b   the hit at loop index `bounce` = b of Trace (:305): the eight draws of :325-339 in the shader's order
 *                           (isSpecular; theta, rho of the x, y, z normal deviates; Russian roulette) = words 0..3 of the first block,
 *                           then words 0..3 of the second.  A bounce that draws nothing (InvisibleLightSource :318-322, a miss) leaves
 *                           its blocks unused.
 * The estimator's sum over the NumRaysPerPixel samples (:384) is a fixed tree instead of a left-to-right chain, so that a wavefront can
 * evaluate it in parallel: sample s goes to sub-stream s mod S, S = orc_philox_substreams(NumRaysPerPixel) = 16 / 4 / 1; a sub-stream
 * adds its samples in increasing order, starting from 0; the S sub-sums are combined pairwise — (k, k + 1) for even k, then (k, k + 2)
 * for k = 0 mod 4, ... — and the root is divided by NumRaysPerPixel (:387).  Scatter, Russian roulette and everything else are the
 * reference's.  Different noise than the PCG mode, same expectation. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { int mode; uint32_t state /* PCG */, key[2], sample, block, n, cache[4]; } orng;

int orc_philox_substreams(int rays_per_pixel) { return rays_per_pixel >= 16 ? 16 : rays_per_pixel >= 4 ? 4 : 1; }

/* Philox mode: the draws that follow are words 0, 1, 2, ... of blocks `block`, `block` + 1, ... of the current sample */
static inline void rng_scope(orng* g, uint32_t block) { g->block = block; g->n = 0; }

static float rnd(orng* g)
{
    if (g->mode == RT_RNG_PCG) return orc_random_value(&g->state);
    if ((g->n & 3u) == 0u) {
        uint32_t ctr[4] = { g->block + (g->n >> 2), g->sample, 0u, 0u };
        orc_philox4x32_10(ctr, g->key, g->cache);
    }
    uint32_t r = g->cache[g->n & 3u];
    g->n++;
    return (float)r * 2.3283064365386963e-10f;
}

/* RayTracing.shader:207-213 */
static float random_normal(orng* state)
{
    const float TWO_PI = 2.0f * 3.1415926f;
    float theta = TWO_PI * rnd(state);
    float rho = sqrtf(-2.0f * om_log(rnd(state)));
    return rho * om_cos(theta);
}
/* RayTracing.shader:216-223 */
static v3 random_direction(orng* state)
{
    float x = random_normal(state);
    float y = random_normal(state);
    float z = random_normal(state);
    return v_normalize(V(x, y, z));
}
/* RayTracing.shader:225-230 with PI = 3.1415 (:35) */
static void random_point_in_circle(orng* state, float* px, float* py)
{
    const float PI = 3.1415f;
    float angle = rnd(state) * 2.0f * PI;
    float cx = om_cos(angle), cy = om_sin(angle);
    float s = sqrtf(rnd(state));
    *px = cx * s;
    *py = cy * s;
}

/* ------------------------------------------------------------------------------------------------ */
/* intersection — RayTracing.shader:120-187                                                          */

typedef struct {
    int   didHit;
    float dst;
    v3    hitPoint, normal;
    const rt_material* material;
} hit_t;

/* RaySphere :120-146 */
static inline int ray_sphere(v3 o, v3 d, v3 centre, float radius, float* dst_out)
{
    v3 oc = v_sub(o, centre);
    float a = v_dot(d, d);
    float b = 2.0f * v_dot(oc, d);
    float c = v_dot(oc, oc) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc >= 0.0f) {
        float dst = (-b - sqrtf(disc)) / (2.0f * a);
        if (dst >= 0.0f) { *dst_out = dst; return 1; }
    }
    return 0;
}

/* RayTriangle :150-174 */
static inline int ray_triangle(v3 o, v3 d, const rt_triangle* t, float* dst_out, float* u_out, float* v_out, float* w_out)
{
    v3 A = v_load(t->posA);
    v3 eAB = v_sub(v_load(t->posB), A);
    v3 eAC = v_sub(v_load(t->posC), A);
    v3 n = v_cross(eAB, eAC);
    v3 ao = v_sub(o, A);
    v3 dao = v_cross(ao, d);
    float det = -v_dot(d, n);
    float inv = 1.0f / det;
    float dst = v_dot(ao, n) * inv;
    float u = v_dot(eAC, dao) * inv;
    float v = -v_dot(eAB, dao) * inv;
    float w = 1.0f - u - v;
    *dst_out = dst; *u_out = u; *v_out = v; *w_out = w;
    return det >= 1e-6f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
}

/* RayBoundingBox :177-187 */
static inline int ray_bounding_box(v3 o, v3 d, const float* bmin, const float* bmax)
{
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    float tminx = (bmin[0] - o.x) * ix, tminy = (bmin[1] - o.y) * iy, tminz = (bmin[2] - o.z) * iz;
    float tmaxx = (bmax[0] - o.x) * ix, tmaxy = (bmax[1] - o.y) * iy, tmaxz = (bmax[2] - o.z) * iz;
    float t1x = om_min(tminx, tmaxx), t1y = om_min(tminy, tmaxy), t1z = om_min(tminz, tmaxz);
    float t2x = om_max(tminx, tmaxx), t2y = om_max(tminy, tmaxy), t2z = om_max(tminz, tmaxz);
    float tNear = om_max(om_max(t1x, t1y), t1z);
    float tFar  = om_min(om_min(t2x, t2y), t2z);
    return tNear <= tFar;
}

typedef struct {
    const rt_params*   p;
    const rt_sphere*   spheres;  int ns;
    const rt_triangle* tris;     int nt;
    const rt_meshinfo* mi;       int nm;
    int mode;
    const struct oaccel* accel;     /* non-NULL: triangles are found through the oracle's own search tree (below) */
} scene_t;

/* ------------------------------------------------------------------------------------------------ */
/* Optional search tree of the ORACLE (orc_set_accel(1)).  It is not part of the reference and not the
 * product's BVH: a plain binary tree (spatial-median splits, leaves <= 4 references) over the (chunk,
 * triangle) references in the order the reference's loop visits them, used only so that the checker can
 * finish full-size images.  It must return exactly what the linear loop of calculate_ray_collision
 * returns: the minimum RayTriangle dst, ties to the reference visited first, FLAT_CHUNKS candidates only
 * from chunks whose literal RayBoundingBox passes; tests/test_oracle_cpu.py checks tree == loop bit for
 * bit.  Box tests are done in double on boxes padded 10x wider than the product pads its own, and the
 * distance prune keeps a 1 % + 1e-2 slack, so the tree is more conservative than the thing it checks. */
typedef struct { uint32_t m, tri, seq; } oref;
typedef struct { float lo[3], hi[3]; int32_t left, right; /* right < 0: leaf, refs [left, left + ~right) */ } onode;
typedef struct oaccel { onode* nodes; int nnodes; oref* refs; uint32_t nrefs; } oaccel;
static int g_accel = 0;
void orc_set_accel(int on) { g_accel = on; }

static void ref_box(const scene_t* sc, const oref* r, float pad_g, float lo[3], float hi[3])
{
    const rt_triangle* t = &sc->tris[r->tri];
    for (int a = 0; a < 3; a++) {
        float p0 = t->posA[a], p1 = t->posB[a], p2 = t->posC[a];
        float mn = fminf(p0, fminf(p1, p2)), mx = fmaxf(p0, fmaxf(p1, p2));
        float pad = 3e-4f * fmaxf(fabsf(mn), fabsf(mx)) + pad_g;
        lo[a] = mn - pad; hi[a] = mx + pad;
    }
}

static oaccel* accel_build(const scene_t* sc)
{
    uint64_t n = 0;
    for (int m = 0; m < sc->nm; m++) n += sc->mi[m].numTriangles;
    oaccel* A = (oaccel*)calloc(1, sizeof *A);
    A->nrefs = (uint32_t)n;
    A->refs = (oref*)malloc((n ? n : 1) * sizeof(oref));
    A->nodes = (onode*)malloc((2 * n + 1) * sizeof(onode));
    float G = fmaxf(fabsf(sc->p->camLocalToWorld[3]), fmaxf(fabsf(sc->p->camLocalToWorld[7]), fabsf(sc->p->camLocalToWorld[11])));
    uint32_t k = 0;
    for (int m = 0; m < sc->nm; m++)
        for (uint32_t i = 0; i < sc->mi[m].numTriangles; i++, k++) {
            oref r = { (uint32_t)m, sc->mi[m].firstTriangleIndex + i, k };
            A->refs[k] = r;
            const rt_triangle* t = &sc->tris[r.tri];
            for (int a = 0; a < 3; a++) {
                float v = fmaxf(fabsf(t->posA[a]), fmaxf(fabsf(t->posB[a]), fabsf(t->posC[a])));
                if (v > G && v < INFINITY) G = v;
            }
        }
    const float pad_g = 2e-5f * G;
    if (n == 0) { A->nnodes = 0; return A; }
    float* cen = (float*)malloc(n * 3 * sizeof(float));      /* centroid per seq */
    for (uint32_t i = 0; i < n; i++) {
        const rt_triangle* t = &sc->tris[A->refs[i].tri];
        for (int a = 0; a < 3; a++) cen[3 * i + a] = (t->posA[a] + t->posB[a] + t->posC[a]) * (1.0f / 3.0f);
    }
    typedef struct { int node; uint32_t first, count; } job;
    job* stack = (job*)malloc(128 * sizeof(job)); int sp = 0, cap = 128;
    A->nnodes = 1;
    stack[sp++] = (job){ 0, 0, (uint32_t)n };
    while (sp) {
        job j = stack[--sp];
        onode* nd = &A->nodes[j.node];
        float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int a = 0; a < 3; a++) { nd->lo[a] = INFINITY; nd->hi[a] = -INFINITY; }
        for (uint32_t i = j.first; i < j.first + j.count; i++) {
            float lo[3], hi[3]; ref_box(sc, &A->refs[i], pad_g, lo, hi);
            const float* c = &cen[3 * A->refs[i].seq];
            for (int a = 0; a < 3; a++) {
                if (lo[a] < nd->lo[a]) nd->lo[a] = lo[a];
                if (hi[a] > nd->hi[a]) nd->hi[a] = hi[a];
                if (c[a] < clo[a]) clo[a] = c[a];
                if (c[a] > chi[a]) chi[a] = c[a];
            }
        }
        if (j.count <= 4) { nd->left = (int32_t)j.first; nd->right = ~(int32_t)j.count; continue; }
        int ax = 0;
        if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
        if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        float mid = 0.5f * (clo[ax] + chi[ax]);
        uint32_t lo_i = j.first, hi_i = j.first + j.count;
        while (lo_i < hi_i) {
            if (cen[3 * A->refs[lo_i].seq + ax] < mid) lo_i++;
            else { oref t = A->refs[lo_i]; A->refs[lo_i] = A->refs[--hi_i]; A->refs[hi_i] = t; }
        }
        uint32_t nl = lo_i - j.first;
        if (nl == 0 || nl == j.count) nl = j.count / 2;       /* coincident centroids (or NaN): split by count */
        int l = A->nnodes, r = A->nnodes + 1; A->nnodes += 2;
        nd->left = l; nd->right = r;
        if (sp + 2 > cap) { cap *= 2; stack = (job*)realloc(stack, cap * sizeof(job)); }
        stack[sp++] = (job){ l, j.first, nl };
        stack[sp++] = (job){ r, j.first + nl, j.count - nl };
    }
    free(stack); free(cen);
    return A;
}
static void accel_free(oaccel* A) { if (A) { free(A->nodes); free(A->refs); free(A); } }

/* conservative slab test in double; a zero direction component constrains only by containment */
static inline int accel_box(const onode* nd, const double o[3], const double d[3], const double inv[3], double limit)
{
    double tn = -INFINITY, tf = INFINITY;
    for (int a = 0; a < 3; a++) {
        if (d[a] == 0.0) { if (o[a] < nd->lo[a] || o[a] > nd->hi[a]) return 0; continue; }
        double t0 = (nd->lo[a] - o[a]) * inv[a], t1 = (nd->hi[a] - o[a]) * inv[a];
        if (t0 > t1) { double t = t0; t0 = t1; t1 = t; }
        if (t0 > tn) tn = t0;
        if (t1 < tf) tf = t1;
    }
    return tn <= tf && tf >= 0.0 && tn <= limit;
}

/* CalculateRayCollision :256-297 */
static hit_t calculate_ray_collision(const scene_t* sc, v3 o, v3 d, orc_counts* cnt)
{
    hit_t closest; memset(&closest, 0, sizeof closest);
    closest.dst = INFINITY;
    cnt->rays++;

    for (int i = 0; i < sc->ns; i++) {
        const rt_sphere* s = &sc->spheres[i];
        float dst;
        cnt->sphereTests++;
        if (ray_sphere(o, d, v_load(s->position), s->radius, &dst) && dst < closest.dst) {
            closest.didHit = 1;
            closest.dst = dst;
            closest.hitPoint = v_add(o, v_scale(d, dst));
            closest.normal = v_normalize(v_sub(closest.hitPoint, v_load(s->position)));
            closest.material = &s->material;
        }
    }
    if (sc->accel) {
        const oaccel* A = sc->accel;
        const double od[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
        const double inv[3] = { 1.0 / dd[0], 1.0 / dd[1], 1.0 / dd[2] };
        int nan_ray = !(o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z);
        uint32_t best_seq = 0; int best_is_tri = 0;
        int32_t stack[256]; int sp = 0;
        if (A->nnodes && !nan_ray) stack[sp++] = 0;     /* a NaN ray fails every RayTriangle comparison */
        while (sp) {
            const onode* nd = &A->nodes[stack[--sp]];
            double limit = (closest.dst == INFINITY) ? INFINITY : (double)closest.dst * 1.01 + 1e-2;
            cnt->boxTests++;
            if (!accel_box(nd, od, dd, inv, limit)) continue;
            if (nd->right >= 0) {
                if (sp + 2 > 256) { fprintf(stderr, "oracle: search-tree stack overflow\n"); abort(); }
                stack[sp++] = nd->left; stack[sp++] = nd->right; continue;
            }
            for (int32_t i = nd->left; i < nd->left + ~nd->right; i++) {
                const oref* r = &A->refs[i];
                const rt_triangle* t = &sc->tris[r->tri];
                float dst, u, v, w;
                cnt->triTests++;
                if (!ray_triangle(o, d, t, &dst, &u, &v, &w)) continue;
                /* the loop's `dst < closest.dst` with first-visited-wins on equal dst (spheres are visited before) */
                if (!(dst < closest.dst || (dst == closest.dst && best_is_tri && r->seq < best_seq))) continue;
                const rt_meshinfo* mi = &sc->mi[r->m];
                if (sc->mode == RT_INTERSECT_FLAT_CHUNKS && !ray_bounding_box(o, d, mi->boundsMin, mi->boundsMax)) continue;
                closest.didHit = 1;
                closest.dst = dst;
                closest.hitPoint = v_add(o, v_scale(d, dst));
                v3 nn = v_add(v_add(v_scale(v_load(t->normalA), w), v_scale(v_load(t->normalB), u)),
                              v_scale(v_load(t->normalC), v));
                closest.normal = v_normalize(nn);
                closest.material = &mi->material;
                best_seq = r->seq; best_is_tri = 1;
            }
        }
        if (closest.didHit) cnt->hits++;
        return closest;
    }
    for (int m = 0; m < sc->nm; m++) {
        const rt_meshinfo* mi = &sc->mi[m];
        if (sc->mode == RT_INTERSECT_FLAT_CHUNKS) {
            cnt->boxTests++;
            if (!ray_bounding_box(o, d, mi->boundsMin, mi->boundsMax)) continue;
        }
        for (uint32_t i = 0; i < mi->numTriangles; i++) {
            const rt_triangle* t = &sc->tris[mi->firstTriangleIndex + i];
            float dst, u, v, w;
            cnt->triTests++;
            if (ray_triangle(o, d, t, &dst, &u, &v, &w) && dst < closest.dst) {
                closest.didHit = 1;
                closest.dst = dst;
                closest.hitPoint = v_add(o, v_scale(d, dst));
                v3 nn = v_add(v_add(v_scale(v_load(t->normalA), w), v_scale(v_load(t->normalB), u)),
                              v_scale(v_load(t->normalC), v));
                closest.normal = v_normalize(nn);
                closest.material = &mi->material;
            }
        }
    }
    if (closest.didHit) cnt->hits++;
    return closest;
}

/* GetEnvironmentLight :238-251 */
static v3 environment_light(const rt_params* p, v3 d)
{
    if (!p->environmentEnabled) return V(0, 0, 0);
    float skyGradientT = om_pow(om_smoothstep(0.0f, 0.4f, d.y), 0.35f);
    float groundToSkyT = om_smoothstep(-0.01f, 0.0f, d.y);
    v3 skyGradient = v_lerp(v_load(p->skyColourHorizon), v_load(p->skyColourZenith), skyGradientT);
    float sun = om_pow(om_max(0.0f, v_dot(d, v_load(p->worldSpaceLightPos0))), p->sunFocus) * p->sunIntensity;
    v3 composite = v_lerp(v_load(p->groundColour), skyGradient, groundToSkyT);
    float sunTerm = sun * ((groundToSkyT >= 1.0f) ? 1.0f : 0.0f);
    return V(composite.x + sunTerm, composite.y + sunTerm, composite.z + sunTerm);
}

/* mod2 :232-235 for y = 2 */
static inline float mod2(float x) { return x - 2.0f * floorf(x / 2.0f); }

/* Trace :300-352 */
static v3 trace(const scene_t* sc, v3 o, v3 d, orng* rng, orc_counts* cnt)
{
    const rt_params* p = sc->p;
    v3 incomingLight = V(0, 0, 0);
    v3 rayColour = V(1, 1, 1);

    for (int bounce = 0; bounce <= p->maxBounceCount; bounce++) {
        hit_t h = calculate_ray_collision(sc, o, d, cnt);
        if (h.didHit) {
            const rt_material* m = h.material;
            v3 colour = v_load(m->colour);
            if (m->flag == 1) {                                                /* CheckerPattern :313-317 */
                float cx = mod2(floorf(h.hitPoint.x)), cz = mod2(floorf(h.hitPoint.z));
                if (!(cx == cz)) colour = v_load(m->emissionColour);
            } else if (m->flag == 2 && bounce == 0) {                          /* InvisibleLightSource :318-322 */
                o = v_add(h.hitPoint, v_scale(d, 0.001f));
                continue;
            }
            rng_scope(rng, 1u + 2u * (uint32_t)bounce);
            int isSpecular = m->specularProbability >= rnd(rng);  /* :325 */
            float specF = isSpecular ? 1.0f : 0.0f;
            o = h.hitPoint;                                                    /* :327 */
            v3 diffuseDir = v_normalize(v_add(h.normal, random_direction(rng)));
            v3 specularDir = v_reflect(d, h.normal);
            d = v_normalize(v_lerp(diffuseDir, specularDir, m->smoothness * specF));

            v3 emitted = v_scale(v_load(m->emissionColour), m->emissionStrength);  /* :333 */
            incomingLight = v_add(incomingLight, v_mul(emitted, rayColour));
            rayColour = v_mul(rayColour, v_lerp(colour, v_load(m->specularColour), specF));

            float pr = om_max(rayColour.x, om_max(rayColour.y, rayColour.z));  /* :338-342 */
            if (rnd(rng) >= pr) break;
            float inv = 1.0f / pr;
            rayColour = v_scale(rayColour, inv);
        } else {
            incomingLight = v_add(incomingLight, v_mul(environment_light(p, d), rayColour));
            break;
        }
    }
    return incomingLight;
}

/* frag :356-389 for pixel (x, y) of the full W x H image */
static void frag(const scene_t* sc, int x, int y, int frame, float* out, orc_counts* cnt)
{
    const rt_params* p = sc->p;
    const float* M = p->camLocalToWorld;
    uint32_t W = (uint32_t)p->width, H = (uint32_t)p->height;
    float Wf = (float)W, Hf = (float)H;
    float uvx = ((float)x + 0.5f) / Wf, uvy = ((float)y + 0.5f) / Hf;
    uint32_t pixelIndex = (uint32_t)y * W + (uint32_t)x;
    orng rng; memset(&rng, 0, sizeof rng);
    rng.mode = p->rngMode;
    rng.state = (p->rngMode == RT_RNG_PCG) ? pixelIndex + (uint32_t)frame * 719393u : 0u;      /* :362 */
    rng.key[0] = pixelIndex; rng.key[1] = (uint32_t)frame;

    float lx = (uvx - 0.5f) * p->viewParams[0];
    float ly = (uvy - 0.5f) * p->viewParams[1];
    float lz = 1.0f * p->viewParams[2];
    v3 focusPoint = V(((M[0] * lx + M[1] * ly) + M[2]  * lz) + M[3]  * 1.0f,
                      ((M[4] * lx + M[5] * ly) + M[6]  * lz) + M[7]  * 1.0f,
                      ((M[8] * lx + M[9] * ly) + M[10] * lz) + M[11] * 1.0f);
    v3 camRight = V(M[0], M[4], M[8]);
    v3 camUp    = V(M[1], M[5], M[9]);
    v3 camPos   = v_load(p->worldSpaceCameraPos);

    v3 total = V(0, 0, 0);
    v3 part[16];                                                   /* Philox mode: the sub-streams' sums */
    const int S = orc_philox_substreams(p->numRaysPerPixel);
    for (int k = 0; k < 16; k++) part[k] = V(0, 0, 0);
    for (int rayIndex = 0; rayIndex < p->numRaysPerPixel; rayIndex++) {
        float jx, jy;
        rng.sample = (uint32_t)rayIndex; rng_scope(&rng, 0u);
        random_point_in_circle(&rng, &jx, &jy);
        jx = jx * p->defocusStrength / Wf;  jy = jy * p->defocusStrength / Wf;
        v3 origin = v_add(v_add(camPos, v_scale(camRight, jx)), v_scale(camUp, jy));

        random_point_in_circle(&rng, &jx, &jy);
        jx = jx * p->divergeStrength / Wf;  jy = jy * p->divergeStrength / Wf;
        v3 jfp = v_add(v_add(focusPoint, v_scale(camRight, jx)), v_scale(camUp, jy));
        v3 dir = v_normalize(v_sub(jfp, origin));
        v3 light = trace(sc, origin, dir, &rng, cnt);
        if (p->rngMode == RT_RNG_PCG) total = v_add(total, light);
        else part[rayIndex % S] = v_add(part[rayIndex % S], light);
    }
    if (p->rngMode != RT_RNG_PCG) {
        for (int step = 1; step < S; step <<= 1)
            for (int k = 0; k < S; k += 2 * step) part[k] = v_add(part[k], part[k + step]);
        total = part[0];
    }
    float n = (float)p->numRaysPerPixel;
    out[0] = total.x / n; out[1] = total.y / n; out[2] = total.z / n; out[3] = 1.0f;
}

/* ------------------------------------------------------------------------------------------------ */

int orc_render_frame(const rt_params* params,
                     const rt_sphere* spheres, int ns,
                     const rt_triangle* tris, int nt,
                     const rt_meshinfo* meshinfo, int nm,
                     int frame, int x0, int y0, int x1, int y1,
                     float* out_rgba, int nthreads, orc_counts* counts)
{
    if (!params || !out_rgba) return -1;
    if (x0 < 0 || y0 < 0 || x1 > params->width || y1 > params->height || x0 > x1 || y0 > y1) return -2;
    for (int m = 0; m < nm; m++)
        if ((uint64_t)meshinfo[m].firstTriangleIndex + meshinfo[m].numTriangles > (uint64_t)nt) return -3;
    scene_t sc = { params, spheres, ns, tris, nt, meshinfo, nm, params->intersectMode, NULL };
    oaccel* accel = g_accel ? accel_build(&sc) : NULL;
    sc.accel = accel;
    orc_counts tot; memset(&tot, 0, sizeof tot);
    int cw = x1 - x0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    tot.threads = nthreads;
#pragma omp parallel num_threads(nthreads)
    {
        orc_counts loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 1)
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++)
                frag(&sc, x, y, frame, out_rgba + ((size_t)(y - y0) * cw + (x - x0)) * 4, &loc);
#pragma omp critical
        {
            tot.rays += loc.rays; tot.sphereTests += loc.sphereTests; tot.boxTests += loc.boxTests;
            tot.triTests += loc.triTests; tot.hits += loc.hits;
        }
    }
    accel_free(accel);
    if (counts) *counts = tot;
    return 0;
}

/* Accumulate.shader:43-54 — weight = 1/(_Frame+1); saturate(prev*(1-weight) + cur*weight), all 4 channels */
void orc_accumulate(float* accum, const float* cur, size_t n_floats, int frame)
{
    float weight = 1.0f / (float)(frame + 1);
    float omw = 1.0f - weight;
    for (size_t i = 0; i < n_floats; i++)
        accum[i] = om_saturate(accum[i] * omw + cur[i] * weight);
}

/* Display step after the path: the reference's final Blit(resultTexture, target) (RayTracingManager.cs:84) writes the
 * linear RGBA32F result into an sRGB 8-bit back buffer (Linear colour space project, ProjectSettings.asset:50).  Frozen:
 * c' = c <= 0.0031308 ? 12.92*c : 1.055*pow(c, 1/2.4) - 0.055 with the frozen pow above; v = (uint8)floor(sat(c')*255 + 0.5);
 * alpha is stored linearly.  Output pixel = R | G << 8 | B << 16 | A << 24. */
static inline uint32_t to_unorm8(float v) { return (uint32_t)floorf(om_saturate(v) * 255.0f + 0.5f); }
static inline float linear_to_srgb(float c)
{
    c = om_saturate(c);
    return (c <= 0.0031308f) ? 12.92f * c : 1.055f * om_pow(c, 0.41666666f) - 0.055f;
}
void orc_display_srgb8(const float* rgba, uint32_t* out, size_t n_pixels)
{
    for (size_t i = 0; i < n_pixels; i++) {
        const float* p = rgba + 4 * i;
        out[i] = to_unorm8(linear_to_srgb(p[0])) | (to_unorm8(linear_to_srgb(p[1])) << 8)
               | (to_unorm8(linear_to_srgb(p[2])) << 16) | (to_unorm8(p[3]) << 24);
    }
}

int orc_hw_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
