// rt_pool.hpp — the wave-pool megakernel: ballot / prefix-sum ray compaction inside the wave.
//
// Same per-pixel arithmetic as k_trace / k_stream (bit-identical images); what changes is which lane executes it.
// k_trace binds a pixel to a lane, so every phase of the state machine runs at the fraction of lanes that happen
// to need it (measured on the 100k-triangle workload: 37 % in BVH node steps, 29 % in triangle tests, 41-63 % in
// shading).  Here a wave owns a pool of P = 128 pixel slots whose path state lives in LDS (19 dwords per slot, SoA),
// and each phase gathers the slots that need it into dense lane sets with a ballot + prefix-sum compaction:
//
//   TRAVERSE  lanes hold one in-flight closest-hit query each (ray, best hit, slab constants in registers, stack in
//             LDS); every step the wave runs a BVH4 node step or a triangle test, whichever more lanes wait for; a
//             lane whose query completes writes the hit to its slot and immediately takes the next pending ray, so
//             the lanes stay busy until the pending list runs dry.  Unfinished queries are suspended in place
//             (cur / sp stay in registers, the rest is re-read from the slot) when the wave leaves the phase.
//   SHADE     up to 64 slots whose query hit something: Trace :309-343 (material, scatter, Russian roulette).
//   FINISH    up to 64 slots whose path ended or that are empty: environment light for misses (:346), sample
//             bookkeeping (frag :384), pixel completion (frag :387-388 + Accumulate.shader:45-50), pixel refill
//             from the global tile-major counter, next camera ray (frag :377-382).
//
// The PCG state is per pixel and only ever advanced by that pixel's own phases in order, so the RNG chain
// (RayTracing.shader:362,374-385) is untouched.  The running sum of a pixel's samples lives in its out_frame texel
// (read-modify-write by the one wave that owns the pixel), which keeps the LDS footprint at 12.8 KB per wave:
// three 256-thread workgroups per CU.
#pragma once
#include "rt_kernels.hpp"

namespace rtk {

struct PoolArgs {
    unsigned int total_pixels;      // tiles_x * tiles_y * 64 (tile-major enumeration, padded)
    int trav_min_lanes;             // leave TRAVERSE when fewer lanes than this are still in flight (and no ray is pending)
    int lds_stack_cap;              // stack entries per lane kept in LDS; deeper entries spill to gstack
    uint32_t* gstack;               // [spill entries][total lanes] overflow stack (may be null when nothing can spill)
    unsigned int gstack_stride;     // total lanes of the launch
};

namespace pool {
constexpr int P = 128;              // slots per wave
enum Field { OX, OY, OZ, DX, DY, DZ, HT, HID, HU, HV, RNG, PIX, SB, LR, LG, LB, CR, CG, CB, NF };
enum State : uint32_t { EMPTY = 0, PATHEND = 1, MISS = 2, HIT = 3, PEND = 4, FLY = 5, DEAD = 6 };
// SB = state | sample << 4 | bounce << 16
__device__ __forceinline__ uint32_t pack_sb(uint32_t st, int sample, int bounce) { return st | ((uint32_t)sample << 4) | ((uint32_t)bounce << 16); }
constexpr int kMaxSamples = 4095;
__host__ __device__ constexpr int wave_dwords(int stack_cap) { return NF * P + stack_cap * 64 + P; }
} // namespace pool

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_pool(DeviceScene S, FrameArgs F, PoolArgs A)
{
    using namespace pool;
    extern __shared__ uint32_t lds_pool[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* const slot = lds_pool + (size_t)wave * wave_dwords(A.lds_stack_cap);
    uint32_t* const stk = slot + NF * P + lane;
    uint32_t* const scratch = slot + NF * P + A.lds_stack_cap * 64;
    float* const slotf = reinterpret_cast<float*>(slot);
    const unsigned int gid = (blockIdx.x * kBlock + threadIdx.x);
    uint32_t* const gstk = A.gstack ? A.gstack + gid : nullptr;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    Counters cnt = {};
    const rt_params& p = F.p;
    const float* M = p.camLocalToWorld;
    const uint32_t W = (uint32_t)p.width;
    const float weight = 1.0f / (float)(F.frame + 1);                                  // Accumulate.shader:48
    const float omw = 1.0f - weight;
    const float INF = __builtin_inff();
    const int cap = A.lds_stack_cap;

    Camera cam;
    cam.W = (float)W;
    cam.right = rtm::mk(M[0], M[4], M[8]);
    cam.up    = rtm::mk(M[1], M[5], M[9]);
    cam.pos   = ld3(p.worldSpaceCameraPos);
    cam.focusPoint = rtm::mk(0.f, 0.f, 0.f);

#define SLOT(f, s) slot[(f) * P + (s)]
#define SLOTF(f, s) slotf[(f) * P + (s)]
    SLOT(SB, lane) = EMPTY; SLOT(SB, lane + 64) = EMPTY;
    SLOT(PIX, lane) = 0; SLOT(PIX, lane + 64) = 0;
    __builtin_amdgcn_wave_barrier();

    // ---- traversal registers that survive across phases (a suspended query stays bound to its lane)
    bool fly = false;
    uint32_t myslot = 0, cur = kNone;
    int sp = 0;
    bool pixels_left = true;            // wave-uniform: the global pixel queue is not exhausted yet

    // stack with LDS head and global spill
    auto push = [&](uint32_t v) {
        if (sp < cap) stk[sp * 64] = v; else gstk[(size_t)(sp - cap) * A.gstack_stride] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        return sp < cap ? stk[sp * 64] : gstk[(size_t)(sp - cap) * A.gstack_stride];
    };

    // Every loop is bounded so that a scheduling bug can only produce a wrong image, never a wave that does not drain.
    for (unsigned int guard = 0; guard < (1u << 28); ++guard) {
        // ================================ census ================================
        __builtin_amdgcn_wave_barrier();
        const uint32_t st0 = SLOT(SB, lane) & 15u, st1 = SLOT(SB, lane + 64) & 15u;
        const unsigned long long hit0 = ballot_(st0 == HIT), hit1 = ballot_(st1 == HIT);
        const unsigned long long pen0 = ballot_(st0 == PEND), pen1 = ballot_(st1 == PEND);
        const bool f0 = st0 == PATHEND || st0 == MISS || (st0 == EMPTY && pixels_left);
        const bool f1 = st1 == PATHEND || st1 == MISS || (st1 == EMPTY && pixels_left);
        const unsigned long long fin0 = ballot_(f0), fin1 = ballot_(f1);
        const int nHit = __popcll(hit0) + __popcll(hit1), nPend = __popcll(pen0) + __popcll(pen1);
        const int nFin = __popcll(fin0) + __popcll(fin1), nFly = __popcll(ballot_(fly));
        if (nHit + nPend + nFin + nFly == 0) break;

        int phase;                      // 0 TRAVERSE, 1 SHADE, 2 FINISH
        if (nHit >= 64) phase = 1;
        else if (nFin >= 64) phase = 2;
        else if (nPend + nFly > 0 && (nPend + nFly >= A.trav_min_lanes || nHit + nFin == 0)) phase = 0;
        else if (nHit >= nFin) phase = 1;
        else phase = 2;

        if (phase == 0) {
            // ================================ TRAVERSE ================================
            // pending slot ids -> scratch[0 .. nPend)
            if ((pen0 >> lane) & 1ull) scratch[__popcll(pen0 & lt_mask)] = (uint32_t)lane;
            if ((pen1 >> lane) & 1ull) scratch[__popcll(pen0) + __popcll(pen1 & lt_mask)] = (uint32_t)lane + 64u;
            __builtin_amdgcn_wave_barrier();
            int pendNext = 0;
            v3 o = rtm::mk(0.f, 0.f, 0.f), d = rtm::mk(1.f, 1.f, 1.f);
            Hit best; best.t = INF; best.id = kNone; best.u = 0.f; best.v = 0.f;
            if (fly) {      // resume a suspended query
                o = rtm::mk(SLOTF(OX, myslot), SLOTF(OY, myslot), SLOTF(OZ, myslot));
                d = rtm::mk(SLOTF(DX, myslot), SLOTF(DY, myslot), SLOTF(DZ, myslot));
                best.t = SLOTF(HT, myslot); best.id = SLOT(HID, myslot); best.u = SLOTF(HU, myslot); best.v = SLOTF(HV, myslot);
            }
            RaySlab slab = make_slab(o, d);
            bool refill = true;
            for (unsigned int tguard = 0; tguard < (1u << 24); ++tguard) {
                if (refill) {
                    // ---- idle lanes take pending rays
                    const unsigned long long idle = ballot_(!fly);
                    const int avail = nPend - pendNext;
                    if (avail > 0 && idle != 0) {
                        const int rank = __popcll(idle & lt_mask);
                        if (!fly && rank < avail) {
                            myslot = scratch[pendNext + rank];
                            o = rtm::mk(SLOTF(OX, myslot), SLOTF(OY, myslot), SLOTF(OZ, myslot));
                            d = rtm::mk(SLOTF(DX, myslot), SLOTF(DY, myslot), SLOTF(DZ, myslot));
                            // ---- new closest-hit query: CalculateRayCollision :256-273 (spheres in buffer order)
                            cnt.rays++;
                            best.t = INF; best.id = kNone; best.u = 0.f; best.v = 0.f;
                            const float a = rtm::dot(d, d);
                            const SphereA sa = sphere_a(a);
                            for (int i = 0; i < S.ns; ++i) {
                                const float4 s = S.sph_geom[i];
                                float dst;
                                if (COUNT) cnt.sph++;
                                if (ray_sphere(o, d, sa, rtm::mk(s.x, s.y, s.z), s.w, dst) && dst < best.t) { best.t = dst; best.id = (uint32_t)i; }
                            }
                            if (S.nn > 0 && ray_traceable(o, d, a)) {
                                slab = make_slab(o, d);
                                cur = 0; sp = 0; fly = true;
                                SLOT(SB, myslot) = (SLOT(SB, myslot) & ~15u) | FLY;
                            } else {
                                SLOTF(HT, myslot) = best.t; SLOT(HID, myslot) = best.id;
                                SLOT(SB, myslot) = (SLOT(SB, myslot) & ~15u) | (best.id == kNone ? MISS : HIT);
                            }
                        }
                        pendNext += min(avail, __popcll(idle));
                    }
                    refill = false;
                }
                const unsigned long long mFly = ballot_(fly);
                const int nF = __popcll(mFly);
                if (nF == 0) { if (pendNext < nPend) { refill = true; continue; } break; }
                if (pendNext >= nPend && nF < A.trav_min_lanes && nF < 64) {
                    // only leave if some slot is waiting for SHADE/FINISH; otherwise finish the stragglers here
                    const uint32_t s0 = SLOT(SB, lane) & 15u, s1 = SLOT(SB, lane + 64) & 15u;
                    const bool other = (s0 == HIT || s0 == MISS || s0 == PATHEND || (s0 == EMPTY && pixels_left))
                                    || (s1 == HIT || s1 == MISS || s1 == PATHEND || (s1 == EMPTY && pixels_left));
                    if (ballot_(other) != 0) break;
                }
                const int nNode = __popcll(ballot_(fly && (int)cur >= 0));
                if (2 * nNode >= nF) {
                    // ---- NODE step
                    if (fly && (int)cur >= 0) {
                        if (COUNT) cnt.nodes++;
                        phase_tick<COUNT>(cnt, 0);
                        float t0, t1, t2, t3;
                        uint32_t c0, c1, c2, c3;
                        node_step<false>(S.nodes, cur, slab, best.t, F.full_sort != 0, t0, t1, t2, t3, c0, c1, c2, c3);
                        if (t3 < INF) push(c3);
                        if (t2 < INF) push(c2);
                        if (t1 < INF) push(c1);
                        if (t0 < INF) cur = c0;
                        else if (sp > 0) cur = pop();
                        else cur = kNone;
                    }
                } else {
                    // ---- TRI step
                    if (fly && (int)cur < 0) {
                        const uint32_t ti = (cur & 0x7FFFFFFFu) >> 2;
                        float4 g0, g1, g2;
                        load_tri(S.tri_geo, ti, g0, g1, g2);
                        float dst, u, v;
                        if (COUNT) cnt.tris++;
                        phase_tick<COUNT>(cnt, 1);
                        const bool hit = ray_triangle(o, d, rtm::mk(g0.x, g0.y, g0.z), rtm::mk(g0.w, g1.x, g1.y),
                                                      rtm::mk(g1.z, g1.w, g2.x), rtm::mk(g2.y, g2.z, g2.w), dst, u, v);
                        if (hit && dst <= best.t) {
                            bool take = dst < best.t;
                            if (!take && (best.id & kTriBit) && best.id != kNone) {
                                // equal dst: the reference keeps the triangle that comes first in the buffer
                                uint32_t oc = __float_as_uint(S.tri_nrm[(size_t)ti * 3 + 1].w);
                                uint32_t ob = __float_as_uint(S.tri_nrm[(size_t)(best.id & ~kTriBit) * 3 + 1].w);
                                take = oc < ob;
                            }
                            if (take && p.intersectMode == RT_INTERSECT_FLAT_CHUNKS) {
                                // the reference only reaches this triangle if its chunk's box test passes (:279)
                                uint32_t chunk = __float_as_uint(S.tri_nrm[(size_t)ti * 3].w);
                                float4 bmn = S.chunk_box[(size_t)chunk * 2], bmx = S.chunk_box[(size_t)chunk * 2 + 1];
                                take = ray_bounding_box(o, slab.inv, rtm::mk(bmn.x, bmn.y, bmn.z), rtm::mk(bmx.x, bmx.y, bmx.z));
                            }
                            if (take) { best.t = dst; best.id = kTriBit | ti; best.u = u; best.v = v; }
                        }
                        if (cur & 3u) cur += 3u;            // next triangle of the leaf (first + 1, count - 1)
                        else if (sp > 0) cur = pop();
                        else cur = kNone;
                    }
                }
                // ---- completed queries leave their lane
                if (fly && cur == kNone) {
                    SLOTF(HT, myslot) = best.t; SLOT(HID, myslot) = best.id; SLOTF(HU, myslot) = best.u; SLOTF(HV, myslot) = best.v;
                    SLOT(SB, myslot) = (SLOT(SB, myslot) & ~15u) | (best.id == kNone ? MISS : HIT);
                    fly = false;
                }
                if (ballot_(!fly) != 0 && pendNext < nPend) refill = true;
            }
            // ---- suspend what is still in flight: the best hit so far goes back to the slot
            if (fly) { SLOTF(HT, myslot) = best.t; SLOT(HID, myslot) = best.id; SLOTF(HU, myslot) = best.u; SLOTF(HV, myslot) = best.v; }
        } else if (phase == 1) {
            // ================================ SHADE ================================
            if ((hit0 >> lane) & 1ull) scratch[__popcll(hit0 & lt_mask)] = (uint32_t)lane;
            if ((hit1 >> lane) & 1ull) scratch[__popcll(hit0) + __popcll(hit1 & lt_mask)] = (uint32_t)lane + 64u;
            __builtin_amdgcn_wave_barrier();
            if (lane < min(nHit, 64)) {
                const uint32_t s = scratch[lane];
                phase_tick<COUNT>(cnt, 2);
                if (COUNT) cnt.hits++;
                v3 o = rtm::mk(SLOTF(OX, s), SLOTF(OY, s), SLOTF(OZ, s));
                v3 d = rtm::mk(SLOTF(DX, s), SLOTF(DY, s), SLOTF(DZ, s));
                const float bt = SLOTF(HT, s), bu = SLOTF(HU, s), bv = SLOTF(HV, s);
                const uint32_t bid = SLOT(HID, s);
                uint32_t rng = SLOT(RNG, s);
                const uint32_t sb = SLOT(SB, s);
                int bounce = (int)(sb >> 16);
                v3 light = rtm::mk(SLOTF(LR, s), SLOTF(LG, s), SLOTF(LB, s));
                v3 rayColour = rtm::mk(SLOTF(CR, s), SLOTF(CG, s), SLOTF(CB, s));
                bool path_done = false;
                // ---- hit: Trace :309-343
                const v3 hitPoint = o + d * bt;
                v3 normal; const float4* mat;
                if (bid & kTriBit) {
                    const uint32_t ti = bid & ~kTriBit;
                    const float4* tn = S.tri_nrm + (size_t)ti * 3;
                    const float4 n0 = tn[0], n1 = tn[1], n2 = tn[2];
                    const float w = 1.0f - bu - bv;
                    normal = rtm::normalize((rtm::mk(n0.x, n0.y, n0.z) * w + rtm::mk(n1.x, n1.y, n1.z) * bu)
                                            + rtm::mk(n2.x, n2.y, n2.z) * bv);
                    mat = S.chunk_mat + (size_t)__float_as_uint(n0.w) * 4;
                } else {
                    const float4 sg = S.sph_geom[bid];
                    normal = rtm::normalize(hitPoint - rtm::mk(sg.x, sg.y, sg.z));
                    mat = S.sph_mat + (size_t)bid * 4;
                }
                const float4 mcol = mat[0], memi = mat[1], mspec = mat[2], mprm = mat[3];
                const int flag = (int)__float_as_uint(mprm.w);
                v3 colour = rtm::mk(mcol.x, mcol.y, mcol.z);
                bool skip = false;
                if (flag == 1) {                                               // CheckerPattern :313-317
                    float cx = mod2(__builtin_floorf(hitPoint.x)), cz = mod2(__builtin_floorf(hitPoint.z));
                    if (!(cx == cz)) colour = rtm::mk(memi.x, memi.y, memi.z);
                } else if (flag == 2 && bounce == 0) {                         // InvisibleLightSource :318-322
                    o = hitPoint + d * 0.001f;
                    skip = true;
                }
                if (!skip) {
                    const bool isSpecular = mprm.z >= rtm::random_value(rng);  // :325
                    const float specF = isSpecular ? 1.0f : 0.0f;
                    o = hitPoint;                                              // :327
                    v3 diffuseDir = rtm::normalize(normal + rtm::random_direction(rng));
                    v3 specularDir = rtm::reflect(d, normal);
                    d = rtm::normalize(rtm::lerp(diffuseDir, specularDir, mprm.y * specF));
                    v3 emitted = rtm::mk(memi.x, memi.y, memi.z) * mprm.x;     // :333-335
                    light = light + emitted * rayColour;
                    rayColour = rayColour * rtm::lerp(colour, rtm::mk(mspec.x, mspec.y, mspec.z), specF);
                    float pr = rtm::fmax_(rayColour.x, rtm::fmax_(rayColour.y, rayColour.z));   // :338-342
                    if (rtm::random_value(rng) >= pr) path_done = true;
                    else { float ip = rtm::rcp_(pr); rayColour = rayColour * ip; }
                }
                ++bounce;
                if (bounce > p.maxBounceCount) path_done = true;               // loop bound :305
                SLOTF(OX, s) = o.x; SLOTF(OY, s) = o.y; SLOTF(OZ, s) = o.z;
                SLOTF(DX, s) = d.x; SLOTF(DY, s) = d.y; SLOTF(DZ, s) = d.z;
                SLOT(RNG, s) = rng;
                SLOTF(LR, s) = light.x; SLOTF(LG, s) = light.y; SLOTF(LB, s) = light.z;
                SLOTF(CR, s) = rayColour.x; SLOTF(CG, s) = rayColour.y; SLOTF(CB, s) = rayColour.z;
                SLOT(SB, s) = (sb & 0xFFF0u) | ((uint32_t)bounce << 16) | (path_done ? PATHEND : PEND);
            }
        } else {
            // ================================ FINISH ================================
            if ((fin0 >> lane) & 1ull) scratch[__popcll(fin0 & lt_mask)] = (uint32_t)lane;
            if ((fin1 >> lane) & 1ull) scratch[__popcll(fin0) + __popcll(fin1 & lt_mask)] = (uint32_t)lane + 64u;
            __builtin_amdgcn_wave_barrier();
            const bool mine = lane < min(nFin, 64);
            uint32_t s = 0, st = DEAD, rng = 0, pix = 0;
            int sample = 0;
            v3 light = rtm::mk(0.f, 0.f, 0.f);
            if (mine) {
                s = scratch[lane];
                const uint32_t sb = SLOT(SB, s);
                st = sb & 15u; sample = (int)((sb >> 4) & 0xFFFu);
                rng = SLOT(RNG, s); pix = SLOT(PIX, s);
                if (st != EMPTY) {
                    light = rtm::mk(SLOTF(LR, s), SLOTF(LG, s), SLOTF(LB, s));
                    if (st == MISS) {
                        phase_tick<COUNT>(cnt, 3);
                        const v3 d = rtm::mk(SLOTF(DX, s), SLOTF(DY, s), SLOTF(DZ, s));
                        const v3 rayColour = rtm::mk(SLOTF(CR, s), SLOTF(CG, s), SLOTF(CB, s));
                        light = light + environment_light(p, d) * rayColour;       // :346-347
                    }
                    // ---- path end: frag :384
                    float4 tot = F.out_frame[pix];
                    tot.x = tot.x + light.x; tot.y = tot.y + light.y; tot.z = tot.z + light.z;
                    ++sample;
                    if (sample >= p.numRaysPerPixel) {
                        // ---- pixel complete: frag :387-388 + Accumulate.shader:45-50
                        const float n = (float)p.numRaysPerPixel;
                        const float cx = tot.x / n, cy = tot.y / n, cz = tot.z / n;
                        F.out_frame[pix] = make_float4(cx, cy, cz, 1.0f);
                        const float4 prev = F.accum[pix];
                        float4 acc;
                        acc.x = rtm::saturate(prev.x * omw + cx * weight);
                        acc.y = rtm::saturate(prev.y * omw + cy * weight);
                        acc.z = rtm::saturate(prev.z * omw + cz * weight);
                        acc.w = rtm::saturate(prev.w * omw + 1.0f * weight);
                        F.accum[pix] = acc;
                        st = EMPTY;
                    } else {
                        F.out_frame[pix] = tot;
                    }
                }
            }
            // ---- pixel refill: tile-major global order; indices outside the strip are skipped
            int px = 0, ly = 0;
            bool have_pixel = mine && st != EMPTY;
            if (have_pixel) { ly = (int)(pix / W); px = (int)(pix - (uint32_t)ly * W); }
            for (unsigned int rguard = 0; rguard < (1u << 24); ++rguard) {
                const unsigned long long need = ballot_(mine && st == EMPTY);
                if (need == 0) break;
                if (!pixels_left) { if (mine && st == EMPTY) st = DEAD; break; }
                unsigned int base = 0;
                const int first = __builtin_ctzll(need);
                if (lane == first) base = atomicAdd(F.tile_counter, (unsigned int)__popcll(need));
                base = __shfl(base, first, 64);
                if (mine && st == EMPTY) {
                    const unsigned int idx = base + (unsigned int)__popcll(need & lt_mask);
                    if (idx < A.total_pixels) {
                        const unsigned int tile = idx >> 6, within = idx & 63u;
                        const int x = (int)(tile % (unsigned)F.tiles_x) * 8 + (int)(within & 7u);
                        const int yy = (int)(tile / (unsigned)F.tiles_x) * 8 + (int)(within >> 3);
                        if (x < p.width && yy < F.nrows) {
                            px = x; ly = yy; pix = (uint32_t)yy * W + (uint32_t)x;
                            rng = ((uint32_t)(F.row0 + (yy >> 3) * F.row_stride + (yy & 7)) * W + (uint32_t)x) + (uint32_t)F.frame * 719393u;   // :361-362
                            sample = 0;
                            F.out_frame[pix] = make_float4(0.f, 0.f, 0.f, 0.f);       // running sum of the pixel's samples
                            st = PEND;      // "has a pixel" (the real state is written below)
                            have_pixel = true;
                        }
                    }
                }
                if (base + (unsigned int)__popcll(need) >= A.total_pixels) pixels_left = false;
            }
            if (mine) {
                if (have_pixel) {
                    // ---- next camera ray: frag :364-382
                    phase_tick<COUNT>(cnt, 4);
                    const int y = F.row0 + (ly >> 3) * F.row_stride + (ly & 7);
                    const float uvx = ((float)px + 0.5f) / cam.W, uvy = ((float)y + 0.5f) / (float)(uint32_t)p.height;
                    const float lx = (uvx - 0.5f) * p.viewParams[0], lyv = (uvy - 0.5f) * p.viewParams[1], lz = 1.0f * p.viewParams[2];
                    cam.focusPoint = rtm::mk(((M[0] * lx + M[1] * lyv) + M[2]  * lz) + M[3]  * 1.0f,
                                             ((M[4] * lx + M[5] * lyv) + M[6]  * lz) + M[7]  * 1.0f,
                                             ((M[8] * lx + M[9] * lyv) + M[10] * lz) + M[11] * 1.0f);
                    v3 o, d;
                    camera_ray(p, cam, rng, o, d);
                    SLOTF(OX, s) = o.x; SLOTF(OY, s) = o.y; SLOTF(OZ, s) = o.z;
                    SLOTF(DX, s) = d.x; SLOTF(DY, s) = d.y; SLOTF(DZ, s) = d.z;
                    SLOT(RNG, s) = rng; SLOT(PIX, s) = pix;
                    SLOTF(LR, s) = 0.f; SLOTF(LG, s) = 0.f; SLOTF(LB, s) = 0.f;
                    SLOTF(CR, s) = 1.f; SLOTF(CG, s) = 1.f; SLOTF(CB, s) = 1.f;
                    SLOT(SB, s) = pack_sb(PEND, sample, 0);
                } else {
                    SLOT(SB, s) = (st == DEAD) ? (uint32_t)DEAD : (uint32_t)EMPTY;
                }
            }
        }
    }
#undef SLOT
#undef SLOTF
    {
        unsigned long long v[kNumCounters] = { cnt.rays, cnt.sph, cnt.nodes, cnt.tris, cnt.hits };
        for (int k = 0; k < 5; ++k) { v[5 + k] = cnt.phase_lanes[k]; v[10 + k] = cnt.phase_execs[k]; }
        for (int k = 0; k < (COUNT ? kNumCounters : 1); ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) atomicAdd(&F.counters[k], s);
        }
    }
}

} // namespace rtk
