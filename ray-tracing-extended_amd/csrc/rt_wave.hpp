// rt_wave.hpp — k_wave: the wave-pool megakernel.  Same per-pixel arithmetic as k_trace / k_stream (bit-identical
// images); what changes is which lane executes it and when.
//
// k_stream keeps one pixel per lane: a lane whose closest-hit query is complete idles until `shade_threshold` lanes
// wait with it, and the SHADE pass then runs hit shading, environment light and camera-ray generation one after the
// other at the fraction of lanes that need each (measured on the 100k-triangle workload: node steps 52 %, triangle
// tests 44 %, hit shading 53 %, environment 35 %, camera rays 45 % of the lanes).  Here a wave owns P = 256 pixel
// slots whose path state lives in global memory (22 dwords per slot, SoA per wave, so a phase's gathers stay inside
// 1 KB rows); only a state byte per slot and the phase's work list (slot ids) are in LDS.  Each turn of the outer loop
// takes a census of the slot states and runs one phase on up to 64 slots gathered with ballot + prefix sums:
//
//   TRAVERSE  lanes hold one in-flight closest-hit query each (ray, slab constants, best hit in registers, stack in
//             LDS).  The loop is k_stream's while-while burst (capped node loop, whole leaves); a lane whose query
//             completes writes the hit to its slot and, as soon as `refill_min` lanes are idle, the idle lanes take
//             the next pending rays together (ray load, sphere loop, slab constants: once per `refill_min` rays, not
//             per ray).  When the pending list is dry and fewer than `trav_min_lanes` queries are in flight, the wave
//             leaves the phase; the stragglers stay suspended on their lanes (cur / sp in registers, stack in LDS,
//             the rest re-read from the slot) and resume beside the next batch.
//   SHADE     64 slots whose query hit something: Trace :309-343 (material, scatter, Russian roulette) -> PEND or PATHEND.
//   FINISH    64 slots whose path ended (MISS: environment light first, :346) or that are empty: sample bookkeeping
//             (frag :384), pixel completion (frag :387-388 + Accumulate.shader:45-50), pixel refill from the launch's
//             work queue ((frame, tile) items in costliest-first order, handed out pixel by pixel), next camera ray
//             (frag :377-382) -> PEND.
//
// The PCG state is per pixel and only ever advanced by that pixel's own phases in order, so the RNG chain
// (RayTracing.shader:362,374-385) is untouched.  All waves of a CU share one vector L1, and a wave's memory
// operations are performed in order, so the slot rows need no more than the workgroup-scope fence at the phase boundary.
#pragma once
#include "rt_kernels.hpp"

namespace rtk {

struct WaveArgs {
    uint32_t* state;            // [waves of the launch][NF][P]
    unsigned int total_pixels;  // work items * 64, work item = (frame, tile)
    int refill_min;             // idle lanes that trigger a refill from the pending list
    int trav_min_lanes;         // leave TRAVERSE below this many in-flight lanes once the pending list is dry
    int node_min;               // as in k_stream: the node loop goes on while at least this many lanes hold an internal node
};

namespace wv {
constexpr int P = 256;              // slots per wave (slot ids fit a byte)
enum Field { OX, OY, OZ, DX, DY, DZ, HT, HID, HU, HV, RNG, PXY, SB, LR, LG, LB, CR, CG, CB, TR, TG, TB, NF };
enum State : uint32_t { EMPTY = 0, PATHEND = 1, MISS = 2, HIT = 3, PEND = 4, FLY = 5, DEAD = 6 };
// SB = sample | bounce << 12 | frame offset << 24
constexpr int kMaxSamples = 4095, kMaxBounce = 4094, kMaxFrames = 255;
__host__ __device__ constexpr size_t wave_lds_bytes(int stack_cap) { return (size_t)stack_cap * 256 + 2 * P; }
__host__ __device__ constexpr size_t wave_state_dwords() { return (size_t)NF * P; }
} // namespace wv

template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wave(DeviceScene S, FrameArgs F, WaveArgs A)
{
    using namespace wv;
    extern __shared__ uint32_t lds_wave[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* const wbase = reinterpret_cast<unsigned char*>(lds_wave) + (size_t)wave * wave_lds_bytes(F.stack_cap);
    uint32_t* const stk = reinterpret_cast<uint32_t*>(wbase) + lane;
    unsigned char* const sst = wbase + (size_t)F.stack_cap * 256;       // state byte per slot
    unsigned char* const list = sst + P;                                // the phase's work list (slot ids)
    uint32_t* const gs = A.state + (size_t)(blockIdx.x * kWavesPerBlock + wave) * wave_state_dwords();
    float* const gsf = reinterpret_cast<float*>(gs);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#define G(f, s)  gs[(f) * P + (s)]
#define GF(f, s) gsf[(f) * P + (s)]

    Counters cnt = {};
    const rt_params& p = F.p;
    const float* M = p.camLocalToWorld;
    const uint32_t W = (uint32_t)p.width;
    const float weight = 1.0f / (float)(F.frame + 1);                                  // Accumulate.shader:48
    const float omw = 1.0f - weight;
    const float INF = __builtin_inff();
    const unsigned int ntiles = (unsigned)(F.tiles_x * F.tiles_y);

    Camera cam;
    cam.W = (float)W;
    cam.right = rtm::mk(M[0], M[4], M[8]);
    cam.up    = rtm::mk(M[1], M[5], M[9]);
    cam.pos   = ld3(p.worldSpaceCameraPos);
    cam.focusPoint = rtm::mk(0.f, 0.f, 0.f);

#pragma unroll
    for (int k = 0; k < P / 64; ++k) sst[lane + 64 * k] = (unsigned char)EMPTY;

    // ---- traversal registers that survive across phases (a suspended query stays bound to its lane)
    bool fly = false;
    uint32_t myslot = 0, cur = kNone;
    int sp = 0;
    bool pixels_left = true;            // wave-uniform: the launch's work queue is not exhausted yet

    // Every loop is bounded so that a scheduling bug can only produce a wrong image, never a wave that does not drain.
    for (unsigned int guard = 0; guard < (1u << 28); ++guard) {
        // ================================ census ================================
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        unsigned long long hitm[P / 64], penm[P / 64], finm[P / 64];
        int nHit = 0, nPend = 0, nFin = 0;
#pragma unroll
        for (int k = 0; k < P / 64; ++k) {
            const uint32_t st = sst[lane + 64 * k];
            hitm[k] = ballot_(st == HIT);
            penm[k] = ballot_(st == PEND);
            finm[k] = ballot_(st == PATHEND || st == MISS || (st == EMPTY && pixels_left));
            nHit += __popcll(hitm[k]); nPend += __popcll(penm[k]); nFin += __popcll(finm[k]);
        }
        const int nFly = __popcll(ballot_(fly));
        if (nHit + nPend + nFin + nFly == 0) break;

        // the phase that fills the most lanes; TRAVERSE only when it can make progress (it leaves again as soon as the
        // pending list is dry, fewer than trav_min_lanes queries are in flight and other work exists)
        const bool trav_ok = nPend > 0 || nFly >= A.trav_min_lanes || (nHit + nFin == 0);
        const int cT = trav_ok ? min(nPend + nFly, 64) : -1, cS = min(nHit, 64), cF = min(nFin, 64);
        int phase;                      // 0 TRAVERSE, 1 SHADE, 2 FINISH
        if (cS >= 64) phase = 1;
        else if (cF >= 64) phase = 2;
        else if (cT >= cS && cT >= cF) phase = 0;
        else if (cS >= cF) phase = 1;
        else phase = 2;

        if (phase == 0) {
            // ================================ TRAVERSE ================================
            {
                int off = 0;
#pragma unroll
                for (int k = 0; k < P / 64; ++k) {
                    if ((penm[k] >> lane) & 1ull) list[off + __popcll(penm[k] & lt_mask)] = (unsigned char)(lane + 64 * k);
                    off += __popcll(penm[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            int pendNext = 0, other = nHit + nFin;
            v3 o = rtm::mk(0.f, 0.f, 0.f), d = rtm::mk(1.f, 1.f, 1.f);
            Hit best; best.t = INF; best.id = kNone; best.u = 0.f; best.v = 0.f;
            if (fly) {      // resume a suspended query
                o = rtm::mk(GF(OX, myslot), GF(OY, myslot), GF(OZ, myslot));
                d = rtm::mk(GF(DX, myslot), GF(DY, myslot), GF(DZ, myslot));
                best.t = GF(HT, myslot); best.id = G(HID, myslot); best.u = GF(HU, myslot); best.v = GF(HV, myslot);
            }
            RaySlab slab = make_slab(o, d);
            for (unsigned int tguard = 0; tguard < (1u << 24); ++tguard) {
                int nF = __popcll(ballot_(fly));
                const int avail = nPend - pendNext;
                if (avail > 0 && 64 - nF >= min(A.refill_min, avail)) {
                    // ---- the idle lanes take pending rays together
                    const unsigned long long idle = ballot_(!fly);
                    const int rank = __popcll(idle & lt_mask);
                    if (!fly && rank < avail) {
                        myslot = list[pendNext + rank];
                        o = rtm::mk(GF(OX, myslot), GF(OY, myslot), GF(OZ, myslot));
                        d = rtm::mk(GF(DX, myslot), GF(DY, myslot), GF(DZ, myslot));
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
                        slab = make_slab(o, d);                                     // RayBoundingBox :179
                        cur = (S.nn > 0 && ray_traceable(o, d, a)) ? 0u : kNone; sp = 0; fly = true;
                        sst[myslot] = (unsigned char)FLY;
                    }
                    pendNext += min(avail, __popcll(idle));
                }
                // ---- one while-while round: node steps, then whole leaves
                for (unsigned int nguard = 0; nguard < (1u << 20); ++nguard) {
                    const int nAtNode = __popcll(ballot2_(fly, (int)cur >= 0));
                    if (nAtNode == 0) break;
                    if (nAtNode < A.node_min && ballot_(fly && (int)cur < 0 && cur != kNone) != 0) break;   // few descenders: serve the leaves first
                    if (fly && (int)cur >= 0) {
                        if (COUNT) cnt.nodes++;
                        phase_tick<COUNT>(cnt, 0);
                        float t0, t1, t2, t3;
                        uint32_t c0, c1, c2, c3;
                        node_step<false>(S.nodes, cur, slab, best.t, F.full_sort != 0, t0, t1, t2, t3, c0, c1, c2, c3);
                        // branch-free push of the three farther children (far -> near); slots past the new top are garbage
                        stk[sp * 64] = c3; sp += (t3 < INF) ? 1 : 0;
                        stk[sp * 64] = c2; sp += (t2 < INF) ? 1 : 0;
                        stk[sp * 64] = c1; sp += (t1 < INF) ? 1 : 0;
                        if (t0 < INF) cur = c0;
                        else if (sp > 0) { --sp; cur = stk[sp * 64]; }
                        else cur = kNone;
                    }
                }
                if (fly && (int)cur < 0 && cur != kNone) {          // a leaf = kLeafBit | first << 2 | count-1
                    uint32_t ti = (cur & 0x7FFFFFFFu) >> 2;
                    const uint32_t last = ti + (cur & 3u);
                    for (; ti <= last; ++ti) {
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
                    }
                    if (sp > 0) { --sp; cur = stk[sp * 64]; }
                    else cur = kNone;
                }
                // ---- completed queries leave their lane
                const bool done = fly && cur == kNone;
                if (done) {
                    GF(HT, myslot) = best.t; G(HID, myslot) = best.id; GF(HU, myslot) = best.u; GF(HV, myslot) = best.v;
                    sst[myslot] = (unsigned char)(best.id == kNone ? MISS : HIT);
                    fly = false;
                }
                other += __popcll(ballot_(done));
                nF = __popcll(ballot_(fly));
                if (pendNext >= nPend) {
                    if (nF == 0) break;
                    if (nF < A.trav_min_lanes && other > 0) break;      // the stragglers stay suspended on their lanes
                }
            }
            // ---- suspend what is still in flight: the best hit so far goes back to the slot
            if (fly) { GF(HT, myslot) = best.t; G(HID, myslot) = best.id; GF(HU, myslot) = best.u; GF(HV, myslot) = best.v; }
        } else if (phase == 1) {
            // ================================ SHADE ================================
            {
                int off = 0;
#pragma unroll
                for (int k = 0; k < P / 64; ++k) {
                    if ((hitm[k] >> lane) & 1ull) list[off + __popcll(hitm[k] & lt_mask)] = (unsigned char)(lane + 64 * k);
                    off += __popcll(hitm[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < min(nHit, 64)) {
                const uint32_t s = list[lane];
                phase_tick<COUNT>(cnt, 2);
                if (COUNT) cnt.hits++;
                v3 o = rtm::mk(GF(OX, s), GF(OY, s), GF(OZ, s));
                v3 d = rtm::mk(GF(DX, s), GF(DY, s), GF(DZ, s));
                const float bt = GF(HT, s), bu = GF(HU, s), bv = GF(HV, s);
                const uint32_t bid = G(HID, s);
                uint32_t rng = G(RNG, s);
                const uint32_t sb = G(SB, s);
                int bounce = (int)((sb >> 12) & 0xFFFu);
                v3 light = rtm::mk(GF(LR, s), GF(LG, s), GF(LB, s));
                v3 rayColour = rtm::mk(GF(CR, s), GF(CG, s), GF(CB, s));
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
                GF(OX, s) = o.x; GF(OY, s) = o.y; GF(OZ, s) = o.z;
                GF(DX, s) = d.x; GF(DY, s) = d.y; GF(DZ, s) = d.z;
                G(RNG, s) = rng;
                GF(LR, s) = light.x; GF(LG, s) = light.y; GF(LB, s) = light.z;
                GF(CR, s) = rayColour.x; GF(CG, s) = rayColour.y; GF(CB, s) = rayColour.z;
                G(SB, s) = (sb & 0xFF000FFFu) | ((uint32_t)bounce << 12);
                sst[s] = (unsigned char)(path_done ? PATHEND : PEND);
            }
        } else {
            // ================================ FINISH ================================
            {
                int off = 0;
#pragma unroll
                for (int k = 0; k < P / 64; ++k) {
                    if ((finm[k] >> lane) & 1ull) list[off + __popcll(finm[k] & lt_mask)] = (unsigned char)(lane + 64 * k);
                    off += __popcll(finm[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const bool mine = lane < min(nFin, 64);
            uint32_t s = 0, st = DEAD, rng = 0, pxy = 0, fi = 0;
            int sample = 0;
            v3 total = rtm::mk(0.f, 0.f, 0.f);
            if (mine) {
                s = list[lane];
                st = sst[s];
                if (st != EMPTY) {
                    const uint32_t sb = G(SB, s);
                    sample = (int)(sb & 0xFFFu); fi = sb >> 24;
                    rng = G(RNG, s); pxy = G(PXY, s);
                    v3 light = rtm::mk(GF(LR, s), GF(LG, s), GF(LB, s));
                    total = rtm::mk(GF(TR, s), GF(TG, s), GF(TB, s));
                    if (st == MISS) {
                        phase_tick<COUNT>(cnt, 3);
                        const v3 d = rtm::mk(GF(DX, s), GF(DY, s), GF(DZ, s));
                        const v3 rayColour = rtm::mk(GF(CR, s), GF(CG, s), GF(CB, s));
                        light = light + environment_light(p, d) * rayColour;       // :346-347
                    }
                    total = total + light;                                         // frag :384
                    ++sample;
                    if (sample >= p.numRaysPerPixel) {
                        // ---- pixel complete: frag :387-388 + Accumulate.shader:45-50
                        const float n = (float)p.numRaysPerPixel;
                        const float cx = total.x / n, cy = total.y / n, cz = total.z / n;
                        const size_t pi = (size_t)(pxy >> 16) * W + (pxy & 0xFFFFu);
                        F.out_frame[(size_t)fi * F.frame_stride + pi] = make_float4(cx, cy, cz, 1.0f);
                        if (F.frames_in_launch <= 1) {
                            const float4 prev = F.accum[pi];
                            float4 acc;
                            acc.x = rtm::saturate(prev.x * omw + cx * weight);
                            acc.y = rtm::saturate(prev.y * omw + cy * weight);
                            acc.z = rtm::saturate(prev.z * omw + cz * weight);
                            acc.w = rtm::saturate(prev.w * omw + 1.0f * weight);
                            F.accum[pi] = acc;
                        }
                        st = EMPTY;
                    }
                }
            }
            // ---- pixel refill from the work queue: item = (frame, tile) in costliest-first order, pixel by pixel
            bool have_pixel = mine && st != EMPTY;
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
                        const unsigned int item = idx >> 6, within = idx & 63u;
                        const unsigned int f = item / ntiles;
                        unsigned int tile = item - f * ntiles;
                        if (F.tile_order) tile = F.tile_order[tile];
                        const int x = (int)(tile % (unsigned)F.tiles_x) * 8 + (int)(within & 7u);
                        const int yy = (int)(tile / (unsigned)F.tiles_x) * 8 + (int)(within >> 3);
                        if (x < p.width && yy < F.nrows) {
                            pxy = (uint32_t)x | ((uint32_t)yy << 16); fi = f;
                            rng = ((uint32_t)(F.row0 + (yy >> 3) * F.row_stride + (yy & 7)) * W + (uint32_t)x) + (uint32_t)(F.frame + (int)f) * 719393u;   // :361-362
                            sample = 0;
                            total = rtm::mk(0.f, 0.f, 0.f);
                            st = PEND;      // "has a pixel"
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
                    const int px = (int)(pxy & 0xFFFFu), ly = (int)(pxy >> 16);
                    const int y = F.row0 + (ly >> 3) * F.row_stride + (ly & 7);
                    const float uvx = ((float)px + 0.5f) / cam.W, uvy = ((float)y + 0.5f) / (float)(uint32_t)p.height;
                    const float lx = (uvx - 0.5f) * p.viewParams[0], lyv = (uvy - 0.5f) * p.viewParams[1], lz = 1.0f * p.viewParams[2];
                    cam.focusPoint = rtm::mk(((M[0] * lx + M[1] * lyv) + M[2]  * lz) + M[3]  * 1.0f,
                                             ((M[4] * lx + M[5] * lyv) + M[6]  * lz) + M[7]  * 1.0f,
                                             ((M[8] * lx + M[9] * lyv) + M[10] * lz) + M[11] * 1.0f);
                    v3 o, d;
                    camera_ray(p, cam, rng, o, d);
                    GF(OX, s) = o.x; GF(OY, s) = o.y; GF(OZ, s) = o.z;
                    GF(DX, s) = d.x; GF(DY, s) = d.y; GF(DZ, s) = d.z;
                    G(RNG, s) = rng; G(PXY, s) = pxy;
                    GF(LR, s) = 0.f; GF(LG, s) = 0.f; GF(LB, s) = 0.f;
                    GF(CR, s) = 1.f; GF(CG, s) = 1.f; GF(CB, s) = 1.f;
                    GF(TR, s) = total.x; GF(TG, s) = total.y; GF(TB, s) = total.z;
                    G(SB, s) = (uint32_t)sample | (fi << 24);
                    sst[s] = (unsigned char)PEND;
                } else {
                    sst[s] = (unsigned char)(st == DEAD ? DEAD : EMPTY);
                }
            }
        }
    }
#undef G
#undef GF
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
