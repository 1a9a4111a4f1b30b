// rt_stream.hpp — k_stream: the resumable-traversal megakernel.  Same per-pixel arithmetic as k_trace (rt_kernels.hpp),
// explicit per-lane mode (TRAV / SHADE / WAIT / DEAD) and a wave-level schedule built on ballots.
//
// k_trace binds an 8x8 tile to a wave and, per loop iteration, runs one complete closest-hit query for all 64 lanes
// before shading them: the wave pays for the longest traversal in it (measured on the 100k-triangle workload: 24 node
// steps executed per 9 needed) and for the slowest pixel of its tile.  Here
//   * traversal is *resumable*: a while-while burst ends as soon as `shade_threshold` lanes hold a complete query; the
//     stragglers keep cur / top / best hit / slab constants in registers, sit out the SHADE pass, and continue in the
//     next burst beside the other lanes' new rays — the wave no longer waits for its longest ray;
//   * inside a burst the node loop hands over to the leaf phase once fewer than `node_min` lanes still hold an internal
//     node (they wait one leaf phase) instead of running until the last descender reaches a leaf;
//   * SHADE (hit/miss shading, next bounce or next sample or next pixel, new ray, sphere loop, traversal reset) runs
//     for the lanes whose query is complete;
//   * a wave reserves `tiles_per_fetch` work items (frame, tile) per fetch — costliest tiles first — and a lane that
//     finishes its pixel of one tile moves on to its position in the next tile of the group; it idles (WAIT) only when
//     the group is exhausted.  (tile_sync = 0 refills lanes pixel by pixel from the global queue instead: the wave
//     loses its tile coherence and node-step utilisation drops from 37 % to 28 %.)
//
// Nothing about a pixel's own sequence of operations changes (same RNG chain, same closest-hit arithmetic, same
// tie-break), so the image is bit-identical to k_trace and to the oracle.
#pragma once
#include "rt_kernels.hpp"

namespace rtk {

struct StreamArgs {
    int shade_threshold;        // lanes waiting for SHADE that trigger it
    unsigned int total_pixels;  // tiles_x * tiles_y * 64 (tile-major enumeration, padded)
    int node_min;               // the node loop of a burst goes on while at least this many lanes hold an internal node (or no lane holds a leaf)
    int tiles_per_fetch;        // tile_sync: a wave reserves up to this many consecutive work items at a time; a lane that finishes its
                                // pixel of one moves on to its position in the next without waiting for the slower lanes
    int guide_div;              // ... as long as more than tiles_per_fetch * guide_div items are left in the launch's queue; below
                                // that the groups shrink to items_left / guide_div, down to single items (= waves of the launch x a factor)
    int tile_sync;              // 1: a wave takes a whole 8x8 tile at a time (coherent lanes), 0: lanes refill pixel by pixel
    int n16, n4, n1;            // tile_sync: the launch's frames as n16 groups of 16, then n4 groups of 4, then n1 single frames.  A work
                                // item is a sub-tile of 2x2 / 4x4 / 8x8 pixels in the 16 / 4 / 1 frames of a group: lane = (frame of the
                                // group, pixel of the sub-tile).  Frames are independent (frag :362 seeds by Frame), so the same pixel
                                // in 16 or 4 frames gives a wave rays that start almost identical and per-lane costs that are
                                // identically distributed
    int sample_lanes_log2;      // PHILOX instantiations: log2 S of the sample lanes per pixel (4 / 2 / 0 for NumRaysPerPixel >= 16 / >= 4 / else).
                                // The counter-based stream makes a pixel's samples independent, so a work item is a 2x2 / 4x4 / 8x8
                                // sub-tile of ONE frame whose pixels are spread over S lanes each: lane = (sample lane k, pixel of the
                                // sub-tile); lane k traces samples k, k + S, k + 2S, ... and the S partial sums meet in a fixed tree
                                // (the estimator of include/rt.h RT_RNG_PHILOX).  n1 = frames of the launch, n16 = n4 = 0
};

// kModeTravStrict: a traversal that evaluates the reference's chunk-box filter at every candidate (see "chunk filter" in k_stream); both
// traversal modes are the values <= 0 as signed integers: is_trav() is one compare
enum : uint32_t { kModeTrav = 0, kModeShade = 1, kModeDead = 2, kModeWait = 3, kModeTravStrict = 0xFFFFFFFFu };
__device__ __forceinline__ bool is_trav(uint32_t mode) { return (int)mode <= 0; }
constexpr uint32_t kNoPixel = 0xFFFFFFFFu;

struct StreamKernArgs { DeviceScene S; FrameArgs F; StreamArgs A; };     // k_stream's argument segment (fresh_kernargs, rt_kernels.hpp)
static_assert(alignof(StreamArgs) <= 8, "kernarg layout = struct layout");
// every by-value argument starts on the next multiple of 8 bytes in the kernarg segment; the struct view must put its members there too
// (tests/test_kernarg_layout_cpu.py reads the offsets back from the code object)
static_assert(offsetof(StreamKernArgs, F) == ((sizeof(DeviceScene) + 7) & ~size_t(7))
              && offsetof(StreamKernArgs, A) == ((offsetof(StreamKernArgs, F) + sizeof(FrameArgs) + 7) & ~size_t(7)), "kernarg layout = struct layout");

#ifndef RT_STREAM_WAVES
#define RT_STREAM_WAVES 6           // waves per SIMD of the PCG / f16-node instantiation (see below)
#endif
#ifndef RT_STREAM_WAVES_PHILOX
#define RT_STREAM_WAVES_PHILOX 5    // (six waves = 80 VGPRs: 34 of them spilled, 13.9 against 16.9 Grays/s — round 4, one box, interleaved)
#endif
constexpr int stream_waves(bool count, bool philox, bool h, bool tri) { return count ? 3 : !tri ? 6 : philox ? RT_STREAM_WAVES_PHILOX : !h ? 5 : RT_STREAM_WAVES; }     // (counting build: 47 counters in registers)
template <bool COUNT, bool PHILOX = false, bool H = false, bool TRI = true>
// Waves per SIMD.  The kernel hides its memory and LDS latencies with resident waves.  Round 2 chose five (96 VGPRs; six = 80 VGPRs spilled
// 26 dwords and lost: 14.18 against 14.70 Grays/s).  Round 4: compiled without structurizing uniform regions (__graft_entry__.py
// STREAM_TU_FLAGS) the PCG / f16-node instantiation fits 80 VGPRs WITHOUT scratch, and six waves per SIMD — with an LDS stack of <= 24
// entries per lane, so that six workgroups fit a CU — measure 18.3 against 17.1 Grays/s at five on the 100k-triangle workload, 16.2
// against 15.1 on the million-triangle one (seven: 72 VGPRs + 5 dwords of scratch, 17.2; eight: 15.1).  The Philox instantiation spills
// at 80 VGPRs (14 dwords: 13.1 against 15.8 Grays/s) and stays at five, like the f32-node and the counting instantiations.
// TRI = false: the instantiation for scenes without triangles (spheres only) — no traversal state, no burst; 76 / 86 VGPRs (PCG / Philox),
// compiled for six waves per SIMD: 36.4 -> 39.3 (PCG; k_trace's sphere instantiation stays ahead at 41.7) and 32.6 -> 35.1 Grays/s (Philox)
// on the sphere workload, eight waves (64 VGPRs, scratch): 37.6 / 29.7.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(stream_waves(COUNT, PHILOX, H, TRI), stream_waves(COUNT, PHILOX, H, TRI)))) void k_stream(DeviceScene S, FrameArgs F, StreamArgs A)
{
    extern __shared__ uint32_t lds_stack[];
    RT_MARK("begin prologue");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* stk = lds_stack + (size_t)wave * F.stack_cap * 64 + lane;
    // The LDS stack holds F.stack_cap entries per lane; when the BVH's worst case is deeper (F.gstack != null) the entries
    // past it spill to global memory ([entry - cap][lane of the launch]) — rare, but it keeps four workgroups per CU
    // resident whatever the tree depth.
    const int cap = F.stack_cap;
    uint32_t* const gstk = F.gstack ? F.gstack + (blockIdx.x * kBlock + threadIdx.x) : nullptr;
    // The stack pointer is the LDS byte address of the next free entry of this lane's column (entries 256 bytes apart): a push is a
    // ds_write at `top` and a select between top and top + 256 — no entry index to shift and add to a base.  With a global spill
    // part, `top` runs past the LDS part as a number only; (top - stk0) >> 8 is the entry index.
    using lds_u32 = __attribute__((address_space(3))) uint32_t;
    const uint32_t stk0 = (uint32_t)(uintptr_t)(lds_u32*)stk;
    uint32_t top = stk0;
    auto slot = [](uint32_t a) -> lds_u32& { return *reinterpret_cast<lds_u32*>((uintptr_t)a); };
    const uint32_t capb = (uint32_t)cap << 8;
    // pop: plain ds_read when nothing can spill (wave-uniform test); otherwise an LDS read from a clamped slot, replaced by
    // the global entry for the rare lane above the LDS part (a select between the two address spaces would turn every pop
    // into a flat load)
    auto pop = [&]() -> uint32_t {
        top -= 256u;
        if (gstk == nullptr) return slot(top);
        RT_RARE_PATH();
        const uint32_t depth = top - stk0;
        uint32_t v = slot(stk0 + min(depth, capb - 256u));
        asm volatile("" : "+v"(v));          // keep this a ds_read: do not fold it into a pointer select with the load below
        if (depth >= capb) { RT_RARE_PATH(); v = gstk[(size_t)((depth - capb) >> 8) * F.gstack_stride]; }
        return v;
    };
    // The work items of the wave's current group, decoded once per group (one lane per item) into LDS behind the stacks: per item
    // (x0 | y0 << 16) of its sub-tile's first pixel (local rows) and (first frame | log2 frames or sample lanes << 28).  A lane that
    // takes a unit of the group reads its item's entry instead of redoing the divisions of `decode` in every SHADE pass.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    using lds_u2 = __attribute__((address_space(3))) u32x2;
    lds_u2* const item_tab = reinterpret_cast<lds_u2*>((uintptr_t)__builtin_amdgcn_readfirstlane(
                                 (uint32_t)(uintptr_t)(lds_u32*)(lds_stack + (size_t)kWavesPerBlock * F.stack_cap * 64) + (uint32_t)wave * (kGroupMax * 8u)));
    Counters cnt = {};
    const float INF = __builtin_inff();
    // PHILOX: where the sums of the sub-streams of item k of the wave's group are parked: wave-private, [item][channel][position in the item]
    // (computed where it is used, from the region's own view of the arguments: no pointer held across the persistent loop)
    auto park_slot = [&](const FrameArgs& F, const StreamArgs& A, unsigned int k) -> float* {
        return F.park + (((size_t)(blockIdx.x * kWavesPerBlock) + (threadIdx.x >> 6)) * (unsigned)A.tiles_per_fetch + k) * 192u;
    };

    // ---- per-lane state -------------------------------------------------------------------------------------
    uint32_t mode = A.tile_sync ? kModeWait : kModeShade;   // every lane starts by asking for a pixel (or the wave for a tile)
    bool fresh = false;                 // the lane was just given a pixel: its first camera ray is due
    uint32_t pxy = kNoPixel;            // current pixel: x | local row << 16 (the host keeps k_stream to targets of at most 65535 x 65535); kNoPixel: none
    uint32_t rng = 0u;                  // RT_RNG_PCG: the reference's stream, a serial chain through the pixel's samples and bounces.  RT_RNG_PHILOX keeps
                                        // no generator state at all: a draw is a function of (pixel, frame, sample, bounce) — rtm::PhiloxScope
    int sample = 0, bounce = 0;         // PCG.  PHILOX keeps both in `sample` (sample | bounce << 16: the host refuses more than 65000 samples or 32000 bounces, so the signed shift stays positive) — the
                                        // generator's temporaries need the register in the scatter code
    v3 total = rtm::mk(0.f, 0.f, 0.f), light = total, rayColour = total, o = total, d = total;
    RaySlabT<H> slab = make_slab<H>(rtm::mk(0.f, 0.f, 0.f), rtm::mk(1.f, 1.f, 1.f));
    uint32_t cur = kNone;
    Hit best; best.t = INF; best.id = kNone; best.u = 0.f; best.v = 0.f;
    bool live = false;                  // a finished closest-hit query is waiting to be shaded
    unsigned long long wave_t0 = 0;
    unsigned int group_base = 0, group_len = 0;   // items [group_base, group_base + group_len) of the launch's queue belong to this wave
    unsigned int next_unit = 0;                   // ... = 64 * group_len units (item of the group << 6 | position in the item); units below this are taken.
                                                  // Wave-uniform: only ever changed in wave-uniform control flow
    unsigned int kidx = 0;                        // (per lane) this lane's unit | frame of its pixel (offset into the launch) << 16
    // A unit is what one lane works through before it needs new work: PCG — a pixel of the item with all its samples (the RNG chain);
    // PHILOX — one sub-stream (samples k, k + S, ...) of a pixel.  Units are handed out in order to whichever lanes ask (take_units):
    // the lanes at work always hold a window of consecutive units — neighbouring pixels, the same few items — and the group ends within
    // one unit's time for every lane instead of behind the lane that happened to draw the most expensive units.

    // Give this lane its position's pixel of work item `item` = (frame, tile); false when the tile has no pixel there.
    // Work items: frame group (16, 4 or 1 frames) x 8x8 tile (costliest first) x sub-tile of the tile.  The frames of a launch are
    // cut into A.n16 groups of 16, then A.n4 groups of 4, then A.n1 single frames; all their items sit in one queue, so whatever the
    // frame count the launch has one tail.
    // (the lambdas take the region's view of the arguments: all of this is scalar arithmetic)
    // item -> its 8x8 tile (before the costliest-first permutation), sub-tile, first frame and log2 of its frame count.  The queue
    // holds the single frames first, then the groups of 4, then the groups of 16 (frames n16*16 + n4*4 .., n16*16 .., 0 ..): the
    // launch ends on the cheap tiles of its most efficient items.
    auto decode = [](const FrameArgs& F, const StreamArgs& A, unsigned int item, unsigned int& tile, unsigned int& sub, unsigned int& frame0) -> int {
        const unsigned int ntiles_ = (unsigned)(F.tiles_x * F.tiles_y);
        if constexpr (PHILOX) {
            // sample lanes instead of frames: every item lies in one frame (frame0), frame-major queue
            const int sgl = A.sample_lanes_log2;
            const unsigned int per_frame = ntiles_ << sgl;
            const unsigned int g = item / per_frame, r = item - g * per_frame;
            tile = r >> sgl; sub = r & ((1u << sgl) - 1u); frame0 = g;
            return sgl;
        }
        const unsigned int items1_ = (unsigned)A.n1 * ntiles_, items4_ = (unsigned)A.n4 * (ntiles_ << 2);
        int fgl = 0; unsigned int fbase = (unsigned)A.n16 * 16u + (unsigned)A.n4 * 4u;
        if (item >= items1_) { item -= items1_; fgl = 2; fbase = (unsigned)A.n16 * 16u; if (item >= items4_) { item -= items4_; fgl = 4; fbase = 0u; } }
        const unsigned int per_group = ntiles_ << fgl;
        const unsigned int g = item / per_group, r = item - g * per_group;
        tile = r >> fgl; sub = r & ((1u << fgl) - 1u); frame0 = fbase + (g << fgl);
        return fgl;
    };
    auto pixel_index = [&](const FrameArgs& F) -> uint32_t {      // frag :360-361 of this lane's pixel (global coordinates)
        const int ly = (int)(pxy >> 16);
        return (uint32_t)(F.row0 + (ly >> 3) * F.row_stride + (ly & 7)) * (uint32_t)F.p.width + (pxy & 0xFFFFu);
    };
    // Give this lane unit `id` of the group; false when the item has no pixel there (image edge, frame count).
    auto start_pixel = [&](const FrameArgs& F, const StreamArgs& A, unsigned int id) -> bool {
        const rt_params& p = F.p;
        const uint32_t W = (uint32_t)p.width;
        const unsigned int nframes_ = (unsigned)(A.n16 * 16 + A.n4 * 4 + A.n1);
        const u32x2 e = item_tab[id >> 6];
        const unsigned int pos = id & 63u;
        const int fgl = (int)(e.y >> 28);
        const unsigned int frame0 = e.y & 0x0FFFFFFFu;
        const int pxl = 6 - fgl, swl = pxl >> 1;              // log2 of: pixels per sub-tile, sub-tile width
        const unsigned int pix = pos & ((1u << pxl) - 1u);
        const unsigned int fi = PHILOX ? frame0 : frame0 + (pos >> pxl);
        const int x = (int)((e.x & 0xFFFFu) + (pix & ((1u << swl) - 1u)));
        const int yy = (int)((e.x >> 16) + (pix >> swl));
        if (!(x < p.width && yy < F.nrows && fi < nframes_)) return false;
        pxy = (uint32_t)x | ((uint32_t)yy << 16); kidx = id | (fi << 16);
        const uint32_t pixelIndex = (uint32_t)(F.row0 + (yy >> 3) * F.row_stride + (yy & 7)) * W + (uint32_t)x;
        if constexpr (PHILOX) sample = (int)(pos >> pxl);                                    // (bounce = 0 in the high half) this unit's sub-stream: samples k, k + S, ... (S <= NumRaysPerPixel)
        else { rng = pixelIndex + (uint32_t)(F.frame + (int)fi) * 719393u; sample = 0; }     // :361-362
        total = rtm::mk(0.f, 0.f, 0.f);
        live = false;
        return true;
    };
    // Lanes with `want` take the next units of the group, in lane order; true for a lane that got one.  Call in wave-uniform control
    // flow only (next_unit must stay uniform).  Units without a pixel are skipped by asking again.
    auto take_units = [&](const FrameArgs& F, const StreamArgs& A, bool want) -> bool {
        bool got = false;
        const unsigned int total_units = group_len << 6;
        for (;;) {
            const unsigned long long need = ballot_(want);
            if (need == 0ull || next_unit >= total_units) break;
            if (want) {
                const unsigned int id = next_unit + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (id < total_units && start_pixel(F, A, id)) { want = false; got = true; }
            }
            next_unit = __builtin_amdgcn_readfirstlane(min(total_units, next_unit + (unsigned int)__popcll(need)));
        }
        return got;
    };

    RT_MARK("end prologue");
    for (;;) {
        const int nTrav = __popcll(ballot_(is_trav(mode))), nShade = __popcll(ballot_(mode == kModeShade));
        if (nTrav + nShade == 0) {
            if (!A.tile_sync) break;                                                  // every lane is dead
            RT_REGION_BEGIN(fetch);
            // ---- the whole wave is done with its group of tiles: reserve the next group (work items in LPT order)
            const StreamKernArgs& KA = fresh_kernargs<StreamKernArgs>();
            const FrameArgs& F = KA.F; const StreamArgs& A = KA.A;
            const unsigned int ntiles_ = (unsigned)(F.tiles_x * F.tiles_y);
            const unsigned int nitems_ = PHILOX ? (unsigned)A.n1 * (ntiles_ << A.sample_lanes_log2)
                                                : (unsigned)A.n1 * ntiles_ + (unsigned)A.n4 * (ntiles_ << 2) + (unsigned)A.n16 * (ntiles_ << 4);
            if constexpr (PHILOX) {
                // ---- the estimator's tree (include/rt.h RT_RNG_PHILOX): every lane parked the sum of its sub-stream for each item of the group;
                // the S sample lanes of a pixel now add them pairwise across the wave — (k, k + 1), then (k, k + 2), ... : lane offsets
                // 2^pxl, 2^(pxl+1), ... — and sample lane 0 divides by NumRaysPerPixel and stores the pixel (frag :387-388, Accumulate)
                const int pxl = 6 - A.sample_lanes_log2, swl = pxl >> 1;
                const uint32_t W = (uint32_t)F.p.width;
                const float nf = (float)F.p.numRaysPerPixel;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every lane's parked sums have left the wave
                for (unsigned int k = 0; k < group_len; ++k) {
                    const u32x2 e = item_tab[k];
                    const unsigned int fi = e.y & 0x0FFFFFFFu;
                    const unsigned int pix = (unsigned)lane & ((1u << pxl) - 1u);
                    const int x = (int)((e.x & 0xFFFFu) + (pix & ((1u << swl) - 1u)));
                    const int yy = (int)((e.x >> 16) + (pix >> swl));
                    const bool present = x < F.p.width && yy < F.nrows;
                    float tx = 0.f, ty = 0.f, tz = 0.f;
                    if (present) {          // (parked by whichever lane worked the unit: read past the L1)
                        const float* q = park_slot(F, A, k) + lane;
                        tx = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ty = __hip_atomic_load(q + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tz = __hip_atomic_load(q + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    for (int off = 1 << pxl; off < 64; off <<= 1) {
                        tx = tx + __shfl_xor(tx, off, 64); ty = ty + __shfl_xor(ty, off, 64); tz = tz + __shfl_xor(tz, off, 64);
                    }
                    if (present && ((unsigned)lane >> pxl) == 0u) {
                        const float cx = tx / nf, cy = ty / nf, cz = tz / nf;
                        const size_t pi = (size_t)yy * W + (uint32_t)x;
                        F.out_frame[(size_t)fi * F.frame_stride + pi] = make_float4(cx, cy, cz, 1.0f);
                        if (F.frames_in_launch <= 1) {
                            const float weight = 1.0f / (float)(F.frame + 1);              // Accumulate.shader:48
                            const float omw = 1.0f - weight;
                            const float4 prev = F.accum[pi];
                            float4 acc;
                            acc.x = rtm::saturate(prev.x * omw + cx * weight);
                            acc.y = rtm::saturate(prev.y * omw + cy * weight);
                            acc.z = rtm::saturate(prev.z * omw + cz * weight);
                            acc.w = rtm::saturate(prev.w * omw + 1.0f * weight);
                            F.accum[pi] = acc;
                        }
                    }
                }
            }
            uint32_t slot = (uint32_t)lane;
            asm volatile("" : "+v"(slot));                  // (keeps the table address out of the persistent loop's registers)
            if (F.tile_cost && group_len != 0 && slot < group_len) {
                const uint32_t share = (uint32_t)(((__builtin_readcyclecounter() - wave_t0) >> 6) / group_len);
                const u32x2 e = item_tab[slot];                                        // lane k: item k of the group
                atomicAdd(&F.tile_cost[((e.x >> 16) >> 3) * (unsigned)F.tiles_x + ((e.x & 0xFFFFu) >> 3)], share);
            }
            // guided self-scheduling: groups of up to tiles_per_fetch items while plenty of work is left (lanes flow from one item to
            // the next instead of idling behind the item's slowest pixel), single items near the end of the launch (balance)
            unsigned int base = 0, K = 1;
            if (lane == 0) {
                const unsigned int handed = __hip_atomic_load(F.tile_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 6;
                const unsigned int rem = nitems_ > handed ? nitems_ - handed : 0u;
                K = min((unsigned)A.tiles_per_fetch, max(1u, rem / (unsigned)A.guide_div));
                base = atomicAdd(F.tile_counter, 64u * K);
            }
            base = __builtin_amdgcn_readfirstlane(base);
            K = __builtin_amdgcn_readfirstlane(K);
            group_base = base >> 6;
            if (group_base >= nitems_) break;
            group_len = min(K, nitems_ - group_base);
            wave_t0 = __builtin_readcyclecounter();
            if (slot < group_len) {                                                    // lane k decodes item k of the group
                unsigned int tile, sub, frame0;
                const int fgl = decode(F, A, group_base + slot, tile, sub, frame0);
                const int swl = (6 - fgl) >> 1;
                if (F.tile_order) tile = F.tile_order[tile];
                const unsigned int x0 = (tile % (unsigned)F.tiles_x) * 8u + ((sub & ((1u << (3 - swl)) - 1u)) << swl);
                const unsigned int y0 = (tile / (unsigned)F.tiles_x) * 8u + ((sub >> (3 - swl)) << swl);
                item_tab[slot] = u32x2{ x0 | (y0 << 16), frame0 | ((unsigned)fgl << 28) };
            }
            next_unit = 0;
            if (take_units(F, A, true)) { fresh = true; mode = kModeShade; }
            RT_REGION_END(fetch);
            continue;
        }

        // SHADE is due when shade_threshold lanes wait for it — of 64; when some lanes have no pixel (end of a group, end of the
        // launch) the same share of the lanes that do (48 of 64 = 3/4), or the few that are left would wait for each other's
        // longest query
        // (+0.6 % headline, +1.6 % on an eighth of the image, +0.9 % single frame against the fixed count)
        const int thr = min(A.shade_threshold, (A.shade_threshold * (nTrav + nShade) + 63) >> 6);
        if (nShade >= thr || nTrav == 0) {
            // ================================ SHADE ================================
            // Wave priority: a wave in a traversal burst alternates short VALU runs with loads it then waits for, a wave in SHADE is
            // one long VALU stream.  Traversing waves get the issue slots first (s_setprio 1), so their loads are in flight while the
            // shading waves fill the gaps: +4.5 % / +3.3 % on the two triangle workloads (0/0: 12.86, trav 1 / shade 0: 13.44,
            // trav 0 / shade 1: 12.91, node loop 2 / leaves 1 / shade 0: 13.44 Grays/s).
            __builtin_amdgcn_s_setprio(0);
            RT_REGION_BEGIN(shade);
            const StreamKernArgs& KA = fresh_kernargs<StreamKernArgs>();
            const DeviceScene& S = KA.S; const FrameArgs& F = KA.F; const StreamArgs& A = KA.A;
            const rt_params& p = F.p;
            const float* M = p.camLocalToWorld;
            const uint32_t W = (uint32_t)p.width;
            bool need_ray = false;                      // a camera ray must be generated
            bool want = false;                          // this lane finished its unit and takes the next one of the group
            bool retrace = false;                       // the answer of this lane's query failed the chunk filter: same ray again, strictly
            if (mode == kModeShade) {
                need_ray = fresh;
                fresh = false;
                bool path_done = false;
                // ---- chunk filter (RT_INTERSECT_FLAT_CHUNKS, the literal result): the reference tests a triangle only when RayBoundingBox of
                // its chunk passes (:279).  A chunk's box contains its triangles, so the test can only fail by rounding (or for boxes that
                // were uploaded too tight) — evaluating it for every candidate of every ray cost 9 % of the frame.  Instead the traversal
                // takes the closest triangle over ALL chunks and the test is made once, here, for the answer: if it passes, the answer is
                // the closest admissible hit as well (the minimum over a superset that lies in the subset); if not, the ray is traced
                // again with the filter at every candidate (kModeTravStrict) — the old behaviour, for the rare ray that needs it.
                if constexpr (TRI) {
                    if (live && best.id != kNone && (best.id & kTriBit) && p.intersectMode == RT_INTERSECT_FLAT_CHUNKS) {
                        RT_REGION_BEGIN(verify);
                        const uint32_t chunk = __float_as_uint(S.tri_nrm[(size_t)(best.id & ~kTriBit) * 3].w);
                        const float4 bmn = S.chunk_box[(size_t)chunk * 2], bmx = S.chunk_box[(size_t)chunk * 2 + 1];
                        if (!ray_bounding_box(o, slab.inv, rtm::mk(bmn.x, bmn.y, bmn.z), rtm::mk(bmx.x, bmx.y, bmx.z))) { retrace = true; live = false; }
                        RT_REGION_END(verify);
                    }
                }
                if (live) {
                    if (best.id != kNone) {
                        // ---- hit: Trace :309-343
                        RT_REGION_BEGIN(hit);
                        phase_tick<COUNT>(cnt, 2);
                        if (COUNT) cnt.hits++;
                        const v3 hitPoint = o + d * best.t;
                        v3 normal; const float4* mat;
                        if (best.id & kTriBit) {
                            const uint32_t ti = best.id & ~kTriBit;
                            const float4* tn = S.tri_nrm + (size_t)ti * 3;
                            const float4 n0 = tn[0], n1 = tn[1], n2 = tn[2];
                            const float w = 1.0f - best.u - best.v;
                            normal = rtm::normalize((rtm::mk(n0.x, n0.y, n0.z) * w + rtm::mk(n1.x, n1.y, n1.z) * best.u)
                                                    + rtm::mk(n2.x, n2.y, n2.z) * best.v);
                            mat = S.chunk_mat + (size_t)__float_as_uint(n0.w) * 4;
                        } else {
                            RT_REGION_BEGIN(hit_sphere);
                            const float4 s = S.sph_geom[best.id];
                            normal = rtm::normalize(hitPoint - rtm::mk(s.x, s.y, s.z));
                            mat = S.sph_mat + (size_t)best.id * 4;
                            RT_REGION_END(hit_sphere);
                        }
                        const float4 mcol = mat[0], memi = mat[1], mprm = mat[3];      // (specularColour: loaded where it is used, below)
                        const int flag = (int)__float_as_uint(mprm.w);
                        v3 colour = rtm::mk(mcol.x, mcol.y, mcol.z);
                        bool skip = false;
                        if (flag == 1) {                                               // CheckerPattern :313-317
                            RT_REGION_BEGIN(hit_checker);
                            float cx = mod2(__builtin_floorf(hitPoint.x)), cz = mod2(__builtin_floorf(hitPoint.z));
                            if (!(cx == cz)) colour = rtm::mk(memi.x, memi.y, memi.z);
                            RT_REGION_END(hit_checker);
                        } else if (flag == 2 && (PHILOX ? (sample >> 16) : bounce) == 0) {   // InvisibleLightSource :318-322
                            o = hitPoint + d * 0.001f;
                            skip = true;
                        }
                        if (!skip) {
                            RT_REGION_BEGIN(hit_scatter);
                            auto scatter = [&](auto& R) {
                                const bool isSpecular = mprm.z >= rtm::random_value(R);    // :325
                                const float specF = isSpecular ? 1.0f : 0.0f;
                                o = hitPoint;                                              // :327
                                v3 diffuseDir = rtm::normalize(normal + rtm::random_direction(R));
                                const float4 mspec = mat[2];           // after the six draws of the direction: three registers fewer across them
                                v3 specularDir = rtm::reflect(d, normal);
                                d = rtm::normalize(rtm::lerp(diffuseDir, specularDir, mprm.y * specF));
                                v3 emitted = rtm::mk(memi.x, memi.y, memi.z) * mprm.x;     // :333-335
                                light = light + emitted * rayColour;
                                rayColour = rayColour * rtm::lerp(colour, rtm::mk(mspec.x, mspec.y, mspec.z), specF);
                                float pr = rtm::fmax_(rayColour.x, rtm::fmax_(rayColour.y, rayColour.z));   // :338-342
                                if (rtm::random_value(R) >= pr) path_done = true;
                                else { float ip = rtm::rcp_(pr); rayColour = rayColour * ip; }
                            };
                            if constexpr (PHILOX) {
                                rtm::PhiloxScope R;                                        // the eight draws of this hit: blocks 1 + 2b, 2 + 2b
                                R.begin(pixel_index(F), (uint32_t)F.frame + (kidx >> 16), (uint32_t)sample & 0xFFFFu, 1u + 2u * ((uint32_t)sample >> 16));
                                scatter(R);
                            } else scatter(rng);
                            RT_REGION_END(hit_scatter);
                        }
                        if constexpr (PHILOX) { sample += 0x10000; if ((sample >> 16) > p.maxBounceCount) path_done = true; }
                        else { ++bounce; if (bounce > p.maxBounceCount) path_done = true; }   // loop bound :305
                        RT_REGION_END(hit);
                    } else {
                        RT_REGION_BEGIN(env);
#if !defined(RT_DIAG_IDLE) && !defined(RT_DIAG_PRIMARY) && !defined(RT_DIAG_TOP)
                        phase_tick<COUNT>(cnt, 3);
#endif
                        light = light + environment_light(p, d) * rayColour;           // :346-347
                        path_done = true;
                        RT_REGION_END(env);
                    }
                    if (path_done) {
                        total = total + light;                                         // :384
                        if constexpr (PHILOX) sample = (sample & 0xFFFF) + (1 << A.sample_lanes_log2); else ++sample;       // (Philox: next sample of the sub-stream, bounce 0)
                        if (PHILOX && sample >= p.numRaysPerPixel) {
                            // ---- this unit (one sub-stream of a pixel) is complete: park its sum (the wave adds the sub-streams up when the
                            // group is done) and ask for the next unit
                            float* q = park_slot(F, A, (kidx & 0xFFFFu) >> 6) + (kidx & 63u);
                            q[0] = total.x; q[64] = total.y; q[128] = total.z;
                            pxy = kNoPixel; want = true;
                        } else if (sample >= p.numRaysPerPixel) {
                            // ---- pixel complete: frag :387-388 + Accumulate.shader:45-50
                            RT_REGION_BEGIN(pixel_done);
                            const float n = (float)p.numRaysPerPixel;
                            const float cx = total.x / n, cy = total.y / n, cz = total.z / n;
                            const size_t pi = (size_t)(pxy >> 16) * W + (pxy & 0xFFFFu);
                            float one = 1.0f;
                            asm volatile("" : "+v"(one));       // (or the 16-byte register tuple of this store is set up, w = 1, at kernel entry and spilled)
                            F.out_frame[(size_t)(kidx >> 16) * F.frame_stride + pi] = make_float4(cx, cy, cz, one);
                            if (F.frames_in_launch <= 1) {
                                const float weight = 1.0f / (float)(F.frame + 1);              // Accumulate.shader:48
                                const float omw = 1.0f - weight;
                                const float4 prev = F.accum[pi];
                                float4 acc;
                                acc.x = rtm::saturate(prev.x * omw + cx * weight);
                                acc.y = rtm::saturate(prev.y * omw + cy * weight);
                                acc.z = rtm::saturate(prev.z * omw + cz * weight);
                                acc.w = rtm::saturate(prev.w * omw + 1.0f * weight);
                                F.accum[pi] = acc;
                            }
                            pxy = kNoPixel; want = A.tile_sync != 0;
                            RT_REGION_END(pixel_done);
                        } else need_ray = true;
                    }
                    live = false;
                }
            }
            // ---- lanes that finished their unit take the group's next units (wave-uniform control flow); idle (WAIT) once the group has none left
            if (ballot_(want) != 0ull) {
                RT_REGION_BEGIN(take);
                const bool got = take_units(F, A, want);
                if (want) { if (got) need_ray = true; else mode = kModeWait; }
                RT_REGION_END(take);
            }
            if (mode == kModeShade) {
                // ---- pixel refill: tile-major global order; indices outside the strip are skipped
                while (!PHILOX && !A.tile_sync) {
                    const unsigned long long need = ballot_(pxy == kNoPixel && mode != kModeDead);
                    if (need == 0) break;
                    RT_REGION_BEGIN(refill);
                    if (pxy == kNoPixel && mode != kModeDead) {
                        unsigned int base = 0;
                        const int first = __builtin_ctzll(need);
                        if (lane == first) base = atomicAdd(F.tile_counter, (unsigned int)__popcll(need));
                        base = __shfl(base, first, 64);
                        const unsigned int idx = base + (unsigned int)__popcll(need & ((1ull << lane) - 1ull));
                        if (idx >= A.total_pixels) mode = kModeDead;
                        else {
                            const unsigned int tile = idx >> 6, within = idx & 63u;
                            const int x = (int)(tile % (unsigned)F.tiles_x) * 8 + (int)(within & 7u);
                            const int yy = (int)(tile / (unsigned)F.tiles_x) * 8 + (int)(within >> 3);
                            if (x < p.width && yy < F.nrows) {
                                pxy = (uint32_t)x | ((uint32_t)yy << 16);
                                const int y = F.row0 + (yy >> 3) * F.row_stride + (yy & 7);
                                rng = ((uint32_t)y * W + (uint32_t)x) + (uint32_t)F.frame * 719393u;      // :361-362
                                total = rtm::mk(0.f, 0.f, 0.f);
                                sample = 0;
                                need_ray = true;        // (numRaysPerPixel < 1 is routed to k_trace by the host)
                            }
                        }
                    }
                    RT_REGION_END(refill);
                }
                if (mode == kModeShade) {
                    if (need_ray) {
                        // ---- frag :364-382
                        RT_REGION_BEGIN(camera);
#if !defined(RT_DIAG_IDLE) && !defined(RT_DIAG_PRIMARY) && !defined(RT_DIAG_TOP)
                        phase_tick<COUNT>(cnt, 4);
#endif
                        Camera cam;
                        cam.W = (float)W;
                        cam.right = rtm::mk(M[0], M[4], M[8]);
                        cam.up    = rtm::mk(M[1], M[5], M[9]);
                        cam.pos   = ld3(p.worldSpaceCameraPos);
                        const int px = (int)(pxy & 0xFFFFu), ly = (int)(pxy >> 16);
                        const int y = F.row0 + (ly >> 3) * F.row_stride + (ly & 7);
#ifdef RT_AB_NO_FOCUS      /* A/B builds only (tools/build_variant.py) */
                        if (false) {
#else
                        if (F.focus != nullptr) {
#endif
                            // a pixel's focus point is the same for all its samples: read what k_primary_lists computed (the operations below, once)
                            const float4 fpt = F.focus[(size_t)ly * W + (uint32_t)px];
                            cam.focusPoint = rtm::mk(fpt.x, fpt.y, fpt.z);
                        } else {
                        RT_MARK("begin camera_focus");      // (wave-uniform: no cached focus points in this launch)
                        const float uvx = ((float)px + 0.5f) / cam.W, uvy = ((float)y + 0.5f) / (float)(uint32_t)p.height;
                        const float lx = (uvx - 0.5f) * p.viewParams[0], lyv = (uvy - 0.5f) * p.viewParams[1], lz = 1.0f * p.viewParams[2];
                        cam.focusPoint = rtm::mk(((M[0] * lx + M[1] * lyv) + M[2]  * lz) + M[3]  * 1.0f,
                                                 ((M[4] * lx + M[5] * lyv) + M[6]  * lz) + M[7]  * 1.0f,
                                                 ((M[8] * lx + M[9] * lyv) + M[10] * lz) + M[11] * 1.0f);
                        RT_MARK("end camera_focus");
                        }
                        if constexpr (PHILOX) {
                            rtm::PhiloxScope R;                                        // the four draws of this sample's camera ray: block 0
                            R.begin((uint32_t)y * W + (uint32_t)px, (uint32_t)F.frame + (kidx >> 16), (uint32_t)sample & 0xFFFFu, 0u);
                            camera_ray(p, cam, R, o, d, F.fixed_origin != 0);
                        } else camera_ray(p, cam, rng, o, d, F.fixed_origin != 0);
                        if constexpr (PHILOX) sample &= 0xFFFF; else bounce = 0;
                        rayColour = rtm::mk(1.f, 1.f, 1.f); light = rtm::mk(0.f, 0.f, 0.f);
                        RT_REGION_END(camera);
                    }
                    {
                        // ---- new closest-hit query: CalculateRayCollision :256-273 (spheres in buffer order)
                        RT_REGION_BEGIN(setup);
                        if (!retrace) cnt.rays++;
                        best.t = INF; best.id = kNone;
                        const float a = rtm::dot(d, d);
                        const SphereA sa = sphere_a(a);
                        for (int i = 0; i < S.ns; ++i) {
                            RT_REGION_BEGIN(setup_spheres);         // (one execution per sphere and SHADE pass)
                            const float4 s = S.sph_geom[i];
                            float dst;
                            if (COUNT && !retrace) cnt.sph++;          // (a strict second traversal is the same ray: counted once)
                            if (ray_sphere(o, d, sa, rtm::mk(s.x, s.y, s.z), s.w, dst) && dst < best.t) { best.t = dst; best.id = (uint32_t)i; }
                            RT_REGION_END(setup_spheres);
                        }
                        live = true;
                        const bool traceable = ray_traceable(o, d, a);      // NaN / zero-direction rays are complete as they stand
                        if constexpr (TRI) {
                            if (S.nn > 0 && traceable) {
                                slab = make_slab<H>(o, d);                              // RayBoundingBox :179
                                cur = 0; top = stk0; mode = retrace ? kModeTravStrict : kModeTrav;
                                // A camera ray of a pixel with a candidate list (rt_primary.hpp: every triangle a ray through the pixel's footprint can hit
                                // first lies in these <= 4 leaves) starts with the leaves on its stack instead of the root: no node step at all.
                                if (F.primary != nullptr && need_ray) {
                                    RT_REGION_BEGIN(setup_list);
                                    const uint4 L = F.primary[(size_t)(pxy >> 16) * W + (pxy & 0xFFFFu)];
                                    if (L.x != 0xFFFFFFFEu) {
                                        cur = L.x;
                                        slot(top) = L.w; top = (L.w != kNone) ? top + 256u : top;      // (branch-free, like the node step's pushes;
                                        slot(top) = L.z; top = (L.z != kNone) ? top + 256u : top;      //  the stack has three entries of slack)
                                        slot(top) = L.y; top = (L.y != kNone) ? top + 256u : top;
                                        if (L.x == kNone) mode = kModeShade;                            // nothing in the footprint's frustum: a certain miss
                                    }
                                    RT_REGION_END(setup_list);
                                }
                            }
                        }
                        RT_REGION_END(setup);
                    }
                }
            }
            RT_REGION_END(shade);
        } else {
            // ================================ TRAVERSAL BURST ================================
            if constexpr (TRI) {
            __builtin_amdgcn_s_setprio(1);
            RT_REGION_BEGIN(burst);
            // while-while over the lanes in flight: node steps until no lane holds an internal node, then every lane
            // tests its whole leaf.  The burst ends when all queries are complete, or as soon as `shade_threshold`
            // lanes wait for SHADE: the stragglers keep their traversal state and continue in the next burst.
            for (;;) {
                RT_REGION_BEGIN(burstiter);
                for (;;) {
                    RT_REGION_BEGIN(nodeloop);
                    // (cur is an internal node only while the lane traverses: every exit from kModeTrav sets cur = kNone)
                    const int nAtNode = __popcll(ballot_((int)cur >= 0));
                    if (nAtNode == 0) { RT_REGION_END(nodeloop); break; }
                    if (nAtNode < A.node_min && ballot2_(is_trav(mode), (int)cur < 0) != 0) { RT_REGION_END(nodeloop); break; }   // few descenders: serve the leaves first
#ifdef RT_DIAG_IDLE      // diagnostic build only: what the lanes that sit out a node step are waiting for (counters 3 / 4 re-used)
                    if (COUNT) {
                        if (is_trav(mode) && (int)cur < 0) cnt.phase_lanes[3]++;          // holds a leaf
                        if (mode == kModeShade) cnt.phase_lanes[4]++;                          // query complete, waits for SHADE
                        if (lane == 0) { cnt.phase_execs[3]++; cnt.phase_execs[4]++; }
                    }
#endif
                    if ((int)cur >= 0) {
                        RT_REGION_BEGIN(node);
                        if (COUNT) cnt.nodes++;
                        phase_tick<COUNT>(cnt, 0);
#ifdef RT_DIAG_TOP       // diagnostic build only (tools/diag_primary.py top): node steps at the first RT_DIAG_TOP / 4 x RT_DIAG_TOP + 1 nodes (breadth-first order)
                        if (COUNT && cur < (uint32_t)RT_DIAG_TOP) { cnt.phase_lanes[3]++; }
                        if (COUNT && cur < 4u * RT_DIAG_TOP + 1u) { cnt.phase_lanes[4]++; }
                        if (COUNT && ballot_(cur >= (uint32_t)RT_DIAG_TOP) == 0ull && (unsigned)__builtin_ctzll(ballot_(true)) == (unsigned)lane) cnt.phase_execs[3]++;
                        if (COUNT && ballot_(cur >= 4u * RT_DIAG_TOP + 1u) == 0ull && (unsigned)__builtin_ctzll(ballot_(true)) == (unsigned)lane) cnt.phase_execs[4]++;
#endif
#ifdef RT_DIAG_PRIMARY   // diagnostic build only (tools/diag_primary.py): node steps / triangle tests of camera rays (bounce 0) in counters 3 / 4; execs = steps with any such lane
                        if (COUNT && (PHILOX ? (sample >> 16) : bounce) == 0) { cnt.phase_lanes[3]++; if ((unsigned)__builtin_ctzll(ballot_(true)) == (unsigned)lane) cnt.phase_execs[3]++; }
#endif
                        float t0, t1, t2, t3;
                        uint32_t c0, c1, c2, c3;
                        node_step<H>(H ? S.nodes_h : S.nodes, cur, slab, best.t, F.full_sort != 0, t0, t1, t2, t3, c0, c1, c2, c3);
                        if (gstk == nullptr || ((void)RT_RARE_PATH_EXPR(), ballot_(top - stk0 + 768u > capb) == 0)) {
                            // branch-free push of the three farther children (far -> near); slots past the new top are garbage
                            slot(top) = c3; top = (t3 < INF) ? top + 256u : top;
                            slot(top) = c2; top = (t2 < INF) ? top + 256u : top;
                            slot(top) = c1; top = (t1 < INF) ? top + 256u : top;
                        } else {
                            // some lane is within three entries of the LDS part: checked pushes, spilling past it
                            RT_REGION_BEGIN(node_spill);
                            auto push = [&](uint32_t c) {
                                const uint32_t depth = top - stk0;
                                if (depth < capb) slot(top) = c; else gstk[(size_t)((depth - capb) >> 8) * F.gstack_stride] = c;
                                top += 256u;
                            };
                            if (t3 < INF) push(c3);
                            if (t2 < INF) push(c2);
                            if (t1 < INF) push(c1);
                            RT_REGION_END(node_spill);
                        }
                        if (t0 < INF) cur = c0;
                        else {
                            RT_REGION_BEGIN(node_pop);
                            if (top != stk0) cur = pop();
                            else { cur = kNone; mode = kModeShade; }
                            RT_REGION_END(node_pop);
                        }
                        RT_REGION_END(node);
                    }
                    RT_REGION_END(nodeloop);
                }
                if (is_trav(mode) && (int)cur < 0) {            // a leaf = kLeafBit | first << 2 | count-1
                    RT_REGION_BEGIN(leaf);
                    uint32_t ti = (cur & 0x7FFFFFFFu) >> 2;
                    const uint32_t last = ti + (cur & 3u);
                    if (cur != kNone)           // (an empty child slot can never be entered by a traceable ray; never decode one)
                    for (; ti <= last; ++ti) {
                        RT_REGION_BEGIN(tri);
                        float4 g0, g1, g2;
                        load_tri(S.tri_geo, ti, g0, g1, g2);
                        float dst, u, v;
                        if (COUNT) cnt.tris++;
                        phase_tick<COUNT>(cnt, 1);
#ifdef RT_DIAG_PRIMARY
                        if (COUNT && (PHILOX ? (sample >> 16) : bounce) == 0) { cnt.phase_lanes[4]++; if ((unsigned)__builtin_ctzll(ballot_(true)) == (unsigned)lane) cnt.phase_execs[4]++; }
#endif
                        const bool hit = ray_triangle(o, d, rtm::mk(g0.x, g0.y, g0.z), rtm::mk(g0.w, g1.x, g1.y),
                                                      rtm::mk(g1.z, g1.w, g2.x), rtm::mk(g2.y, g2.z, g2.w), dst, u, v);
                        if (hit && dst <= best.t) {
                            RT_REGION_BEGIN(tri_accept);
                            bool take = dst < best.t;
                            if (!take && (best.id & kTriBit) && best.id != kNone) {
                                // equal dst: the reference keeps the triangle that comes first in the buffer
                                RT_REGION_BEGIN(tri_tie);
                                uint32_t oc = __float_as_uint(S.tri_nrm[(size_t)ti * 3 + 1].w);
                                uint32_t ob = __float_as_uint(S.tri_nrm[(size_t)(best.id & ~kTriBit) * 3 + 1].w);
                                take = oc < ob;
                                RT_REGION_END(tri_tie);
                            }
                            if (take && mode == kModeTravStrict) {
                                // the reference only reaches this triangle if its chunk's box test passes (:279) — evaluated here only on the
                                // second, strict traversal of a ray whose first answer failed it in SHADE (see "chunk filter" there)
                                RT_RARE_PATH();
                                RT_REGION_BEGIN(tri_chunk);
                                uint32_t chunk = __float_as_uint(S.tri_nrm[(size_t)ti * 3].w);
                                float4 bmn = S.chunk_box[(size_t)chunk * 2], bmx = S.chunk_box[(size_t)chunk * 2 + 1];
                                take = ray_bounding_box(o, slab.inv, rtm::mk(bmn.x, bmn.y, bmn.z), rtm::mk(bmx.x, bmx.y, bmx.z));
                                RT_REGION_END(tri_chunk);
                            }
                            if (take) { best.t = dst; best.id = kTriBit | ti; best.u = u; best.v = v; }
                            RT_REGION_END(tri_accept);
                        }
                        RT_REGION_END(tri);
                    }
                    if (top != stk0) cur = pop();
                    else { cur = kNone; mode = kModeShade; }
                    RT_REGION_END(leaf);
                }
                RT_REGION_END(burstiter);
                if (ballot_(is_trav(mode)) == 0) break;
                if ((int)__popcll(ballot_(mode == kModeShade)) >= thr) break;
            }
            RT_REGION_END(burst);
            }       // (TRI)
        }
    }
    RT_MARK("begin epilogue");
    {
        unsigned long long v[kNumCounters] = { cnt.rays, cnt.sph, cnt.nodes, cnt.tris, cnt.hits };
        for (int k = 0; k < 5; ++k) { v[5 + k] = cnt.phase_lanes[k]; v[10 + k] = cnt.phase_execs[k]; }
        for (int k = 0; k < kNumRegions; ++k) v[15 + k] = cnt.region[k];
        for (int k = 0; k < (COUNT ? kNumCounters : 1); ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) atomicAdd(&fresh_kernargs<StreamKernArgs>().F.counters[k], s);     // (read here: not held across the persistent loop)
        }
    }
    RT_MARK("end epilogue");
}

} // namespace rtk
