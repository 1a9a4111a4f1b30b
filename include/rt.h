/*
 * rt.h — C-ABI of the MI355X-native path tracer (drop-in for the per-pixel ray-trace path of
 * MaxLayar/Ray-Tracing-Extended).
 *
 * The reference tracer sits behind Unity's material-property API; its only caller is
 * RayTracingManager (Assets/Scripts/RayTracingManager.cs:49-187).  Every entry point below names the
 * reference call it replaces.  All structs are plain little-endian PODs; buffer strides are the
 * reference's own Marshal.SizeOf strides (Assets/Scripts/Helpers/ShaderHelper.cs:106,124):
 *
 *   RayTracingMaterial 64 B   Assets/Scripts/Data Types/RayTracingMaterial.cs:13-19, RayTracing.shader:67-76
 *   Sphere             80 B   Assets/Scripts/Data Types/Sphere.cs:5-7,              RayTracing.shader:78-83
 *   Triangle           72 B   Assets/Scripts/Data Types/Triangle.cs:8-14,           RayTracing.shader:85-89
 *   MeshInfo           96 B   Assets/Scripts/Data Types/MeshInfo.cs:5-9,            RayTracing.shader:91-98
 *
 * No torch / C++ types cross this boundary.  Errors are int status codes (0 = ok); the message of the
 * last failure is available from rt_last_error().  Nothing throws or aborts across the ABI.
 * A context is not thread-safe (the reference is single-threaded: Unity main thread in OnRenderImage).
 * The library is GPU-only: rt_create() fails when no HIP device is present — there is no CPU fallback.
 */
#ifndef RT_H_
#define RT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- buffer element layouts (accepted verbatim from the reference host) ------------------------- */

typedef struct rt_material {            /* 64 B */
    float   colour[4];
    float   emissionColour[4];
    float   specularColour[4];
    float   emissionStrength;
    float   smoothness;
    float   specularProbability;
    int32_t flag;                       /* 0 None, 1 CheckerPattern, 2 InvisibleLight (RayTracingMaterial.cs:6-11) */
} rt_material;

typedef struct rt_sphere {              /* 80 B */
    float       position[3];
    float       radius;
    rt_material material;
} rt_sphere;

typedef struct rt_triangle {            /* 72 B */
    float posA[3], posB[3], posC[3];
    float normalA[3], normalB[3], normalC[3];
} rt_triangle;

typedef struct rt_meshinfo {            /* 96 B */
    uint32_t    firstTriangleIndex;
    uint32_t    numTriangles;
    rt_material material;
    float       boundsMin[3];
    float       boundsMax[3];
} rt_meshinfo;

/* ---- on-device geometry pipeline (optional) -----------------------------------------------------------------
 * The reference transforms every mesh to world space on the CPU and re-uploads the scene every frame
 * (RayTracedMesh.cs:36-84, RayTracingManager.cs:135-164; TODO at RayTracedMesh.cs:37).  With these two layouts the
 * local-space chunks are uploaded once and a frame only sends one transform per mesh.                       */
typedef struct rt_mesh_transform {      /* 40 B: transform.position / rotation (x,y,z,w) / lossyScale (RayTracedMesh.cs:38-40) */
    float position[3];
    float rotation[4];
    float lossyScale[3];
} rt_mesh_transform;

typedef struct rt_local_chunk {         /* 80 B: one MeshChunk of RayTracedMesh.localChunks (MeshChunk.cs:6-17) */
    uint32_t    firstTriangleIndex;     /* into the local triangle buffer                                        */
    uint32_t    numTriangles;
    uint32_t    meshIndex;              /* which rt_mesh_transform moves it                                      */
    uint32_t    _reserved;
    rt_material material;               /* RayTracedMesh.GetMaterial(chunk.subMeshIndex), RayTracedMesh.cs:96-99 */
} rt_local_chunk;

/* ---- uniforms ------------------------------------------------------------------------------------
 * One POD holding exactly what RayTracingManager pushes with Material.Set{Int,Float,Vector,Matrix,Color}
 * (RayTracingManager.cs:113-123,131-132) plus the three Unity built-ins the shader reads
 * (_ScreenParams.xy, _WorldSpaceCameraPos, _WorldSpaceLightPos0: RayTracing.shader:359,378,247).     */

enum {
    RT_RNG_PCG = 0,                     /* RayTracing.shader:193-204 — the reference's stream, the parity mode            */
    RT_RNG_PHILOX = 1                   /* the LATENCY mode: counter-based Philox4x32-10 (Salmon et al., SC'11).  NOT the reference's
                                           stream: different noise, same expectation.  A single frame finishes sooner in it (a pixel's samples
                                           spread over 16 lanes: 16.3 against 18.3 ms on the headline workload, round 4); in launches of many
                                           frames the PCG stream is the faster one (18.3 against 15.8 Grays/s: the generator costs more
                                           instructions than the chain's hash and its kernel keeps five waves per SIMD instead of six).  The reference chains one PCG state through every
                                           sample and bounce of a pixel (RayTracing.shader:362,374-385), which forbids spreading a
                                           pixel's samples over lanes; here every draw is addressed by what it is for:
                                             key     = (pixelIndex, Frame)                     (frag :360-362)
                                             counter = (block, sample, 0, 0)
                                             block 0               the sample's camera ray: words 0..3 = the four draws of frag
                                                                   :377,380 in the shader's order
                                             blocks 1+2b, 2+2b     the hit at loop index b of Trace (:305): the eight draws of
                                                                   :325-339 in the shader's order, words 0..3 of the first block,
                                                                   then of the second (a bounce that draws nothing leaves them unused)
                                           and the estimator's sum over the NumRaysPerPixel samples (:384) is a fixed tree instead of a
                                           left-to-right chain: sample s goes to sub-stream s mod S, S = 16 / 4 / 1 for NumRaysPerPixel
                                           >= 16 / >= 4 / else; a sub-stream adds its samples in increasing order starting from 0;
                                           the S sub-sums are added pairwise, (k, k+1) for even k, then (k, k+2) for k = 0 mod 4, ...;
                                           the root is divided by NumRaysPerPixel (:387).  Scatter, Russian roulette and everything
                                           else are the reference's.  On the device the S sub-streams of a pixel are S lanes of one
                                           wave (k_stream's Philox instantiation), the tree is an in-wave reduction                  */
};
enum {
    RT_INTERSECT_FLAT_CHUNKS = 0,       /* literal reference result: a triangle counts only if its chunk's
                                           RayBoundingBox test passes (RayTracing.shader:276-294).  k_stream applies the test to a
                                           ray's answer (the closest triangle over all chunks) and traces the ray again, with the
                                           test at every candidate, only if the answer fails it: same result, 8-10 % faster       */
    RT_INTERSECT_BRUTE = 1              /* no chunk cull: closest hit over all triangles                  */
};

typedef struct rt_params {
    int32_t width, height;              /* _ScreenParams.xy (render-target size)                          */
    int32_t maxBounceCount;             /* "MaxBounceCount"  (loop is inclusive: B+1 casts, shader :305)  */
    int32_t numRaysPerPixel;            /* "NumRaysPerPixel"                                             */
    float   defocusStrength;            /* "DefocusStrength"                                             */
    float   divergeStrength;            /* "DivergeStrength"                                             */
    float   viewParams[3];              /* "ViewParams" = (planeWidth, planeHeight, focusDistance)       */
    float   camLocalToWorld[16];        /* "CamLocalToWorldMatrix", row-major m[row*4+col]               */
    float   worldSpaceCameraPos[3];     /* _WorldSpaceCameraPos                                          */
    float   worldSpaceLightPos0[3];     /* _WorldSpaceLightPos0.xyz (= -forward of the directional light) */
    int32_t environmentEnabled;         /* "EnvironmentEnabled"                                          */
    float   groundColour[4];            /* "GroundColour"     (already in the space the shader sees)     */
    float   skyColourHorizon[4];        /* "SkyColourHorizon"                                            */
    float   skyColourZenith[4];         /* "SkyColourZenith"                                             */
    float   sunFocus;                   /* "SunFocus"                                                    */
    float   sunIntensity;               /* "SunIntensity"                                                */
    int32_t rngMode;                    /* RT_RNG_*                                                      */
    int32_t intersectMode;              /* RT_INTERSECT_*                                                */
} rt_params;

/* ---- statistics (the reference exposes numRenderedFrames / numMeshChunks / numTriangles,
 *      RayTracingManager.cs:25-28,156-157; the work counters are new)                                 */
typedef struct rt_stats {
    int32_t  numRenderedFrames;         /* frames accumulated so far                                     */
    int32_t  numMeshChunks;
    int32_t  numTriangles;
    int32_t  numSpheres;
    int32_t  numBvhNodes;
    int32_t  bvhMaxStack;               /* worst-case traversal stack depth of the built BVH             */
    uint64_t rays;                      /* CalculateRayCollision invocations of the last render call     */
    uint64_t sphereTests;               /* the remaining counters: rt_render_counting only               */
    uint64_t nodeVisits;                /* BVH nodes fetched                                             */
    uint64_t triTests;
    uint64_t hits;                      /* rays that hit something                                       */
    uint64_t phaseLanes[5];             /* active lanes summed over executions of phase k: 0 BVH node step,  */
    uint64_t phaseExecs[5];             /* 1 triangle test, 2 hit shading, 3 environment, 4 camera ray;      */
                                        /* lane utilisation of phase k = phaseLanes[k] / (64*phaseExecs[k])  */
    double   lastKernelMs;              /* HIP-event time of the last trace(+accumulate) launch          */
    double   totalKernelMs;             /* sum over launches since rt_reset_accum                        */
    double   lastGeometryMs;            /* HIP-event time of the last on-device transform + bounds + re-layout + refit */
    double   lastDisplayMs;             /* HIP-event time of the last linear -> sRGB8 display kernel     */
    int32_t  lastFramesPerLaunch;       /* frames traced per k_trace launch in the last rt_render (1 = frame by frame) */
    int32_t  autoKernel;                /* kernel the automatic choice picked for single-frame launches (-1 = not decided yet / not automatic) */
    int32_t  lastKernel;                /* kernel that ran the last launch: 0 k_trace, 1 k_stream, 4 flat twin */
    int32_t  lastFramesInterleaved;     /* k_stream: frames interleaved in a wave by the last launch (1, 4 or 16)        */
    double   lastBvhBuildMs;            /* last BVH build: HIP-event time of the device builder (sort + PLOC + collapse + records), */
                                        /* or host wall time of the binned-SAH builder                                   */
    float    refitAreaRatio;            /* internal box area after the last refit / right after the last build           */
    float    bvhInternalArea;           /* sum of the half-areas of all child boxes right after the last build (tree quality) */
    int32_t  bvhBuiltOnDevice;          /* 1: the current tree came from the device builder                              */
    int32_t  bvhBuilds;                 /* builds since rt_create                                                        */
    int32_t  bvhRebuilds;               /* ... of which triggered by a refit that had inflated the tree                  */
    int32_t  bvhRepads;                 /* times the box padding was widened on the device (a ray origin moved beyond the magnitude the
                                           boxes were padded for: refit of the existing tree, no rebuild, no re-upload)   */
    int32_t  lastSampleLanes;           /* Philox mode: lanes of a wave that shared a pixel's samples in the last launch (16, 4 or 1) */
    int32_t  queuedLaunches;            /* launches the rt_submit_frame queue has made since rt_reset_accum                */
    uint64_t regionExecs[32];           /* rt_render_counting, k_stream: wave-level executions of its regions, in the order of csrc/rt_kernels.hpp      */
                                        /* RT_REGION_LIST (loop, fetch, shade, hit, ...).  Multiplied with the regions' static VALU counts (from the code */
                                        /* object's assembly, tools/static_valu.py) they give the launch's VALU instruction count without a profiler:      */
                                        /* bench.py roofline.valu_model                                                                                   */
    uint32_t primaryLists[4];           /* camera rays' candidate lists of the last build (k_stream, static camera): pixels whose camera rays  */
                                        /* start from <= 4 leaves bounded by a common triangle / from an unbounded list / certainly miss         */
                                        /* everything / start at the root (no list)                                                              */
    int32_t  primaryListBuilds;         /* times the lists were built since rt_create                                                            */
    int32_t  _reserved;
    double   lastPrimaryListsMs;        /* HIP-event time of the last build of the lists                                                         */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* Lifetime.  Replaces ShaderHelper.InitMaterial / Release (ShaderHelper.cs:147-158,266-289;
 * RayTracingManager.cs:98-99,190-194).  device = HIP device ordinal.  Returns NULL on failure
 * (message retrievable with rt_last_error(NULL)).                                                      */
rt_ctx*     rt_create(int device);
void        rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);

/* Optional: run on a caller-owned hipStream_t (pass as void*).  NULL restores the context's own stream. */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);

/* Uniform upload.  Replaces SetShaderParams + UpdateCameraParams (RayTracingManager.cs:111-133).
 * Changing width/height re-creates (zeroes) the accumulation target, as ShaderHelper.CreateRenderTexture
 * does (ShaderHelper.cs:186-205).                                                                     */
int rt_set_params(rt_ctx* ctx, const rt_params* params);

/* Buffer upload = copy (caller may free immediately); n == 0 is legal.  Replaces
 * ShaderHelper.CreateStructuredBuffer + Material.SetBuffer/SetInt
 * (RayTracingManager.cs:159-163 "Triangles"/"AllMeshInfo"/"NumMeshes", :184-186 "Spheres"/"NumSpheres").
 * The library re-lays-out triangles and builds its BVH lazily at the next render.                      */
int rt_upload_spheres  (rt_ctx* ctx, const rt_sphere*   spheres,  int n);
int rt_upload_triangles(rt_ctx* ctx, const rt_triangle* tris,     int n);
int rt_upload_meshinfo (rt_ctx* ctx, const rt_meshinfo* meshinfo, int n);

/* On-device geometry pipeline.  rt_upload_local_meshes replaces rt_upload_triangles + rt_upload_meshinfo: local-space
 * triangles (reference layout, 72 B) and their chunks; rt_set_mesh_transforms sends the n_meshes poses.  The library
 * transforms to world space on the GPU (rot * Scale(p, lossyScale) + pos; normals rot * n — RayTracedMesh.cs:86-94),
 * recomputes the chunks' world AABBs (:74-82), and refits its BVH; the image is bit-identical to uploading the
 * host-transformed buffers.  rt_read_world_geometry returns what the reference's CreateMeshes would have uploaded
 * (n_tris rt_triangle, n_chunks rt_meshinfo).                                                                  */
int rt_upload_local_meshes(rt_ctx* ctx, const rt_triangle* local_tris, int n_tris,
                           const rt_local_chunk* chunks, int n_chunks, int n_meshes);
int rt_set_mesh_transforms(rt_ctx* ctx, const rt_mesh_transform* transforms, int n_meshes);
int rt_read_world_geometry(rt_ctx* ctx, rt_triangle* tris_out, int n_tris, rt_meshinfo* meshinfo_out, int n_chunks);

/* Tuning knobs; the image never depends on them (tested bitwise).
 *   "kernel"          -1 = automatic (default): the first frames after a scene / camera change time k_trace and k_stream on
 *                     ordinary frames of the render and the faster one takes the rest; 0 = tile-per-wave megakernel k_trace,
 *                     1 = k_stream (resumable traversal, stragglers deferred; the only kernel of the Philox mode)
 *   "max_leaf"        triangles per BVH leaf, 1..4 (default 2)
 *   "bvh_bins", "bvh_cost_exp", "bvh_reinsert"   BVH builder: SAH bins per axis (32); exponent, in percent, of the triangle
 *                     count in the SAH's subtree-cost model (100); passes of insertion-based optimisation of the binary tree (0:
 *                     measured -3 % node visits per ray but no fewer wave-level steps)
 *   "bvh_collapse", "bvh_node_cost"   host builder: how the binary SAH tree becomes 4-wide nodes — 0 = greedy, open the child of largest
 *                     area (default); 1 = cost-driven dynamic programme that also forms the leaves; 2 = the same over the split search's
 *                     leaves; with a node step costing bvh_node_cost percent of a triangle test (130)
 *   "stream_stack"    k_stream: traversal-stack entries per lane kept in LDS; a BVH whose worst case is deeper spills the rest to global memory.
 *                     0 (default) = automatic: 24 for the six-waves-per-SIMD instantiation (PCG stream, f16 nodes: six workgroups per CU), 30 for the
 *                     five-wave ones (Philox mode, f32 nodes, counting build)
 *   "full_sort"       1 = sort all four children of a node by entry distance, 0 = nearest first only (default)
 *   "tile_lpt"        k_trace: 1 = hand tiles out costliest first, by the costs the previous launch measured (default), 0 = in order
 *   "frame_batch"     k_trace: frames traced per launch by rt_render (0 = auto: as many as fit 4 GiB, at most 256; 1 = one per launch)
 *   "lds_stack"       k_trace: traversal-stack entries kept in LDS, deeper ones spill to global memory (0 = all in LDS)
 *   "shade_threshold" k_stream: lanes (1..64) with a complete query that end a traversal burst (default 48)
 *   "node_min"        k_stream: inside a burst the node loop goes on while at least this many lanes hold an internal node (or no
 *                     lane holds a leaf); below it the leaves are served first (default 10; 1 = classic while-while)
 *   "tiles_per_fetch" k_stream: work items a wave reserves per fetch while the launch's queue is long (1..32, default 16 = the 16 sub-tiles of
 *                     one 8x8 tile); the units of the group — a pixel's sample chain (PCG) or one sub-stream of a pixel's samples (Philox) —
 *                     are handed to whichever lanes ask, in order, so nobody waits for the lane that drew the most expensive ones
 *   "fetch_guide", "fetch_guide_philox"   k_stream: guided self-scheduling — groups of tiles_per_fetch items while more than fetch_guide
 *                     groups per wave of the launch are left in the queue, then items_left / (waves x fetch_guide), down to single items
 *                     (default 4 for the PCG stream, 1 in Philox mode, whose units are small)
 *   "tile_sync"       k_stream: 1 = a wave takes a whole 8x8 tile at a time, 0 = lanes refill pixel by pixel
 *   "stream_tile"     k_stream: frames interleaved in a wave, as log2: 0 = 8x8 pixels of one frame, 2 = 4x4 pixels x 4 frames of the
 *                     launch, 4 = 2x2 pixels x 16 frames (default; launches shorter than the group fall back to 8x8 x 1)
 *   "device_bvh"      1 = build the BVH on the device (Morton order, PLOC clustering, sweep-SAH treelet passes, breadth-first collapse to
 *                     4-wide nodes: 100k triangles in 2.5 ms, 1M in 5.4 ms, traced within about 1 % / within -4 .. +2 % of the host tree), 0 = the
 *                     host's binned-SAH builder (50 ms / 600 ms), -1 (default) = device for rt_upload_local_meshes (meshes that move) and
 *                     for a world-space scene that is uploaded again within 16 traced frames of its last build (the reference's way of
 *                     animating: everything re-sent every frame), host for the first build of a world-space scene
 *   "bvh_treelets"    device builder: number of sweep-SAH passes over the clustering's tree (default 6; 0 = none).  Pass k rebuilds, one wave
 *                     each, every maximal subtree of <= 64 * 8^k triangles over its subtrees of <= 8^k triangles (pass 0: over the triangles
 *                     themselves) with an exact sweep SAH — all three axes, every split position, large-box isolation — in the node slots the
 *                     subtree already has (csrc/rt_bvh_gpu.hpp step 3c).  "bvh_treelet_ratio" (8), "bvh_treelet_first" (1) and
 *                     "bvh_treelet_isolate" (1) tune the scale step, the first pass's item size and the isolation candidate
 *   "bvh_radius"      device builder: PLOC search radius; n > 0: n in the first rounds, doubled once a quarter and again once a sixteenth of
 *                     the clusters is left; n < 0: |n| in every round (default -16)
 *   "bvh_top"         device builder: n > 0 = once its bottom-up rounds have left at most n clusters, the top of the tree is built by the host's
 *                     binned-SAH split search over the clusters' boxes (round 3's way, default then 1024); 0 (default) = all on the device
 *   "peer_copies"     rt_multi: 1 = its device-to-device copies (scene fan-out, frame-end gather) go through hipMemcpyPeerAsync even between
 *                     contexts of one device — the branch a multi-GPU node takes, made runnable on a one-GPU box (default 0: peer API only
 *                     across devices)
 *   "rebuild_percent" on-device geometry pipeline: after a refit, rebuild on the device once the summed internal box area exceeds
 *                     this percentage of its value right after the last build (default 200; 0 = never)
 *   "compact_nodes"   k_trace / k_stream: 1 = traverse the f16 form of the BVH nodes (5 loads per node visit, default), 0 = the
 *                     f32 form (7 loads)
 *   "primary_lists"   k_stream, depth of field off: 1 (default) = every pixel's camera rays start from the <= 4 BVH leaves that can hold their
 *                     closest hit — found once per camera / scene (0.7 ms at 1080p) by tracing the corners of the pixel's jitter footprint
 *                     (csrc/rt_primary.hpp) — instead of from the root; 0 = always from the root.  The image does not depend on it.
 *   "queue_depth", "queue_linger_us"   rt_submit_frame: most frames the queue's worker puts into one launch (1..256, default 64); how long
 *                     it waits for more frames after the first one of an idle queue arrived (default 200 us: a burst becomes one launch)
 *   "blocks_per_cu"   cap on resident workgroups per CU (0 = occupancy query)                                     */
int rt_set_option(rt_ctx* ctx, const char* name, int value);

/* Row strip rendered by this context: rows [row0, row0+nrows) of the full width x height image.
 * Seeds use global pixel coordinates (RayTracing.shader:360-362) so the image is decomposition-invariant.
 * Default = the whole image.                                                                          */
int rt_set_rows(rt_ctx* ctx, int row0, int nrows);
/* Interleaved decomposition for load balance: this context renders the 8-row bands first_band, first_band +
 * band_stride, first_band + 2*band_stride, ... (band b = image rows [8b, 8b+8)); its targets hold those rows back to
 * back.  (0, 1) is the whole image.  Overrides rt_set_rows; rt_set_rows switches back to one contiguous strip.       */
int rt_set_bands(rt_ctx* ctx, int first_band, int band_stride);

/* One frame = Graphics.Blit(null, currentFrame, rayTracingMaterial) with "Frame" = frame_index, followed by
 * the Accumulate blit with "_Frame" = frame_index (RayTracingManager.cs:74-81).  Returns after the
 * stream has been synchronised.                                                                        */
int rt_render_frame(rt_ctx* ctx, int frame_index);
/* n_frames consecutive frames first_frame, first_frame+1, ... (one stream sync at the end).            */
int rt_render(rt_ctx* ctx, int first_frame, int n_frames);
/* As rt_render, but a counting build of the same kernel fills rt_stats' work counters.                 */
int rt_render_counting(rt_ctx* ctx, int first_frame, int n_frames);
/* Same frame computed by the reference's own flat loop (all spheres, all chunks, all triangles of
 * passing chunks) on the GPU — a validation/baseline path, not the fast path.                          */
int rt_render_frame_flat(rt_ctx* ctx, int frame_index);

/* Queued submission for a host that renders frame by frame, as the reference does (one trace blit + one accumulate blit per
 * OnRenderImage, RayTracingManager.cs:74-91).  rt_submit_frame returns at once; a worker thread of the library traces what has queued
 * up while the previous launch ran — consecutive frame indices share ONE launch (frame-interleaved work items, one launch tail: the
 * batched rate instead of the single-frame rate) — and accumulates in submission order, so the image equals rt_render's over the same
 * frames bit for bit.  rt_wait returns when everything submitted is in resultTexture; every other call on the context waits first, so
 * reading, changing the camera or uploading between submissions is always safe.  An error of a queued launch is reported (once) by
 * the next call that waits.  Option "queue_depth": most frames per launch (default 64).                                        */
int rt_submit_frame(rt_ctx* ctx, int frame_index);
int rt_wait(rt_ctx* ctx);

/* Zero the accumulation target and the frame counter (RayTracingManager.Start, :43-46).                */
int rt_reset_accum(rt_ctx* ctx);

/* Read back RGBA32F rows of this context's strip (nrows*width*4 floats).  accum = resultTexture,
 * last_frame = currentFrame (RayTracingManager.cs:33,75).                                             */
int rt_read_accum     (rt_ctx* ctx, float* rgba, size_t n_floats);
int rt_read_last_frame(rt_ctx* ctx, float* rgba, size_t n_floats);
/* Restore a saved accumulation state: resultTexture + numRenderedFrames are all the reference carries from frame to frame
 * (RayTracingManager.cs:26,33; the counter restarts in Start, :43-46).  rgba = rows*width*4 floats as returned by
 * rt_read_accum; the next frame to render is frame index frames_rendered (frames are independent given their index,
 * RayTracing.shader:362, so a resumed render equals an uninterrupted one bit for bit).                                  */
int rt_write_accum(rt_ctx* ctx, const float* rgba, size_t n_floats, int frames_rendered);
/* Device-side copy of the strip into caller-owned device memory (e.g. a torch tensor handed to RCCL). */
int rt_copy_accum_to_device(rt_ctx* ctx, void* dst_device_ptr, size_t n_floats);

/* Display step after the path: resultTexture converted linear -> sRGB, 8 bits per channel (R | G<<8 | B<<16 | A<<24),
 * as the final Blit(resultTexture, target) into the sRGB back buffer does (RayTracingManager.cs:84;
 * ProjectSettings.asset:50).  n_pixels = rows*width of this context's strip; row 0 is the bottom row.             */
int rt_read_display(rt_ctx* ctx, uint32_t* rgba8, size_t n_pixels);

int rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* Diagnostics: the acceleration structure the library built behind the reference's flat chunk list (the reference has none:
 * RayTracing.shader:276-294 loops over all chunks) — n_nodes = rt_stats.numBvhNodes records of 128 bytes each, in the f32
 * form (six plane arrays of 4 floats, child[4], meta[4]) and in the f16 form the kernels traverse (csrc/bvh.hpp Node4h).
 * Either destination may be NULL.  Tests check that every f16 box contains its f32 box.                                 */
int rt_read_bvh(rt_ctx* ctx, void* nodes_f32, void* nodes_f16, size_t n_nodes);

/* ---- several GPUs of one node behind one handle ---------------------------------------------------------------------
 * The reference renders on one GPU; its path shards into independent pixels (seed = global pixel index + Frame * 719393,
 * RayTracing.shader:360-362; Accumulate.shader is per pixel), so the frame tiles across devices by rows.  An rt_multi owns one
 * rt_ctx per entry of `devices` (a device may appear more than once: several contexts on one GPU, which is how the single-GPU
 * tests exercise this path).  Scene and uniforms are replicated (rt_multi_set_params / rt_multi_upload_* replace the same
 * RayTracingManager calls as their single-device forms); context i renders the 8-row bands b with b % N == i
 * (rt_set_bands(i, N)); rt_multi_render runs the N contexts concurrently for all n_frames and ends with the path's only
 * exchange: one gather of the accumulated strips to the first device (N - 1 peer copies over xGMI, each issued on its source
 * context's stream so that the links run concurrently, joined by events on the first device's stream + a row scatter).  The
 * assembled image is bit-identical to a single-context render (tested).  rt_multi_context gives the per-device context for
 * options, statistics and per-strip read-back.                                                                         */
typedef struct rt_multi rt_multi;
rt_multi*   rt_multi_create(const int* devices, int n_devices);
void        rt_multi_destroy(rt_multi* m);
const char* rt_multi_last_error(const rt_multi* m);
int         rt_multi_count(const rt_multi* m);
rt_ctx*     rt_multi_context(rt_multi* m, int i);
int rt_multi_set_params      (rt_multi* m, const rt_params* params);
int rt_multi_upload_spheres  (rt_multi* m, const rt_sphere*   spheres,  int n);
int rt_multi_upload_triangles(rt_multi* m, const rt_triangle* tris,     int n);
int rt_multi_upload_meshinfo (rt_multi* m, const rt_meshinfo* meshinfo, int n);
/* The on-device geometry pipeline behind the handle (rt_upload_local_meshes / rt_set_mesh_transforms for every context): the local
 * meshes go to every device once, a frame sends the poses (40 B per mesh, RayTracedMesh.cs:36-84: the reference moves meshes every
 * frame) and every device transforms, builds and refits its own copy — no geometry crosses xGMI per frame.                       */
int rt_multi_upload_local_meshes(rt_multi* m, const rt_triangle* local_tris, int n_tris, const rt_local_chunk* chunks, int n_chunks, int n_meshes);
int rt_multi_set_mesh_transforms(rt_multi* m, const rt_mesh_transform* transforms, int n_meshes);
int rt_multi_set_option      (rt_multi* m, const char* name, int value);
int rt_multi_reset_accum     (rt_multi* m);
int rt_multi_render          (rt_multi* m, int first_frame, int n_frames);
/* the assembled resultTexture: height*width*4 floats, row 0 = bottom */
int rt_multi_read_accum      (rt_multi* m, float* rgba, size_t n_floats);
/* the assembled image through the display step (rt_read_display's twin: linear -> sRGB8 on the first device; height*width pixels) */
int rt_multi_read_display    (rt_multi* m, uint32_t* rgba8, size_t n_pixels);
/* restore a saved accumulation state (rt_write_accum's twin): the whole image in, every context takes the rows of its bands */
int rt_multi_write_accum     (rt_multi* m, const float* rgba, size_t n_floats, int frames_rendered);
/* rays and work counters summed over the contexts, kernel times = the slowest context's; gather_ms (may be NULL) = wall
 * time of the last gather (copies + scatter)                                                                            */
int rt_multi_get_stats       (rt_multi* m, rt_stats* out, double* gather_ms);
/* How the handle is set up and what a scene change cost: the uploads go to the first context only, which builds the scene once;
 * the other contexts receive the built scene device to device (over xGMI between GPUs) — bvhBuilds counts the builds of ALL
 * contexts since rt_multi_create (one per scene change, whatever N).                                                          */
typedef struct rt_multi_info {
    int32_t numContexts;
    int32_t bvhBuilds;                  /* BVH builds summed over the contexts since rt_multi_create                        */
    double  lastSetupMs;                /* host wall time of the last scene change: build on the first context + fan-out   */
    double  lastGatherMs;               /* host wall time of the last gather: first copy submitted -> image assembled    */
    int32_t device[16];                 /* HIP ordinal of context i (the first 16)                                         */
    int32_t peerAccess[16];             /* 1: the first context's device and context i's read each other's memory directly */
                                        /* (copies go GPU to GPU over xGMI), 0: the runtime stages them; [0] = 1            */
} rt_multi_info;
int rt_multi_get_info        (rt_multi* m, rt_multi_info* out);

/* ABI self-description for binding generators / tests. */
int rt_abi_version(void);
int rt_sizeof(const char* struct_name);   /* "rt_material" | "rt_sphere" | "rt_triangle" | "rt_meshinfo" | "rt_params" | "rt_stats" | "rt_mesh_transform" | "rt_local_chunk" | "rt_multi_info" */

#ifdef __cplusplus
}
#endif
#endif /* RT_H_ */
