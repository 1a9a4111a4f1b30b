// RtNative.cs — P/Invoke binding of include/rt.h (librt_mi355x.so), complete: every exported entry point and every struct.
//
// Drop into Assets/Scripts/Native/ of the reference project together with RtBackend.cs, and apply RayTracingManager.cs.ed (the line
// edits that make the reference's own RayTracingManager call RtBackend instead of its material blits; INTEGRATION.md).
// The structs below are plain sequential layouts with primitive fields and fixed buffers only (no UnityEngine types), so that
// tests/test_csharp_binding_cpu.py can parse them and check every field offset against the C headers' layouts
// (this image has no C# toolchain: the file is shipped as source and verified structurally).
//
// The reference's own GPU-buffer structs (Assets/Scripts/Data Types/: RayTracingMaterial 64 B, Sphere 80 B, Triangle 72 B,
// MeshInfo 96 B) are blittable with exactly the strides the library expects, so managed arrays of them are pinned and
// passed as they are (the IntPtr overloads); RtMaterial / RtSphere / RtTriangle / RtMeshInfo are their UnityEngine-free twins.
using System;
using System.Runtime.InteropServices;

namespace RtMi355x
{
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtMaterial                 // rt_material, 64 B — RayTracingMaterial.cs:13-19
    {
        public fixed float colour[4];
        public fixed float emissionColour[4];
        public fixed float specularColour[4];
        public float emissionStrength;
        public float smoothness;
        public float specularProbability;
        public int flag;                            // 0 None, 1 CheckerPattern, 2 InvisibleLight
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtSphere                   // rt_sphere, 80 B — Sphere.cs:5-7
    {
        public fixed float position[3];
        public float radius;
        public RtMaterial material;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtTriangle                 // rt_triangle, 72 B — Triangle.cs:8-14
    {
        public fixed float posA[3];
        public fixed float posB[3];
        public fixed float posC[3];
        public fixed float normalA[3];
        public fixed float normalB[3];
        public fixed float normalC[3];
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtMeshInfo                 // rt_meshinfo, 96 B — MeshInfo.cs:5-9
    {
        public uint firstTriangleIndex;
        public uint numTriangles;
        public RtMaterial material;
        public fixed float boundsMin[3];
        public fixed float boundsMax[3];
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtMeshTransform            // rt_mesh_transform, 40 B — transform.position / rotation / lossyScale
    {
        public fixed float position[3];
        public fixed float rotation[4];             // x, y, z, w
        public fixed float lossyScale[3];
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct RtLocalChunk                      // rt_local_chunk, 80 B — one MeshChunk of RayTracedMesh.localChunks
    {
        public uint firstTriangleIndex;
        public uint numTriangles;
        public uint meshIndex;
        public uint _reserved;
        public RtMaterial material;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtParams                   // rt_params — everything RayTracingManager pushes with Material.Set*
    {
        public int width;
        public int height;
        public int maxBounceCount;
        public int numRaysPerPixel;
        public float defocusStrength;
        public float divergeStrength;
        public fixed float viewParams[3];           // planeWidth, planeHeight, focusDistance
        public fixed float camLocalToWorld[16];     // row-major m[row * 4 + col]
        public fixed float worldSpaceCameraPos[3];
        public fixed float worldSpaceLightPos0[3];
        public int environmentEnabled;
        public fixed float groundColour[4];
        public fixed float skyColourHorizon[4];
        public fixed float skyColourZenith[4];
        public float sunFocus;
        public float sunIntensity;
        public int rngMode;                         // RngMode
        public int intersectMode;                   // IntersectMode
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtStats                    // rt_stats
    {
        public int numRenderedFrames;
        public int numMeshChunks;
        public int numTriangles;
        public int numSpheres;
        public int numBvhNodes;
        public int bvhMaxStack;
        public ulong rays;
        public ulong sphereTests;
        public ulong nodeVisits;
        public ulong triTests;
        public ulong hits;
        public fixed ulong phaseLanes[5];
        public fixed ulong phaseExecs[5];
        public double lastKernelMs;
        public double totalKernelMs;
        public double lastGeometryMs;
        public double lastDisplayMs;
        public int lastFramesPerLaunch;
        public int autoKernel;
        public int lastKernel;
        public int lastFramesInterleaved;
        public double lastBvhBuildMs;
        public float refitAreaRatio;
        public float bvhInternalArea;
        public int bvhBuiltOnDevice;
        public int bvhBuilds;
        public int bvhRebuilds;
        public int bvhRepads;
        public int lastSampleLanes;
        public int queuedLaunches;
        public fixed ulong regionExecs[32];
        public fixed uint primaryLists[4];
        public int primaryListBuilds;
        public int _reserved;
        public double lastPrimaryListsMs;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct RtMultiInfo                // rt_multi_info
    {
        public int numContexts;
        public int bvhBuilds;
        public double lastSetupMs;
        public double lastGatherMs;
        public fixed int device[16];
        public fixed int peerAccess[16];
    }

    public enum RngMode { Pcg = 0, Philox = 1 }
    public enum IntersectMode { FlatChunks = 0, Brute = 1 }

    public static class RtNative
    {
        const string Lib = "rt_mi355x";             // librt_mi355x.so on the library path

        // ---- lifetime / errors
        [DllImport(Lib)] public static extern IntPtr rt_create(int device);
        [DllImport(Lib)] public static extern void rt_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_set_stream(IntPtr ctx, IntPtr hipStream);
        // ---- uniforms and buffers
        [DllImport(Lib)] public static extern int rt_set_params(IntPtr ctx, ref RtParams p);
        [DllImport(Lib)] public static extern int rt_upload_spheres(IntPtr ctx, IntPtr spheres, int n);
        [DllImport(Lib)] public static extern int rt_upload_triangles(IntPtr ctx, IntPtr tris, int n);
        [DllImport(Lib)] public static extern int rt_upload_meshinfo(IntPtr ctx, IntPtr meshinfo, int n);
        [DllImport(Lib)] public static extern int rt_upload_local_meshes(IntPtr ctx, IntPtr localTris, int nTris, IntPtr chunks, int nChunks, int nMeshes);
        [DllImport(Lib)] public static extern int rt_set_mesh_transforms(IntPtr ctx, IntPtr transforms, int nMeshes);
        [DllImport(Lib)] public static extern int rt_read_world_geometry(IntPtr ctx, IntPtr trisOut, int nTris, IntPtr meshinfoOut, int nChunks);
        [DllImport(Lib, CharSet = CharSet.Ansi)] public static extern int rt_set_option(IntPtr ctx, string name, int value);
        [DllImport(Lib)] public static extern int rt_set_rows(IntPtr ctx, int row0, int nrows);
        [DllImport(Lib)] public static extern int rt_set_bands(IntPtr ctx, int firstBand, int bandStride);
        // ---- rendering
        [DllImport(Lib)] public static extern int rt_render_frame(IntPtr ctx, int frameIndex);
        [DllImport(Lib)] public static extern int rt_render(IntPtr ctx, int firstFrame, int nFrames);
        [DllImport(Lib)] public static extern int rt_render_counting(IntPtr ctx, int firstFrame, int nFrames);
        [DllImport(Lib)] public static extern int rt_render_frame_flat(IntPtr ctx, int frameIndex);
        [DllImport(Lib)] public static extern int rt_reset_accum(IntPtr ctx);
        [DllImport(Lib)] public static extern int rt_submit_frame(IntPtr ctx, int frameIndex);
        [DllImport(Lib)] public static extern int rt_wait(IntPtr ctx);
        // ---- read-back / restore
        [DllImport(Lib)] public static extern int rt_read_accum(IntPtr ctx, IntPtr rgba, UIntPtr nFloats);
        [DllImport(Lib)] public static extern int rt_read_last_frame(IntPtr ctx, IntPtr rgba, UIntPtr nFloats);
        [DllImport(Lib)] public static extern int rt_write_accum(IntPtr ctx, IntPtr rgba, UIntPtr nFloats, int framesRendered);
        [DllImport(Lib)] public static extern int rt_copy_accum_to_device(IntPtr ctx, IntPtr dstDevicePtr, UIntPtr nFloats);
        [DllImport(Lib)] public static extern int rt_read_display(IntPtr ctx, IntPtr rgba8, UIntPtr nPixels);
        [DllImport(Lib)] public static extern int rt_read_bvh(IntPtr ctx, IntPtr nodesF32, IntPtr nodesF16, UIntPtr nNodes);
        [DllImport(Lib)] public static extern int rt_get_stats(IntPtr ctx, out RtStats stats);
        // ---- ABI self-description
        [DllImport(Lib)] public static extern int rt_abi_version();
        [DllImport(Lib, CharSet = CharSet.Ansi)] public static extern int rt_sizeof(string structName);
        // ---- several GPUs behind one handle
        [DllImport(Lib)] public static extern IntPtr rt_multi_create(int[] devices, int nDevices);
        [DllImport(Lib)] public static extern void rt_multi_destroy(IntPtr multi);
        [DllImport(Lib)] public static extern IntPtr rt_multi_last_error(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_count(IntPtr multi);
        [DllImport(Lib)] public static extern IntPtr rt_multi_context(IntPtr multi, int i);
        [DllImport(Lib)] public static extern int rt_multi_set_params(IntPtr multi, ref RtParams p);
        [DllImport(Lib)] public static extern int rt_multi_upload_spheres(IntPtr multi, IntPtr spheres, int n);
        [DllImport(Lib)] public static extern int rt_multi_upload_triangles(IntPtr multi, IntPtr tris, int n);
        [DllImport(Lib)] public static extern int rt_multi_upload_meshinfo(IntPtr multi, IntPtr meshinfo, int n);
        [DllImport(Lib, CharSet = CharSet.Ansi)] public static extern int rt_multi_set_option(IntPtr multi, string name, int value);
        [DllImport(Lib)] public static extern int rt_multi_reset_accum(IntPtr multi);
        [DllImport(Lib)] public static extern int rt_multi_render(IntPtr multi, int firstFrame, int nFrames);
        [DllImport(Lib)] public static extern int rt_multi_read_accum(IntPtr multi, IntPtr rgba, UIntPtr nFloats);
        [DllImport(Lib)] public static extern int rt_multi_get_stats(IntPtr multi, out RtStats stats, out double gatherMs);
        [DllImport(Lib)] public static extern int rt_multi_get_info(IntPtr multi, out RtMultiInfo info);
        [DllImport(Lib)] public static extern int rt_multi_upload_local_meshes(IntPtr multi, IntPtr localTris, int nTris, IntPtr chunks, int nChunks, int nMeshes);
        [DllImport(Lib)] public static extern int rt_multi_set_mesh_transforms(IntPtr multi, IntPtr transforms, int nMeshes);
        [DllImport(Lib)] public static extern int rt_multi_read_display(IntPtr multi, IntPtr rgba8, UIntPtr nPixels);
        [DllImport(Lib)] public static extern int rt_multi_write_accum(IntPtr multi, IntPtr rgba, UIntPtr nFloats, int framesRendered);

        // ---- helpers --------------------------------------------------------------------------------------------------
        public static string LastError(IntPtr ctx) { return Marshal.PtrToStringAnsi(rt_last_error(ctx)) ?? ""; }
        public static string LastMultiError(IntPtr multi) { return Marshal.PtrToStringAnsi(rt_multi_last_error(multi)) ?? ""; }

        public static void Check(IntPtr ctx, int rc, string what)
        {
            if (rc != 0) throw new InvalidOperationException(what + " failed (" + rc + "): " + LastError(ctx));
        }
        public static void CheckMulti(IntPtr multi, int rc, string what)
        {
            if (rc != 0) throw new InvalidOperationException(what + " failed (" + rc + "): " + LastMultiError(multi));
        }

        // The library copies on upload, so a managed array only has to stay pinned for the duration of the call.
        public delegate int UploadCall(IntPtr ctx, IntPtr data, int n);
        public static void Upload<T>(IntPtr ctx, UploadCall call, T[] items, int count, string what) where T : struct
        {
            if (items == null || count == 0) { Check(ctx, call(ctx, IntPtr.Zero, 0), what); return; }
            GCHandle pin = GCHandle.Alloc(items, GCHandleType.Pinned);
            try { Check(ctx, call(ctx, pin.AddrOfPinnedObject(), count), what); }
            finally { pin.Free(); }
        }

        // Struct sizes as the loaded library sees them; throws when this file and the library disagree.
        public static void VerifyLayout()
        {
            void Same(string name, int managed)
            {
                int native = rt_sizeof(name);
                if (native != managed) throw new InvalidOperationException("ABI mismatch: sizeof(" + name + ") = " + native + " in librt_mi355x, " + managed + " in RtNative.cs");
            }
            Same("rt_material", Marshal.SizeOf<RtMaterial>());
            Same("rt_sphere", Marshal.SizeOf<RtSphere>());
            Same("rt_triangle", Marshal.SizeOf<RtTriangle>());
            Same("rt_meshinfo", Marshal.SizeOf<RtMeshInfo>());
            Same("rt_mesh_transform", Marshal.SizeOf<RtMeshTransform>());
            Same("rt_local_chunk", Marshal.SizeOf<RtLocalChunk>());
            Same("rt_params", Marshal.SizeOf<RtParams>());
            Same("rt_stats", Marshal.SizeOf<RtStats>());
            Same("rt_multi_info", Marshal.SizeOf<RtMultiInfo>());
        }
    }
}
