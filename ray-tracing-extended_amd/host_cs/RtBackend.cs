// RtBackend.cs — what the reference's RayTracingManager calls instead of its two material blits.
//
// The reference reaches its tracer only through Unity's material-property API inside RayTracingManager
// (Assets/Scripts/RayTracingManager.cs:49-187: Material.Set* + ComputeBuffer uploads + three Graphics.Blit).  This class is the
// other side of that boundary for librt_mi355x.so: it takes exactly what those calls carry — the settings, the camera, the three
// arrays CreateSpheres / CreateMeshes build — through RtNative.cs and hands back resultTexture.  RayTracingManager itself stays the
// reference's file: RayTracingManager.cs.ed in this directory is the handful of line edits that swap the calls (INTEGRATION.md), so
// its serialised fields, and with them the six .unity scenes, are untouched.
//
// Not compiled in the build image of this repository (no C# toolchain, no UnityEngine): shipped as source; the struct layouts it
// shares with the library are verified by tests/test_csharp_binding_cpu.py.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using Unity.Collections;
using Unity.Collections.LowLevel.Unsafe;
using UnityEngine;

namespace RtMi355x
{
    public sealed class RtBackend : IDisposable
    {
        readonly int[] devices;
        IntPtr ctx = IntPtr.Zero, multi = IntPtr.Zero, previewCtx = IntPtr.Zero;
        RtParams p;
        int width, height;
        Texture2D presentTexture;
        Array spheres = Array.Empty<byte>(), triangles = Array.Empty<byte>(), meshInfo = Array.Empty<byte>();
        ulong sphereHash, triangleHash, meshInfoHash;
        bool uploadedOnce;

        /// <param name="devices">HIP device ordinals; more than one tiles the frame across the GPUs (8-row bands + one gather)</param>
        public RtBackend(int[] devices)
        {
            this.devices = devices != null && devices.Length > 0 ? (int[])devices.Clone() : new[] { 0 };
            RtNative.VerifyLayout();
        }

        // ---- the data the reference pushes with Material.Set* -------------------------------------------------------------
        /// InitFrame: the render target's size (_ScreenParams.xy) and the built-ins the shader reads
        /// (_WorldSpaceCameraPos, _WorldSpaceLightPos0 = -forward of the directional light).
        public unsafe void BeginFrame(Camera cam, int targetWidth, int targetHeight)
        {
            width = targetWidth; height = targetHeight;
            p.width = width; p.height = height;
            Vector3 pos = cam.transform.position;
            p.worldSpaceCameraPos[0] = pos.x; p.worldSpaceCameraPos[1] = pos.y; p.worldSpaceCameraPos[2] = pos.z;
            Light sun = RenderSettings.sun;
            if (sun == null)
                foreach (Light l in UnityEngine.Object.FindObjectsOfType<Light>())
                    if (l.type == LightType.Directional) { sun = l; break; }
            Vector3 toLight = sun != null ? -sun.transform.forward : Vector3.up;
            p.worldSpaceLightPos0[0] = toLight.x; p.worldSpaceLightPos0[1] = toLight.y; p.worldSpaceLightPos0[2] = toLight.z;
        }

        /// UpdateCameraParams: "ViewParams", "CamLocalToWorldMatrix"
        public unsafe void SetCamera(Camera cam, float planeWidth, float planeHeight, float focusDistance)
        {
            p.viewParams[0] = planeWidth; p.viewParams[1] = planeHeight; p.viewParams[2] = focusDistance;
            Matrix4x4 m = cam.transform.localToWorldMatrix;
            for (int row = 0; row < 4; row++)
                for (int col = 0; col < 4; col++)
                    p.camLocalToWorld[row * 4 + col] = m[row, col];
        }

        /// SetShaderParams: the ten uniforms.  Material.SetColor hands a shader linear values in a Linear-colour-space project
        /// (ProjectSettings: m_ActiveColorSpace 1), so the colours are converted the same way here.
        public unsafe void SetSettings(int maxBounceCount, int numRaysPerPixel, float defocusStrength, float divergeStrength,
                                       bool environmentEnabled, Color ground, Color horizon, Color zenith, float sunFocus, float sunIntensity,
                                       bool literalChunkCull = true, bool philox = false)
        {
            p.maxBounceCount = maxBounceCount; p.numRaysPerPixel = numRaysPerPixel;
            p.defocusStrength = defocusStrength; p.divergeStrength = divergeStrength;
            bool linear = QualitySettings.activeColorSpace == ColorSpace.Linear;
            Color g = linear ? ground.linear : ground, h = linear ? horizon.linear : horizon, z = linear ? zenith.linear : zenith;
            p.environmentEnabled = environmentEnabled ? 1 : 0;
            p.groundColour[0] = g.r; p.groundColour[1] = g.g; p.groundColour[2] = g.b; p.groundColour[3] = g.a;
            p.skyColourHorizon[0] = h.r; p.skyColourHorizon[1] = h.g; p.skyColourHorizon[2] = h.b; p.skyColourHorizon[3] = h.a;
            p.skyColourZenith[0] = z.r; p.skyColourZenith[1] = z.g; p.skyColourZenith[2] = z.b; p.skyColourZenith[3] = z.a;
            p.sunFocus = sunFocus; p.sunIntensity = sunIntensity;
            p.rngMode = (int)(philox ? RngMode.Philox : RngMode.Pcg);            // Pcg = the reference's stream
            p.intersectMode = (int)(literalChunkCull ? IntersectMode.FlatChunks : IntersectMode.Brute);
        }

        // ---- the three structured buffers (the reference's own blittable structs: 80 / 72 / 96 bytes, passed as they are) --------
        public void SetSpheres<T>(T[] items) where T : struct { spheres = items ?? (Array)Array.Empty<T>(); }
        public void SetMeshes<TTri, TInfo>(List<TTri> tris, List<TInfo> infos) where TTri : struct where TInfo : struct
        {
            triangles = tris != null ? tris.ToArray() : (Array)Array.Empty<TTri>();
            meshInfo = infos != null ? infos.ToArray() : (Array)Array.Empty<TInfo>();
        }

        // ---- the blits ----------------------------------------------------------------------------------------------------
        /// "Frame" = frame, trace blit, "_Frame" = frame, accumulate blit (reference :74-81); on several devices every one renders
        /// its rows and one gather assembles resultTexture.
        public void RenderFrame(int frame)
        {
            EnsureContexts();
            Push(ctx, multi, false);
            if (multi != IntPtr.Zero) RtNative.CheckMulti(multi, RtNative.rt_multi_render(multi, frame, 1), "rt_multi_render");
            else RtNative.Check(ctx, RtNative.rt_render_frame(ctx, frame), "rt_render_frame");
        }

        /// Blit(resultTexture, target)
        public void Present(RenderTexture target)
        {
            UIntPtr n = (UIntPtr)((ulong)width * (ulong)height * 4UL);
            if (multi != IntPtr.Zero) Show(target, ptr => RtNative.CheckMulti(multi, RtNative.rt_multi_read_accum(multi, ptr, n), "rt_multi_read_accum"));
            else Show(target, ptr => RtNative.Check(ctx, RtNative.rt_read_accum(ctx, ptr, n), "rt_read_accum"));
        }

        /// The scene view's Blit(null, target, rayTracingMaterial): one un-accumulated frame, in a context of its own so that the
        /// game view's accumulation is not disturbed.
        public void RenderPreview(RenderTexture target, int frame)
        {
            if (previewCtx == IntPtr.Zero) previewCtx = Create(devices[0]);
            Push(previewCtx, IntPtr.Zero, true);
            RtNative.Check(previewCtx, RtNative.rt_reset_accum(previewCtx), "rt_reset_accum");
            RtNative.Check(previewCtx, RtNative.rt_render_frame(previewCtx, frame), "rt_render_frame");
            UIntPtr n = (UIntPtr)((ulong)width * (ulong)height * 4UL);
            Show(target, ptr => RtNative.Check(previewCtx, RtNative.rt_read_last_frame(previewCtx, ptr, n), "rt_read_last_frame"));
        }

        /// RayTracingManager.Start: numRenderedFrames = 0
        public void ResetAccumulation()
        {
            if (multi != IntPtr.Zero) RtNative.CheckMulti(multi, RtNative.rt_multi_reset_accum(multi), "rt_multi_reset_accum");
            else if (ctx != IntPtr.Zero) RtNative.Check(ctx, RtNative.rt_reset_accum(ctx), "rt_reset_accum");
        }

        // ---- beyond the reference: accumulation state, statistics -----------------------------------------------------------
        public float[] SaveAccumulation()
        {
            float[] rgba = new float[(long)width * height * 4];
            GCHandle pin = GCHandle.Alloc(rgba, GCHandleType.Pinned);
            try
            {
                UIntPtr n = (UIntPtr)(ulong)rgba.LongLength;
                if (multi != IntPtr.Zero) RtNative.CheckMulti(multi, RtNative.rt_multi_read_accum(multi, pin.AddrOfPinnedObject(), n), "rt_multi_read_accum");
                else RtNative.Check(ctx, RtNative.rt_read_accum(ctx, pin.AddrOfPinnedObject(), n), "rt_read_accum");
            }
            finally { pin.Free(); }
            return rgba;
        }

        public void RestoreAccumulation(float[] rgba, int framesRendered)
        {
            if (ctx == IntPtr.Zero) throw new InvalidOperationException("RestoreAccumulation needs the single-device context (render one frame first)");
            GCHandle pin = GCHandle.Alloc(rgba, GCHandleType.Pinned);
            try { RtNative.Check(ctx, RtNative.rt_write_accum(ctx, pin.AddrOfPinnedObject(), (UIntPtr)(ulong)rgba.LongLength, framesRendered), "rt_write_accum"); }
            finally { pin.Free(); }
        }

        public RtStats Stats()
        {
            RtStats s;
            if (multi != IntPtr.Zero) { double gatherMs; RtNative.CheckMulti(multi, RtNative.rt_multi_get_stats(multi, out s, out gatherMs), "rt_multi_get_stats"); }
            else RtNative.Check(ctx, RtNative.rt_get_stats(ctx, out s), "rt_get_stats");
            return s;
        }

        /// ShaderHelper.Release of the buffers and of resultTexture (reference :190-194)
        public void Dispose()
        {
            if (multi != IntPtr.Zero) { RtNative.rt_multi_destroy(multi); multi = IntPtr.Zero; }
            if (ctx != IntPtr.Zero) { RtNative.rt_destroy(ctx); ctx = IntPtr.Zero; }
            if (previewCtx != IntPtr.Zero) { RtNative.rt_destroy(previewCtx); previewCtx = IntPtr.Zero; }
            if (presentTexture != null) { UnityEngine.Object.DestroyImmediate(presentTexture); presentTexture = null; }
            uploadedOnce = false;
        }

        // ---- internals ------------------------------------------------------------------------------------------------------
        static IntPtr Create(int device)
        {
            IntPtr c = RtNative.rt_create(device);
            if (c == IntPtr.Zero) throw new InvalidOperationException("rt_create(" + device + "): " + RtNative.LastError(IntPtr.Zero));
            return c;
        }

        void EnsureContexts()
        {
            if (ctx != IntPtr.Zero || multi != IntPtr.Zero) return;
            if (devices.Length > 1)
            {
                multi = RtNative.rt_multi_create(devices, devices.Length);
                if (multi == IntPtr.Zero) throw new InvalidOperationException("rt_multi_create: " + RtNative.LastMultiError(IntPtr.Zero));
            }
            else ctx = Create(devices[0]);
            uploadedOnce = false;
        }

        // The reference re-creates and re-uploads its three buffers every frame (its own TODO at RayTracedMesh.cs:37); the library
        // rebuilds its acceleration structure on upload, so content that did not change is not sent again.
        void Push(IntPtr c, IntPtr m, bool always)
        {
            if (m != IntPtr.Zero) RtNative.CheckMulti(m, RtNative.rt_multi_set_params(m, ref p), "rt_multi_set_params");
            else RtNative.Check(c, RtNative.rt_set_params(c, ref p), "rt_set_params");
            ulong hs = Hash(spheres), ht = Hash(triangles), hm = Hash(meshInfo);
            bool first = always || !uploadedOnce;
            if (first || hs != sphereHash) Upload(c, m, RtNative.rt_upload_spheres, RtNative.rt_multi_upload_spheres, spheres, "rt_upload_spheres");
            if (first || ht != triangleHash) Upload(c, m, RtNative.rt_upload_triangles, RtNative.rt_multi_upload_triangles, triangles, "rt_upload_triangles");
            if (first || hm != meshInfoHash) Upload(c, m, RtNative.rt_upload_meshinfo, RtNative.rt_multi_upload_meshinfo, meshInfo, "rt_upload_meshinfo");
            if (!always) { sphereHash = hs; triangleHash = ht; meshInfoHash = hm; uploadedOnce = true; }
        }

        static void Upload(IntPtr c, IntPtr m, RtNative.UploadCall single, RtNative.UploadCall many, Array items, string what)
        {
            int n = items.Length;
            GCHandle pin = n > 0 ? GCHandle.Alloc(items, GCHandleType.Pinned) : default(GCHandle);
            try
            {
                IntPtr data = n > 0 ? pin.AddrOfPinnedObject() : IntPtr.Zero;
                if (m != IntPtr.Zero) RtNative.CheckMulti(m, many(m, data, n), what);
                else RtNative.Check(c, single(c, data, n), what);
            }
            finally { if (n > 0) pin.Free(); }
        }

        static unsafe ulong Hash(Array items)           // FNV-1a over the array's bytes
        {
            ulong h = 14695981039346656037UL;
            if (items.Length == 0) return h;
            long bytes = (long)Marshal.SizeOf(items.GetType().GetElementType()) * items.Length;
            GCHandle pin = GCHandle.Alloc(items, GCHandleType.Pinned);
            try
            {
                byte* b = (byte*)pin.AddrOfPinnedObject();
                for (long i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211UL; }
            }
            finally { pin.Free(); }
            return h;
        }

        unsafe void Show(RenderTexture target, Action<IntPtr> read)
        {
            if (presentTexture == null || presentTexture.width != width || presentTexture.height != height)
            {
                if (presentTexture != null) UnityEngine.Object.DestroyImmediate(presentTexture);
                presentTexture = new Texture2D(width, height, TextureFormat.RGBAFloat, false, true) { name = "Result", filterMode = FilterMode.Bilinear };
            }
            NativeArray<float> pixels = presentTexture.GetRawTextureData<float>();
            read((IntPtr)NativeArrayUnsafeUtility.GetUnsafePtr(pixels));
            presentTexture.Apply(false, false);
            Graphics.Blit(presentTexture, target);          // row 0 of the library's image is the bottom row, as in Unity's uv space
        }
    }
}
