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
        // what each native handle (the game view's context or rt_multi, the scene view's preview context) has been sent: content that did not
        // change is not sent again, for the preview either (an upload makes the library rebuild its acceleration structure)
        sealed class Sent { public ulong spheres, triangles, meshInfo, local; public bool once; }
        readonly Dictionary<IntPtr, Sent> sent = new Dictionary<IntPtr, Sent>();
        ulong sphereHash, triangleHash, meshInfoHash;       // of the arrays above, computed when they are set
        // on-device geometry pipeline (SetMeshObjects): local chunks once, poses per frame
        bool localGeometry;
        Triangle[] localTriangles = Array.Empty<Triangle>();
        RtLocalChunk[] localChunks = Array.Empty<RtLocalChunk>();
        RtMeshTransform[] meshTransforms = Array.Empty<RtMeshTransform>();
        ulong localHash;
        readonly List<MeshChunk[]> localChunkSources = new List<MeshChunk[]>();
        public int NumMeshChunks { get; private set; }
        public int NumTriangles { get; private set; }

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
        public void SetSpheres<T>(T[] items) where T : struct { spheres = items ?? (Array)Array.Empty<T>(); sphereHash = Hash(spheres); }
        public void SetMeshes<TTri, TInfo>(List<TTri> tris, List<TInfo> infos) where TTri : struct where TInfo : struct
        {
            triangles = tris != null ? tris.ToArray() : (Array)Array.Empty<TTri>();
            meshInfo = infos != null ? infos.ToArray() : (Array)Array.Empty<TInfo>();
            triangleHash = Hash(triangles); meshInfoHash = Hash(meshInfo);
            localGeometry = false;
            NumTriangles = triangles.Length; NumMeshChunks = meshInfo.Length;
        }

        /// The on-device geometry pipeline: what the reference's own TODO asks for ("upload matrices to gpu to avoid having to contantly
        /// upload all mesh data", RayTracedMesh.cs:37).  The local chunks of every RayTracedMesh go to the library once (again only when a
        /// mesh object or its cached chunk array changes); a frame sends position / rotation / lossyScale per mesh — 40 bytes — and the
        /// library transforms, re-bounds and refits on the GPU (on every GPU of an rt_multi: no geometry crosses xGMI per frame).
        /// Needs RayTracedMesh.GetLocalChunks() (RayTracedMesh.cs.ed).  Returns false when there is nothing to trace through it.
        public unsafe bool SetMeshObjects(RayTracedMesh[] meshObjects)
        {
            if (meshObjects == null) meshObjects = Array.Empty<RayTracedMesh>();
            bool same = localGeometry && localChunkSources.Count == meshObjects.Length;
            var sources = new List<MeshChunk[]>(meshObjects.Length);
            for (int i = 0; i < meshObjects.Length; i++)
            {
                MeshChunk[] lc = meshObjects[i].GetLocalChunks();
                sources.Add(lc);
                same = same && ReferenceEquals(lc, localChunkSources[i]);
            }
            if (!same)
            {
                var tris = new List<Triangle>();
                var chunks = new List<RtLocalChunk>();
                for (int i = 0; i < meshObjects.Length; i++)
                    foreach (MeshChunk chunk in sources[i])
                    {
                        RtLocalChunk c = new RtLocalChunk { firstTriangleIndex = (uint)tris.Count, numTriangles = (uint)chunk.triangles.Length, meshIndex = (uint)i };
                        RayTracingMaterial mat = meshObjects[i].GetMaterial(chunk.subMeshIndex);
                        UnsafeUtility.CopyStructureToPtr(ref mat, UnsafeUtility.AddressOf(ref c.material));      // both are the 64-byte material
                        chunks.Add(c);
                        tris.AddRange(chunk.triangles);
                    }
                localTriangles = tris.ToArray(); localChunks = chunks.ToArray();
                localChunkSources.Clear(); localChunkSources.AddRange(sources);
                localHash = Hash(localChunks) ^ (Hash(localTriangles) * 31UL);
            }
            else
            {
                // materials can change in the inspector without the chunk arrays changing: they are part of the upload
                int k = 0;
                for (int i = 0; i < meshObjects.Length; i++)
                    foreach (MeshChunk chunk in sources[i])
                    {
                        RayTracingMaterial mat = meshObjects[i].GetMaterial(chunk.subMeshIndex);
                        UnsafeUtility.CopyStructureToPtr(ref mat, UnsafeUtility.AddressOf(ref localChunks[k].material));
                        k++;
                    }
                localHash = Hash(localChunks) ^ (Hash(localTriangles) * 31UL);
            }
            if (meshTransforms.Length != meshObjects.Length) meshTransforms = new RtMeshTransform[meshObjects.Length];
            for (int i = 0; i < meshObjects.Length; i++)
            {
                Transform t = meshObjects[i].transform;
                Vector3 pos = t.position, sc = t.lossyScale; Quaternion q = t.rotation;
                fixed (RtMeshTransform* x = &meshTransforms[i])
                {
                    x->position[0] = pos.x; x->position[1] = pos.y; x->position[2] = pos.z;
                    x->rotation[0] = q.x; x->rotation[1] = q.y; x->rotation[2] = q.z; x->rotation[3] = q.w;
                    x->lossyScale[0] = sc.x; x->lossyScale[1] = sc.y; x->lossyScale[2] = sc.z;
                }
            }
            localGeometry = true;
            NumTriangles = localTriangles.Length; NumMeshChunks = localChunks.Length;
            return true;
        }

        // ---- the blits ----------------------------------------------------------------------------------------------------
        /// "Frame" = frame, trace blit, "_Frame" = frame, accumulate blit (reference :74-81); on several devices every one renders
        /// its rows and one gather assembles resultTexture.
        public void RenderFrame(int frame)
        {
            EnsureContexts();
            Push(ctx, multi);
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
            Push(previewCtx, IntPtr.Zero);
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
            EnsureContexts();
            Push(ctx, multi);                   // (the targets must exist at the current size before a state can be written into them)
            GCHandle pin = GCHandle.Alloc(rgba, GCHandleType.Pinned);
            try
            {
                UIntPtr n = (UIntPtr)(ulong)rgba.LongLength;
                if (multi != IntPtr.Zero) RtNative.CheckMulti(multi, RtNative.rt_multi_write_accum(multi, pin.AddrOfPinnedObject(), n, framesRendered), "rt_multi_write_accum");
                else RtNative.Check(ctx, RtNative.rt_write_accum(ctx, pin.AddrOfPinnedObject(), n, framesRendered), "rt_write_accum");
            }
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
            sent.Clear();
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
        }

        // The reference re-creates and re-uploads its three buffers every frame (its own TODO at RayTracedMesh.cs:37); the library
        // rebuilds its acceleration structure on upload, so a handle is only sent what it has not got yet (hashes per handle: the game
        // view's and the scene view's preview context each keep their own record).
        void Push(IntPtr c, IntPtr m)
        {
            if (m != IntPtr.Zero) RtNative.CheckMulti(m, RtNative.rt_multi_set_params(m, ref p), "rt_multi_set_params");
            else RtNative.Check(c, RtNative.rt_set_params(c, ref p), "rt_set_params");
            IntPtr key = m != IntPtr.Zero ? m : c;
            Sent had;
            if (!sent.TryGetValue(key, out had)) { had = new Sent(); sent[key] = had; }
            if (!had.once || had.spheres != sphereHash) Upload(c, m, RtNative.rt_upload_spheres, RtNative.rt_multi_upload_spheres, spheres, "rt_upload_spheres");
            if (localGeometry)
            {
                if (!had.once || had.local != localHash || had.triangles != 0UL) UploadLocal(c, m);
                GCHandle pin = meshTransforms.Length > 0 ? GCHandle.Alloc(meshTransforms, GCHandleType.Pinned) : default(GCHandle);
                try
                {
                    IntPtr x = meshTransforms.Length > 0 ? pin.AddrOfPinnedObject() : IntPtr.Zero;
                    if (m != IntPtr.Zero) RtNative.CheckMulti(m, RtNative.rt_multi_set_mesh_transforms(m, x, meshTransforms.Length), "rt_multi_set_mesh_transforms");
                    else RtNative.Check(c, RtNative.rt_set_mesh_transforms(c, x, meshTransforms.Length), "rt_set_mesh_transforms");
                }
                finally { if (meshTransforms.Length > 0) pin.Free(); }
                had.local = localHash; had.triangles = 0UL; had.meshInfo = 0UL;
            }
            else
            {
                if (!had.once || had.triangles != triangleHash || had.local != 0UL) Upload(c, m, RtNative.rt_upload_triangles, RtNative.rt_multi_upload_triangles, triangles, "rt_upload_triangles");
                if (!had.once || had.meshInfo != meshInfoHash || had.local != 0UL) Upload(c, m, RtNative.rt_upload_meshinfo, RtNative.rt_multi_upload_meshinfo, meshInfo, "rt_upload_meshinfo");
                had.triangles = triangleHash; had.meshInfo = meshInfoHash; had.local = 0UL;
            }
            had.spheres = sphereHash; had.once = true;
        }

        void UploadLocal(IntPtr c, IntPtr m)
        {
            GCHandle pt = localTriangles.Length > 0 ? GCHandle.Alloc(localTriangles, GCHandleType.Pinned) : default(GCHandle);
            GCHandle pc = localChunks.Length > 0 ? GCHandle.Alloc(localChunks, GCHandleType.Pinned) : default(GCHandle);
            try
            {
                IntPtr t = localTriangles.Length > 0 ? pt.AddrOfPinnedObject() : IntPtr.Zero, ch = localChunks.Length > 0 ? pc.AddrOfPinnedObject() : IntPtr.Zero;
                if (m != IntPtr.Zero) RtNative.CheckMulti(m, RtNative.rt_multi_upload_local_meshes(m, t, localTriangles.Length, ch, localChunks.Length, meshTransforms.Length), "rt_multi_upload_local_meshes");
                else RtNative.Check(c, RtNative.rt_upload_local_meshes(c, t, localTriangles.Length, ch, localChunks.Length, meshTransforms.Length), "rt_upload_local_meshes");
            }
            finally { if (localTriangles.Length > 0) pt.Free(); if (localChunks.Length > 0) pc.Free(); }
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

        static unsafe ulong Hash(Array items)           // FNV-1a over the array's content, 64 bits at a time
        {
            ulong h = 14695981039346656037UL;
            if (items.Length == 0) return h;
            long bytes = (long)Marshal.SizeOf(items.GetType().GetElementType()) * items.Length;
            GCHandle pin = GCHandle.Alloc(items, GCHandleType.Pinned);
            try
            {
                byte* b = (byte*)pin.AddrOfPinnedObject();
                long words = bytes / 8;
                ulong* w = (ulong*)b;
                for (long i = 0; i < words; i++) { h ^= w[i]; h *= 1099511628211UL; }              // (eight bytes per step: every element size here is a multiple of 8)
                for (long i = words * 8; i < bytes; i++) { h ^= b[i]; h *= 1099511628211UL; }
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
