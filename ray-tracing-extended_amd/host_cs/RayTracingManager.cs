// RayTracingManager.cs — replaces Assets/Scripts/RayTracingManager.cs of MaxLayar/Ray-Tracing-Extended.
//
// Same class name, same serialised fields (the six .unity scenes load unchanged), same public constant; the per-pixel
// ray-trace path (RayTracing.shader + Accumulate.shader behind Material.Set* / Graphics.Blit, RayTracingManager.cs:74-81 of the
// reference) runs in librt_mi355x.so through RtNative.cs instead.  The scene components (RayTracedSphere, RayTracedMesh,
// MeshSplitter, the Data Types) are the reference's own files, untouched: their structs are blittable with the strides the
// library expects, so their arrays are pinned and handed over as they are.
//
// Not compiled in the build image of this repository (no C# toolchain, no UnityEngine): shipped as source; the layout of the
// structs it shares with the library is verified by tests/test_csharp_binding_cpu.py.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using Unity.Collections;
using UnityEngine;
using RtMi355x;

[ExecuteAlways, ImageEffectAllowedInSceneView]
public class RayTracingManager : MonoBehaviour
{
	// (kept from the reference: RayTracedMesh.GetSubMeshes reads it)
	public const int TriangleLimit = 1500;

	[Header("Ray Tracing Settings")]
	[SerializeField, Range(0, 32)] int maxBounceCount = 4;
	[SerializeField, Range(0, 64)] int numRaysPerPixel = 2;
	[SerializeField, Min(0)] float defocusStrength = 0;
	[SerializeField, Min(0)] float divergeStrength = 0.3f;
	[SerializeField, Min(0)] float focusDistance = 1;
	[SerializeField] EnvironmentSettings environmentSettings;

	[Header("View Settings")]
	[SerializeField] bool useShaderInSceneView;
	[Header("References")]
	[SerializeField] Shader rayTracingShader;                       // unused by the native path; kept so scenes keep their data
	[SerializeField, HideInInspector] Shader accumulateShader;      // "

	[Header("Info")]
	[SerializeField] int numRenderedFrames;
	[SerializeField] int numMeshChunks;
	[SerializeField] int numTriangles;

	[Header("MI355X")]
	[Tooltip("HIP device ordinals; more than one tiles the frame across the GPUs (row bands + one gather)")]
	[SerializeField] int[] devices = { 0 };
	[Tooltip("On: the reference's literal result (a triangle counts only if its chunk's box test passes). Off: closest hit over all triangles")]
	[SerializeField] bool literalChunkCull = true;

	// native state: one context, or an rt_multi when several devices are listed
	IntPtr ctx = IntPtr.Zero, multi = IntPtr.Zero, sceneViewCtx = IntPtr.Zero;
	Texture2D presentTexture;
	ulong sphereHash, triangleHash, meshInfoHash;
	bool uploadedOnce;

	List<Triangle> allTriangles;
	List<MeshInfo> allMeshInfo;

	void Start()
	{
		numRenderedFrames = 0;
		if (multi != IntPtr.Zero) RtNative.CheckMulti(multi, RtNative.rt_multi_reset_accum(multi), "rt_multi_reset_accum");
		else if (ctx != IntPtr.Zero) RtNative.Check(ctx, RtNative.rt_reset_accum(ctx), "rt_reset_accum");
	}

	// Called after any camera (e.g. game or scene camera) has finished rendering into the src texture
	void OnRenderImage(RenderTexture src, RenderTexture target)
	{
		bool isSceneCam = Camera.current.name == "SceneCamera";
		if (isSceneCam && !useShaderInSceneView)
		{
			Graphics.Blit(src, target);
			return;
		}
		int width = src.width, height = src.height;
		if (isSceneCam)
		{
			// one un-accumulated frame, as the reference's Blit(null, target, rayTracingMaterial): a context of its own keeps
			// the game view's accumulation untouched
			if (sceneViewCtx == IntPtr.Zero) sceneViewCtx = CreateContext(devices.Length > 0 ? devices[0] : 0);
			PushScene(sceneViewCtx, IntPtr.Zero, Camera.current, width, height, force: true);
			RtNative.Check(sceneViewCtx, RtNative.rt_reset_accum(sceneViewCtx), "rt_reset_accum");
			RtNative.Check(sceneViewCtx, RtNative.rt_render_frame(sceneViewCtx, numRenderedFrames), "rt_render_frame");
			Present(target, width, height, ptr => RtNative.Check(sceneViewCtx, RtNative.rt_read_last_frame(sceneViewCtx, ptr, (UIntPtr)((ulong)width * (ulong)height * 4UL)), "rt_read_last_frame"));
			return;
		}

		InitFrame(Camera.current, width, height);
		UIntPtr nFloats = (UIntPtr)((ulong)width * (ulong)height * 4UL);
		if (multi != IntPtr.Zero)
		{
			// trace + accumulate frame numRenderedFrames on every device's rows, then the one gather
			RtNative.CheckMulti(multi, RtNative.rt_multi_render(multi, numRenderedFrames, 1), "rt_multi_render");
			Present(target, width, height, ptr => RtNative.CheckMulti(multi, RtNative.rt_multi_read_accum(multi, ptr, nFloats), "rt_multi_read_accum"));
		}
		else
		{
			// "Frame" = numRenderedFrames, trace blit, "_Frame", accumulate blit  (reference :74-81)
			RtNative.Check(ctx, RtNative.rt_render_frame(ctx, numRenderedFrames), "rt_render_frame");
			Present(target, width, height, ptr => RtNative.Check(ctx, RtNative.rt_read_accum(ctx, ptr, nFloats), "rt_read_accum"));
		}
		numRenderedFrames += Application.isPlaying ? 1 : 0;
	}

	// resultTexture -> target (the reference's final Blit(resultTexture, target))
	void Present(RenderTexture target, int width, int height, Action<IntPtr> read)
	{
		if (presentTexture == null || presentTexture.width != width || presentTexture.height != height)
		{
			if (presentTexture != null) DestroyImmediate(presentTexture);
			presentTexture = new Texture2D(width, height, TextureFormat.RGBAFloat, false, true) { name = "Result", filterMode = FilterMode.Bilinear };
		}
		NativeArray<float> pixels = presentTexture.GetRawTextureData<float>();
		unsafe
		{
			read((IntPtr)Unity.Collections.LowLevel.Unsafe.NativeArrayUnsafeUtility.GetUnsafePtr(pixels));
		}
		presentTexture.Apply(false, false);
		Graphics.Blit(presentTexture, target);          // row 0 of the library's image is the bottom row, as in Unity's uv space
	}

	IntPtr CreateContext(int device)
	{
		RtNative.VerifyLayout();
		IntPtr c = RtNative.rt_create(device);
		if (c == IntPtr.Zero) throw new InvalidOperationException("rt_create(" + device + "): " + RtNative.LastError(IntPtr.Zero));
		return c;
	}

	void InitFrame(Camera cam, int width, int height)
	{
		if (ctx == IntPtr.Zero && multi == IntPtr.Zero)
		{
			if (devices != null && devices.Length > 1)
			{
				RtNative.VerifyLayout();
				multi = RtNative.rt_multi_create(devices, devices.Length);
				if (multi == IntPtr.Zero) throw new InvalidOperationException("rt_multi_create: " + RtNative.LastMultiError(IntPtr.Zero));
			}
			else ctx = CreateContext(devices != null && devices.Length == 1 ? devices[0] : 0);
			uploadedOnce = false;
		}
		PushScene(ctx, multi, cam, width, height, force: false);
	}

	// UpdateCameraParams + CreateSpheres + CreateMeshes + SetShaderParams of the reference, ending in the C-ABI's uploads
	void PushScene(IntPtr c, IntPtr m, Camera cam, int width, int height, bool force)
	{
		RtParams p = BuildParams(cam, width, height);
		if (m != IntPtr.Zero) RtNative.CheckMulti(m, RtNative.rt_multi_set_params(m, ref p), "rt_multi_set_params");
		else RtNative.Check(c, RtNative.rt_set_params(c, ref p), "rt_set_params");

		Sphere[] spheres = CreateSpheres();
		CreateMeshes();
		Triangle[] tris = allTriangles.ToArray();
		MeshInfo[] infos = allMeshInfo.ToArray();

		// the reference re-creates and re-uploads its three buffers every frame (its own TODO at RayTracedMesh.cs:37); the library
		// rebuilds its acceleration structure on upload, so unchanged content is not sent again
		ulong hs = Hash(spheres), ht = Hash(tris), hm = Hash(infos);
		bool first = force || !uploadedOnce;
		if (first || hs != sphereHash) UploadAll(c, m, RtNative.rt_upload_spheres, RtNative.rt_multi_upload_spheres, spheres, "rt_upload_spheres");
		if (first || ht != triangleHash) UploadAll(c, m, RtNative.rt_upload_triangles, RtNative.rt_multi_upload_triangles, tris, "rt_upload_triangles");
		if (first || hm != meshInfoHash) UploadAll(c, m, RtNative.rt_upload_meshinfo, RtNative.rt_multi_upload_meshinfo, infos, "rt_upload_meshinfo");
		if (!force) { sphereHash = hs; triangleHash = ht; meshInfoHash = hm; uploadedOnce = true; }
	}

	static void UploadAll<T>(IntPtr c, IntPtr m, RtNative.UploadCall single, RtNative.UploadCall many, T[] items, string what) where T : struct
	{
		if (m != IntPtr.Zero)
		{
			GCHandle pin = GCHandle.Alloc(items, GCHandleType.Pinned);
			try { RtNative.CheckMulti(m, many(m, items.Length == 0 ? IntPtr.Zero : pin.AddrOfPinnedObject(), items.Length), what); }
			finally { pin.Free(); }
		}
		else RtNative.Upload(c, single, items, items.Length, what);
	}

	static ulong Hash<T>(T[] items) where T : struct
	{
		// FNV-1a over the array's bytes
		ulong h = 14695981039346656037UL;
		if (items == null || items.Length == 0) return h;
		int bytes = Marshal.SizeOf<T>() * items.Length;
		GCHandle pin = GCHandle.Alloc(items, GCHandleType.Pinned);
		try
		{
			unsafe
			{
				byte* b = (byte*)pin.AddrOfPinnedObject();
				for (int i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211UL; }
			}
		}
		finally { pin.Free(); }
		return h;
	}

	unsafe RtParams BuildParams(Camera cam, int width, int height)
	{
		RtParams p = new RtParams();
		p.width = width;                                   // _ScreenParams.xy
		p.height = height;
		p.maxBounceCount = maxBounceCount;                 // "MaxBounceCount"
		p.numRaysPerPixel = numRaysPerPixel;               // "NumRaysPerPixel"
		p.defocusStrength = defocusStrength;               // "DefocusStrength"
		p.divergeStrength = divergeStrength;               // "DivergeStrength"

		// "ViewParams", "CamLocalToWorldMatrix"
		float planeHeight = focusDistance * Mathf.Tan(cam.fieldOfView * 0.5f * Mathf.Deg2Rad) * 2;
		float planeWidth = planeHeight * cam.aspect;
		p.viewParams[0] = planeWidth; p.viewParams[1] = planeHeight; p.viewParams[2] = focusDistance;
		Matrix4x4 l2w = cam.transform.localToWorldMatrix;
		for (int row = 0; row < 4; row++)
			for (int col = 0; col < 4; col++)
				p.camLocalToWorld[row * 4 + col] = l2w[row, col];

		// built-ins the shader reads: _WorldSpaceCameraPos, _WorldSpaceLightPos0 (= -forward of the directional light)
		Vector3 camPos = cam.transform.position;
		p.worldSpaceCameraPos[0] = camPos.x; p.worldSpaceCameraPos[1] = camPos.y; p.worldSpaceCameraPos[2] = camPos.z;
		Light sun = RenderSettings.sun;
		if (sun == null)
			foreach (Light l in FindObjectsOfType<Light>())
				if (l.type == LightType.Directional) { sun = l; break; }
		Vector3 toLight = sun != null ? -sun.transform.forward : Vector3.up;
		p.worldSpaceLightPos0[0] = toLight.x; p.worldSpaceLightPos0[1] = toLight.y; p.worldSpaceLightPos0[2] = toLight.z;

		// "EnvironmentEnabled", "GroundColour", "SkyColourHorizon", "SkyColourZenith", "SunFocus", "SunIntensity".
		// Material.SetColor hands the shader linear values in a Linear-colour-space project (ProjectSettings: m_ActiveColorSpace 1)
		bool linear = QualitySettings.activeColorSpace == ColorSpace.Linear;
		p.environmentEnabled = environmentSettings.enabled ? 1 : 0;
		Color g = linear ? environmentSettings.groundColour.linear : environmentSettings.groundColour;
		Color hz = linear ? environmentSettings.skyColourHorizon.linear : environmentSettings.skyColourHorizon;
		Color z = linear ? environmentSettings.skyColourZenith.linear : environmentSettings.skyColourZenith;
		p.groundColour[0] = g.r; p.groundColour[1] = g.g; p.groundColour[2] = g.b; p.groundColour[3] = g.a;
		p.skyColourHorizon[0] = hz.r; p.skyColourHorizon[1] = hz.g; p.skyColourHorizon[2] = hz.b; p.skyColourHorizon[3] = hz.a;
		p.skyColourZenith[0] = z.r; p.skyColourZenith[1] = z.g; p.skyColourZenith[2] = z.b; p.skyColourZenith[3] = z.a;
		p.sunFocus = environmentSettings.sunFocus;
		p.sunIntensity = environmentSettings.sunIntensity;

		p.rngMode = (int)RngMode.Pcg;                      // the reference's stream
		p.intersectMode = (int)(literalChunkCull ? IntersectMode.FlatChunks : IntersectMode.Brute);
		return p;
	}

	void CreateMeshes()
	{
		RayTracedMesh[] meshObjects = FindObjectsOfType<RayTracedMesh>();
		allTriangles ??= new List<Triangle>();
		allMeshInfo ??= new List<MeshInfo>();
		allTriangles.Clear();
		allMeshInfo.Clear();
		foreach (RayTracedMesh meshObject in meshObjects)
		{
			foreach (MeshChunk chunk in meshObject.GetSubMeshes())
			{
				RayTracingMaterial material = meshObject.GetMaterial(chunk.subMeshIndex);
				allMeshInfo.Add(new MeshInfo(allTriangles.Count, chunk.triangles.Length, material, chunk.bounds));
				allTriangles.AddRange(chunk.triangles);
			}
		}
		numMeshChunks = allMeshInfo.Count;
		numTriangles = allTriangles.Count;
	}

	Sphere[] CreateSpheres()
	{
		RayTracedSphere[] sphereObjects = FindObjectsOfType<RayTracedSphere>();
		Sphere[] spheres = new Sphere[sphereObjects.Length];
		for (int i = 0; i < sphereObjects.Length; i++)
		{
			spheres[i] = new Sphere()
			{
				position = sphereObjects[i].transform.position,
				radius = sphereObjects[i].transform.localScale.x * 0.5f,
				material = sphereObjects[i].material
			};
		}
		return spheres;
	}

	// Save / restore of the accumulation state (resultTexture + numRenderedFrames are all the reference carries between frames)
	public float[] SaveAccumulation(int width, int height)
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
		numRenderedFrames = framesRendered;
	}

	public RtStats Stats()
	{
		RtStats s;
		if (multi != IntPtr.Zero) { double gatherMs; RtNative.CheckMulti(multi, RtNative.rt_multi_get_stats(multi, out s, out gatherMs), "rt_multi_get_stats"); }
		else RtNative.Check(ctx, RtNative.rt_get_stats(ctx, out s), "rt_get_stats");
		return s;
	}

	void OnDisable()
	{
		if (multi != IntPtr.Zero) { RtNative.rt_multi_destroy(multi); multi = IntPtr.Zero; }
		if (ctx != IntPtr.Zero) { RtNative.rt_destroy(ctx); ctx = IntPtr.Zero; }
		if (sceneViewCtx != IntPtr.Zero) { RtNative.rt_destroy(sceneViewCtx); sceneViewCtx = IntPtr.Zero; }
		if (presentTexture != null) { DestroyImmediate(presentTexture); presentTexture = null; }
		uploadedOnce = false;
	}

	void OnValidate()
	{
		maxBounceCount = Mathf.Max(0, maxBounceCount);
		numRaysPerPixel = Mathf.Max(1, numRaysPerPixel);
		environmentSettings.sunFocus = Mathf.Max(1, environmentSettings.sunFocus);
		environmentSettings.sunIntensity = Mathf.Max(0, environmentSettings.sunIntensity);
	}
}
