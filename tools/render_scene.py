#!/usr/bin/env python3
"""Render one of the reference's scenes (a Unity .unity file, or a tests/golden/scenes/*.npz conversion of one) on the
MI355X tracer and write what the reference shows on screen: the accumulated resultTexture after N frames
(RayTracingManager.OnRenderImage, Assets/Scripts/RayTracingManager.cs:49-93) as sRGB PNG (the back-buffer blit :84) and,
optionally, the linear RGBA32F image as OpenEXR / PFM.

    python tools/render_scene.py Assets/Scenes/Chess.unity --frames 16 --png chess.png --exr chess.exr
    python tools/render_scene.py tests/golden/scenes/Knight.npz --width 960 --height 540 --rays 16 --frames 4 --png k.png
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", help=".unity scene of the reference project, or an .npz written by unity_scene.save_scene_npz")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=16, help="frames to accumulate (spp = frames x rays per pixel)")
    ap.add_argument("--rays", type=int, default=0, help="override numRaysPerPixel of the scene's RayTracingManager")
    ap.add_argument("--bounces", type=int, default=0, help="override maxBounceCount")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--png"); ap.add_argument("--exr"); ap.add_argument("--pfm")
    args = ap.parse_args(argv)

    import rtx_pkg
    rtx = rtx_pkg.load()
    from rtx_amd import unity_scene
    tracer = rtx.Tracer(args.device)                  # raises if the HIP library or a GPU is missing: there is no CPU path
    if args.scene.endswith(".npz"):
        mgr = unity_scene.load_scene_npz(args.scene, args.width, args.height, backend=tracer)
    else:
        mgr = unity_scene.load_unity_scene(args.scene, args.width, args.height, backend=tracer)
    if args.rays:
        mgr.numRaysPerPixel = args.rays
    if args.bounces:
        mgr.maxBounceCount = args.bounces
    t0 = time.time()
    image = mgr.OnRenderImage(frames=args.frames)
    dt = time.time() - t0
    st = tracer.stats()
    print(f"{os.path.basename(args.scene)}: {args.width}x{args.height}, {mgr.numRaysPerPixel} rays/pixel x {args.frames} frames, "
          f"{mgr.maxBounceCount} bounces; {mgr.numTriangles} triangles in {mgr.numMeshChunks} chunks; "
          f"{st['rays']:,} rays in {st['totalKernelMs']:.1f} ms of kernels ({dt:.2f} s with scene upload and BVH build)")
    if args.png:
        rtx.imageio.write_png(args.png, tracer.read_display())
    if args.exr:
        rtx.imageio.write_exr(args.exr, image)
    if args.pfm:
        rtx.imageio.write_pfm(args.pfm, image)
    tracer.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
