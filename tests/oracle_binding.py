"""ctypes binding of oracle/librt_oracle.so — the CPU checker.  Test infrastructure only: nothing under the
product package imports this."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_size_t, c_uint32, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "librt_oracle.so")

COUNTS = np.dtype([("rays", "<u8"), ("sphereTests", "<u8"), ("boxTests", "<u8"), ("triTests", "<u8"),
                   ("hits", "<u8"), ("threads", "<i4"), ("_pad", "<i4")])


def build_oracle():
    src = os.path.join(ORACLE_DIR, "rt_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB


class Oracle:
    def __init__(self, lib=None):
        """lib: another build of the same source (oracle/Makefile `variants`: libm intrinsics, FMA contraction) for the sensitivity
        study of tools/oracle_sensitivity.py; default = the parity oracle"""
        import rtx_pkg
        self.rtx = rtx_pkg.load()
        self.lib = ctypes.CDLL(lib or build_oracle())
        L = self.lib
        L.orc_render_frame.restype = c_int
        L.orc_render_frame.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                       c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_void_p]
        L.orc_accumulate.restype = None
        L.orc_accumulate.argtypes = [POINTER(c_float), POINTER(c_float), c_size_t, c_int]
        L.orc_next_random.restype = c_uint32
        L.orc_next_random.argtypes = [POINTER(c_uint32)]
        L.orc_random_value.restype = c_float
        L.orc_random_value.argtypes = [POINTER(c_uint32)]
        for n in ("om_sin", "om_cos", "om_log", "om_exp2"):
            getattr(L, n).restype = c_float
            getattr(L, n).argtypes = [c_float]
        for n in ("om_pow", "om_min", "om_max"):
            getattr(L, n).restype = c_float
            getattr(L, n).argtypes = [c_float, c_float]
        L.orc_display_srgb8.restype = None
        L.orc_display_srgb8.argtypes = [POINTER(c_float), c_void_p, c_size_t]
        L.orc_hw_threads.restype = c_int
        L.orc_philox_substreams.restype = c_int
        L.orc_philox_substreams.argtypes = [c_int]
        L.orc_set_accel.restype = None
        L.orc_set_accel.argtypes = [c_int]

    def render_frame(self, params, spheres, tris, meshinfo, frame, rect=None, mode=None, nthreads=0, accel=False):
        """Returns (image[h, w, 4] float32, counts dict) for pixel rect (x0, y0, x1, y1) of the full image.
        accel=True: the oracle finds triangles through its own search tree (same image bit for bit, tested; the
        box/triangle counters then describe the tree)."""
        self.lib.orc_set_accel(1 if accel else 0)
        r = self.rtx
        p = np.array(params, dtype=r.PARAMS).reshape(()).copy()
        if mode is not None:
            p["intersectMode"] = mode
        s = np.ascontiguousarray(spheres, dtype=r.SPHERE)
        t = np.ascontiguousarray(tris, dtype=r.TRIANGLE)
        m = np.ascontiguousarray(meshinfo, dtype=r.MESHINFO)
        W, H = int(p["width"]), int(p["height"])
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        out = np.empty((y1 - y0, x1 - x0, 4), np.float32)
        cnt = np.zeros((), COUNTS)
        rc = self.lib.orc_render_frame(p.ctypes.data_as(c_void_p), s.ctypes.data_as(c_void_p), len(s),
                                       t.ctypes.data_as(c_void_p), len(t), m.ctypes.data_as(c_void_p), len(m),
                                       frame, x0, y0, x1, y1, out.ctypes.data_as(POINTER(c_float)), nthreads,
                                       cnt.ctypes.data_as(c_void_p))
        if rc != 0:
            raise RuntimeError(f"orc_render_frame failed: {rc}")
        return out, {k: cnt[k].item() for k in COUNTS.names if k != "_pad"}

    def accumulate(self, accum, cur, frame):
        assert accum.dtype == np.float32 and cur.dtype == np.float32 and accum.flags.c_contiguous and cur.flags.c_contiguous
        self.lib.orc_accumulate(accum.ctypes.data_as(POINTER(c_float)), cur.ctypes.data_as(POINTER(c_float)), accum.size, frame)

    def display_srgb8(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        out = np.empty(a.shape[:-1], np.uint32)
        self.lib.orc_display_srgb8(a.ctypes.data_as(POINTER(c_float)), out.ctypes.data_as(c_void_p), out.size)
        return out.view(np.uint8).reshape(a.shape[:-1] + (4,))

    def render(self, params, spheres, tris, meshinfo, first_frame, n_frames, rect=None, mode=None, accel=False):
        """Trace + accumulate n_frames frames (OnRenderImage order); returns (accum, last_frame, counts)."""
        acc = None
        total = None
        for f in range(first_frame, first_frame + n_frames):
            cur, cnt = self.render_frame(params, spheres, tris, meshinfo, f, rect, mode, accel=accel)
            if acc is None:
                acc = np.zeros_like(cur)
            self.accumulate(acc, cur, f)
            total = cnt if total is None else {k: (total[k] + v if k != "threads" else v) for k, v in cnt.items()}
        return acc, cur, total
