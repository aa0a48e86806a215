#!/usr/bin/env python
"""Drive a saved context image (NativeEngine.save -> es_ctx_load) WITHOUT torch or any model code: ctypes on
libedgestyle_hip.so and the HIP runtime only.  Used by tests/test_native_gpu.py in a child process, and a template for a
non-Python host: the calls below are everything a C program would make.

    python tests/run_ctx_image.py <image.esctx> <inputs.npz> <outputs.npz>
"""
import ctypes as C
import os
import sys

import numpy as np


def main(image_path, in_path, out_path):
    assert "torch" not in sys.modules
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip = C.CDLL("libamdhip64.so")
    lib = C.CDLL(os.path.join(root, "edgestyle_amd", "lib", "libedgestyle_hip.so"))
    lib.es_last_error.restype = C.c_char_p
    P = C.c_void_p

    def ok(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {(lib.es_last_error() or b'').decode()}")

    def dev(a):
        p = P()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, P(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0        # host -> device
        return p

    def host(p, shape, dtype):
        a = np.empty(shape, dtype=dtype)
        assert hip.hipMemcpy(P(a.ctypes.data), p, C.c_size_t(a.nbytes), 2) == 0        # device -> host
        return a

    assert hip.hipSetDevice(0) == 0
    z = np.load(in_path)
    n = int(z["n_conds"])
    ctx = P()
    lib.es_ctx_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(P)]
    ok(lib.es_ctx_load(image_path.encode(), 0, C.byref(ctx)), "es_ctx_load")
    imgs = (P * n)(*[dev(np.ascontiguousarray(z[f"img{i}"])) for i in range(n)])
    noise = (P * n)(*[dev(np.ascontiguousarray(z[f"noise{i}"])) if f"noise{i}" in z else None for i in range(n)])
    lib.es_prepare_conds.argtypes = [P, C.POINTER(P), C.POINTER(P), P]
    ok(lib.es_prepare_conds(ctx, imgs, noise, None), "es_prepare_conds")
    lat = dev(np.ascontiguousarray(z["latents"]))
    ehs = dev(np.ascontiguousarray(z["ehs"]))
    ts = np.ascontiguousarray(z["timesteps"], dtype=np.float32)
    lib.es_denoise_loop.argtypes = [P, P, P, C.c_float, P, C.c_int, P]
    ok(lib.es_denoise_loop(ctx, lat, ehs, float(z["guidance_scale"]), P(ts.ctypes.data), len(ts), None), "es_denoise_loop")
    B, h, w, Lc = z["latents"].shape
    img = P()
    assert hip.hipMalloc(C.byref(img), C.c_size_t(B * 3 * 8 * h * 8 * w * 4)) == 0
    lib.es_vae_decode.argtypes = [P, P, P, P]
    ok(lib.es_vae_decode(ctx, lat, img, None), "es_vae_decode")
    assert hip.hipDeviceSynchronize() == 0
    np.savez(out_path, latents=host(lat, (B, h, w, Lc), np.float32), image=host(img, (B, 3, 8 * h, 8 * w), np.float32))
    lib.es_ctx_destroy.argtypes = [P]
    lib.es_ctx_destroy(ctx)
    print("ok")


if __name__ == "__main__":
    main(*sys.argv[1:4])
