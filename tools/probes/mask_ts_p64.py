"""Phase timestamps of the software-pipelined folded mask kernel (mask_logits_p64_kernel, MSM_OPT_MASK_KERNEL = 4); probe build
as tools/probes/mask_ts.py.  Slots: 0 wave start, 1 staging done, 2 + it end of the K loop of tile it, 15 last epilogue done."""
import ctypes, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "unseenobjectswithmeanshift_amd")
EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]
LIB = os.path.join(PKG, "build", "libmsm_ts%s.so" % "".join(a.replace("-D", "_").replace("=", "") for a in EXTRA))
if not os.path.exists(LIB) or "--rebuild" in sys.argv:
    srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.cpp")))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMSM_MASK_TS", *EXTRA, *srcs, "-o", LIB], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
from unseenobjectswithmeanshift_amd import _lib
_lib.LIB_PATH = LIB
from unseenobjectswithmeanshift_amd import ops
KERNEL = int(os.environ.get("MASK_KERNEL", "4"))
_lib.set_option("MASK_KERNEL", KERNEL)
DEV = "cuda"
C = 64
wide = torch.randn(8, 100, 256, device=DEV) * 0.3
e, qb = wide[..., :C], wide[..., 64]
f = torch.randn(8, C, 120, 160, device=DEV)
print(LIB, "kernel", KERNEL)
for tgt in ((15, 20), (30, 40), (60, 80)):
    for _ in range(3):
        ops.mask_logits(e, f, want_mask=False, target_size=tgt, qbias=qb)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
    L = _lib.lib()
    L.msm_debug_mask_ts.argtypes = [ctypes.c_void_p]
    L.msm_debug_mask_ts(buf.ctypes.data_as(ctypes.c_void_p))
    ts = buf.reshape(256, 8, 16).astype(np.int64)[:, :int(os.environ.get("NWAVES", "8"))]
    t0 = ts[:, :, 0].min()
    rel = (ts - t0) * 0.01
    print(f"target={tgt}")
    print(f"  wave start   : min {rel[:,:,0].min():6.2f} mean {rel[:,:,0].mean():6.2f} max {rel[:,:,0].max():6.2f} us")
    print(f"  staging done : mean {rel[:,:,1].mean():6.2f} max {rel[:,:,1].max():6.2f}   (staging itself {np.mean(rel[:,:,1]-rel[:,:,0]):5.2f})")
    prev = rel[:, :, 1]
    for it in range(6):
        cur = rel[:, :, 2 + it]
        valid = ts[:, :, 2 + it] > 0
        if not valid.any():
            break
        n = valid.sum()
        print(f"  tile {it}: waves {int(n):5d}  K loop {np.where(valid, cur - prev, 0).sum() / n:6.2f} us  ends mean {np.where(valid, cur, 0).sum() / n:6.2f} max {np.where(valid, cur, 0).max():6.2f}")
        prev = np.where(valid, cur, prev)
    dclk = (ts[:, :, 14] - ts[:, :, 13]).astype(np.float64)
    dwall = (ts[:, :, 15] - ts[:, :, 1]).astype(np.float64) * 10.0          # ns
    print(f"  shader clock between staging and end: {np.mean(dclk / dwall):5.3f} GHz (s_memtime ticks / 100 MHz wall clock)")
    print(f"  last epilogue: {np.mean(rel[:,:,15] - prev):5.2f} us; wave end mean {rel[:,:,15].mean():6.2f} max {rel[:,:,15].max():6.2f}")
