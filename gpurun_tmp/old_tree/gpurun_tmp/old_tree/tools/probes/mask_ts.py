"""Phase timestamps of the fp32 mask step (probe build: hipcc -DMSM_MASK_TS of the library, see the command below).
    python tools/probes/mask_ts.py        # builds unseenobjectswithmeanshift_amd/build/libmsm_ts.so, runs one launch, prints phases
"""
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
DEV = "cuda"
C = 64
wide = torch.randn(8, 100, 256, device=DEV) * 0.3
e, qb = wide[..., :C], wide[..., 64]
f = torch.randn(8, C, 120, 160, device=DEV)
print(LIB)
for tgt, wm in (((30, 40), False), ((60, 80), False), (None, True)):
    for _ in range(3):
        ops.mask_logits(e, f, want_mask=wm, target_size=tgt, qbias=qb)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
    L = _lib.lib()
    L.msm_debug_mask_ts.argtypes = [ctypes.c_void_p]
    L.msm_debug_mask_ts(buf.ctypes.data_as(ctypes.c_void_p))
    ts = buf.reshape(256, 8, 16).astype(np.int64)
    t0 = ts[:, :, 0].min()
    rel = (ts - t0) * 0.01          # us (100 MHz)
    print(f"target={tgt} write={wm}")
    print(f"  wave start      : min {rel[:,:,0].min():6.2f} mean {rel[:,:,0].mean():6.2f} max {rel[:,:,0].max():6.2f} us")
    print(f"  staging done    : mean {rel[:,:,1].mean():6.2f} max {rel[:,:,1].max():6.2f}")
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(f"  {grp}: staging done {rel[:, sl, 1].mean():5.2f}")
        for it in range(4):
            a, d, b_ = rel[:, sl, 2 + 3 * it], rel[:, sl, 3 + 3 * it], rel[:, sl, 4 + 3 * it]
            valid = ts[:, sl, 2 + 3 * it] > 0
            if not valid.any():
                break
            prev = rel[:, sl, 1] if it == 0 else rel[:, sl, 1 + 3 * it]
            n = valid.sum()
            print(f"    tile {it}: waves {int(n):5d}  K-loop {np.where(valid, a - prev, 0).sum() / n:6.2f} us (ends {np.where(valid, a, 0).sum() / n:6.2f})  MFMA done +{np.where(valid, d - a, 0).sum() / n:5.2f}  "
                  f"epilogue {np.where(valid, b_ - d, 0).sum() / n:6.2f} us   ends at mean {np.where(valid, b_, 0).sum() / n:6.2f} max {np.where(valid, b_, 0).max():6.2f}")
