"""Input projections at B = 8, 640x480: time per level (single-level launches) and of the one multi-level launch, fp32 and lp forms,
on inputs rotated through a pool larger than the Infinity Cache (cold, as in a pass); plus a plain device copy of the same bytes."""
import os
import sys
import torch

if os.environ.get("MSM_TREE"):
    sys.path.insert(0, os.path.abspath(os.environ["MSM_TREE"]))
sys.path.append(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B = 8
levels = [(2048, 15, 20), (1024, 30, 40), (512, 60, 80)]
POOL = 6          # 6 x 137 MB > 256 MB
g = torch.Generator(device="cpu").manual_seed(1)
xs = [[torch.randn(B, c, h, w, generator=g).to(dev) for (c, h, w) in levels] for _ in range(POOL)]
ws = [torch.randn(64, c, generator=g).to(dev) * c ** -0.5 for (c, h, w) in levels]
bs = [torch.randn(64, generator=g).to(dev) for _ in levels]
wp = [ops.pack_conv_in_weight(w) for w in ws]
wl = [ops.pack_conv_in_weight_lp(w) for w in ws]
S = sum(h * w for _, h, w in levels)


def timed(fn, n=60):
    """Best of three event-timed runs of n back-to-back calls (a single run now and then carries a one-off stall of tens of
    milliseconds -- 750 us per call where the kernel trace shows 46: rocprofv3's per-dispatch durations are the reference)."""
    best = float("inf")
    for _ in range(3):
        for i in range(6):
            fn(i % POOL)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i % POOL)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / n)
    return best


print(os.path.dirname(ops.__file__))
for lp in (False,):
    w_ = wl if lp else wp
    for l, (c, h, w) in enumerate(levels):
        out = torch.empty(B, h * w, 64, device=dev)
        st = torch.zeros(B, 64, 2, device=dev, dtype=torch.float64)
        t = timed(lambda i: ops.conv1x1_in(xs[i][l], w_[l], bs[l], out=out, stats=st, stats_cleared=True, lp=lp))
        mb = xs[0][l].numel() * 4 / 1e6
        print(f"lp={lp} level Cin={c} {h}x{w}: {t:.1f} us  ({mb:.0f} MB -> {mb / t:.2f} TB/s)")
    out = torch.empty(B, S, 64, device=dev)
    st = torch.zeros(len(levels), B, 64, 2, device=dev, dtype=torch.float64)
    t = timed(lambda i: ops.conv1x1_in_multi(xs[i], w_, bs, out, st, stats_cleared=True, lp=lp))
    print(f"lp={lp} multi: {t:.1f} us")
dst = [torch.empty_like(x) for x in xs[0]]
t = timed(lambda i: [d.copy_(x) for d, x in zip(dst, xs[i])])
print(f"device copies of the three levels (read + write 137 MB each way): {t:.1f} us")
red = lambda i: [x.sum() for x in xs[i]]
print(f"torch sum of the three levels: {timed(red):.1f} us")
if hasattr(ops.lib(), "msm_conv1x1_in_multi_wide"):
    out = torch.empty(B, S, 64, device=dev)
    st = torch.zeros(len(levels), B, 64, 2, device=dev, dtype=torch.float64)
    t = timed(lambda i: ops.conv1x1_in_multi(xs[i], wl, bs, out, st, stats_cleared=True, lp="wide"))
    print(f"wide multi: {t:.1f} us")
    for l, (c, h, w) in enumerate(levels):
        out1 = torch.empty(B, h * w, 64, device=dev)
        st1 = torch.zeros(1, B, 64, 2, device=dev, dtype=torch.float64)
        t = timed(lambda i: ops.conv1x1_in_multi([xs[i][l]], [wl[l]], [bs[l]], out1, st1, stats_cleared=True, lp="wide"))
        print(f"wide level Cin={c} {h}x{w}: {t:.1f} us")
    from unseenobjectswithmeanshift_amd import _lib
    out = torch.empty(B, S, 64, device=dev)
    st = torch.zeros(len(levels), B, 64, 2, device=dev, dtype=torch.float64)
    for rep in range(2):
        for cfg in (421, 422, 442, 441, 444, 222, 221, 211, 411):
            _lib.set_option("CONVIN_NT", cfg)
            t = timed(lambda i: ops.conv1x1_in_multi(xs[i], wl, bs, out, st, stats_cleared=True, lp="wide"), n=100)
            print(f"wide multi, K slices {cfg}: {t:.1f} us")
    # the FPN lateral (res2: 256 channels, 120x160) through the same kernel, against its own lp form
    _lib.set_option("CONVIN_NT", _lib.OPT_AUTO)
    xl = [torch.randn(B, 256, 120, 160, generator=g).to(dev) for _ in range(3)]
    w2 = torch.randn(64, 256, generator=g).to(dev) / 16
    w2l = ops.pack_conv_in_weight_lp(w2)
    o2 = torch.empty(B, 19200, 64, device=dev)
    s2 = torch.zeros(1, B, 64, 2, device=dev, dtype=torch.float64)
    print(f"lateral, lp form: {timed(lambda i: ops.conv1x1_in(xl[i % 3], w2l, None, out=o2, stats=s2[0], stats_cleared=True, lp=True)):.1f} us")
    for cfg in (_lib.OPT_AUTO, 1, 2, 4):
        _lib.set_option("CONVIN_NT", cfg)
        print(f"lateral, wide form, K slices {cfg}: {timed(lambda i: ops.conv1x1_in_multi([xl[i % 3]], [w2l], [None], o2, s2, stats_cleared=True, lp='wide')):.1f} us")
