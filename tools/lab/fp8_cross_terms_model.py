"""CPU model (r5): what would the split-bf16 products lose if their two CROSS terms ran on the fp8 matrix pipe (half the cycles of a
bf16 MFMA on gfx950) -- hi*hi in bf16 as now, lo*hi and hi*lo with both operands quantised to fp8 (e4m3 with a power-of-two scale per
block of 32 along the contraction, as v_mfma_scale_f32_32x32x64_f8f6f4 takes them)?  Exact accumulation (float64) in every variant,
so that only the operand formats differ.  Reports the error of A = V H^T, of W H, and of the residual objective against float64.

    python tools/lab/fp8_cross_terms_model.py [m n k iterations]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import nmf_ref as R  # noqa: E402


def bf16(x):
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32).astype(np.float64)


def fp8_e4m3_blocks(x, axis, block=32):
    """x quantised to e4m3 (4 significant bits, RNE) with a power-of-two scale per block of `block` entries along `axis` chosen so
    that the block maximum lands below 448; values below the scaled subnormal step 2^-9 round to multiples of it."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    shp = x.shape
    pad = (-shp[-1]) % block
    if pad:
        x = np.concatenate([x, np.zeros(shp[:-1] + (pad,))], axis=-1)
    xb = x.reshape(x.shape[:-1] + (-1, block))
    mx = np.abs(xb).max(axis=-1, keepdims=True)
    scale = np.where(mx > 0, 2.0 ** np.ceil(np.log2(np.maximum(mx, 1e-300) / 448.0)), 1.0)
    y = xb / scale
    e = np.floor(np.log2(np.maximum(np.abs(y), 2.0 ** -9)))
    e = np.maximum(e, -6.0)                                    # subnormals share the exponent of the smallest normal
    step = 2.0 ** (e - 3)                                      # 3 explicit mantissa bits
    q = np.round(y / step) * step
    out = (q * scale).reshape(x.shape)[..., :shp[-1]]
    return np.moveaxis(out, -1, axis)


def split(x):
    hi = bf16(x)
    return hi, bf16(np.asarray(x, dtype=np.float64) - hi)


def product(xh, xl, yh, yl, scheme, kaxis_x, kaxis_y):
    """x y with the contraction along kaxis_x of x and kaxis_y of y (x: [.., K] or [K, ..])"""
    def mm(a, b):
        a2 = a if kaxis_x == 1 else a.T
        b2 = b if kaxis_y == 0 else b.T
        return a2 @ b2
    main = mm(xh, yh)
    if scheme == "bf16x3":
        return main + mm(xl, yh) + mm(xh, yl)
    if scheme == "fp8cross":
        q = lambda a, ax: fp8_e4m3_blocks(a, ax)               # noqa: E731
        return main + mm(q(xl, kaxis_x), q(yh, kaxis_y)) + mm(q(xh, kaxis_x), q(yl, kaxis_y))
    if scheme == "bf16x2":
        return main + mm(xh, yl)
    raise ValueError(scheme)


def main():
    m, n, k, iters = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 4096, 64, 300)
    v = R.planted_matrix(m, n, k, seed=0, dtype=np.float32).astype(np.float64)
    rs = np.random.RandomState(0)
    w, h = np.abs(rs.randn(m, k)), np.abs(rs.randn(k, n))
    for it in range(iters):                                    # float64 MUR to a small residual, like a long device run
        wh = w @ h
        w = R.mur_w_step("eu", v, w, h, wh, 0.0)
        h = R.mur_h_step("eu", v, w, h, w @ h, 0.0)
    w, h = w.astype(np.float32).astype(np.float64), h.astype(np.float32).astype(np.float64)
    wh = w @ h
    obj = 0.5 * np.sum((v - wh) ** 2)
    a_ref = v @ h.T
    vh, vl = split(v)
    hh, hl = split(h)
    wh_, wl_ = split(w)
    print(f"{m}x{n} k={k} after {iters} f64 iterations: residual / ||V|| = {np.sqrt(2 * obj) / np.linalg.norm(v):.3e}, objective {obj:.6g}")
    for scheme in ("bf16x3", "fp8cross", "bf16x2"):
        a = product(vh, vl, hh, hl, scheme, 1, 1)              # V H^T: contraction over n (axis 1 of both)
        p = product(wh_, wl_, hh, hl, scheme, 1, 0)            # W H: contraction over k
        o = 0.5 * np.sum((v - p) ** 2)
        print(f"  {scheme:9s} A = V H^T: max rel {np.max(np.abs(a - a_ref) / np.abs(a_ref)):.2e}, rms rel {np.sqrt(np.mean(((a - a_ref) / a_ref) ** 2)):.2e};"
              f"  W H: rms abs {np.sqrt(np.mean((p - wh) ** 2)):.2e};  objective rel diff {(o - obj) / obj:+.2e}")


if __name__ == "__main__":
    main()
