"""numerical side of the split-bf16 option: a 15x15 x 16-channel dot product (K = 3600, the reduction
of the 16->128 layer) and a K = 28800 one (128 channels) computed with fp32 operands, with bf16 operands, and with the
two- and three-product splits x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), all accumulated in fp32 like the MFMA does;
errors against float64, relative to the rms of the exact results.  Inputs like the model's: activations log(1 + 10 Gamma),
weights N(0, 2 / fan_in)."""
import numpy as np
import torch


def bf16(x):
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


rng = np.random.default_rng(0)
for K in (3600, 28800):
    n = 4096
    x = np.log1p(10 * rng.gamma(0.3, 0.05, size=(n, K))).astype(np.float32)
    w = (rng.standard_normal((K, 16)) * np.sqrt(2.0 / K)).astype(np.float32)
    exact = x.astype(np.float64) @ w.astype(np.float64)
    rms = np.sqrt((exact ** 2).mean())
    xh, wh = bf16(x), bf16(w)
    xl, wl = bf16(x - xh), bf16(w - wh)
    f32 = lambda a, b: (torch.from_numpy(a) @ torch.from_numpy(b)).numpy()      # fp32 accumulate
    res = {"fp32 operands": f32(x, w),
           "bf16 operands (1 product)": f32(xh, wh),
           "split, 3 products (hh + hl + lh)": f32(xh, wh) + f32(xh, wl) + f32(xl, wh),
           "split, 4 products (+ ll)": f32(xh, wh) + f32(xh, wl) + f32(xl, wh) + f32(xl, wl)}
    print(f"K = {K}: rms of the exact result {rms:.3f}")
    for name, r in res.items():
        e = r.astype(np.float64) - exact
        print(f"   {name:34s} max |err| / rms = {np.abs(e).max() / rms:.2e}   rms err / rms = {np.sqrt((e ** 2).mean()) / rms:.2e}")
