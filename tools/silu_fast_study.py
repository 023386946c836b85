#!/usr/bin/env python
"""CPU study: accuracy of the epilogue's SiLU (csrc/h2_common.h: h2_exp_neg + h2_div — v_exp_f32 on a compensated
argument, v_rcp_f32 + one Newton step) against the fp32 formula the reference evaluates (correctly rounded exp +
division, what torch's CPU silu amounts to).  The hardware's 1-ulp v_rcp_f32 is emulated pessimistically (always one
ulp low); exp2 is numpy's.    python tools/silu_fast_study.py > profiles/h2_silu_accuracy_r3.txt"""
import numpy as np

f32 = np.float32


def silu_fast(x):
    x = x.astype(f32)
    hi, lo, ln2 = f32(1.4426950216293335), f32(1.9259629911783190e-8), f32(0.6931471805599453)
    nx = -x
    t = (nx * hi).astype(f32)
    tl = ((nx.astype(np.float64) * np.float64(hi)) - t.astype(np.float64)).astype(f32)            # fma(-x, hi, -t)
    tl = (nx.astype(np.float64) * np.float64(lo) + tl.astype(np.float64)).astype(f32)             # fma(-x, lo, tl)
    e0 = np.exp2(np.minimum(t, f32(126.0)).astype(np.float64)).astype(f32)
    e = (e0.astype(np.float64) * (tl * ln2).astype(f32).astype(np.float64) + e0.astype(np.float64)).astype(f32)
    d = (f32(1.0) + e).astype(f32)
    r = np.nextafter((1.0 / d.astype(np.float64)).astype(f32), f32(0))                             # 1 ulp low
    y = (x * r).astype(f32)
    rem = (-(y.astype(np.float64)) * d.astype(np.float64) + x.astype(np.float64)).astype(f32)
    return (rem.astype(np.float64) * r.astype(np.float64) + y.astype(np.float64)).astype(f32)


def main():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 3, 2000000), rng.uniform(-90, 90, 500000),
                        np.array([0.0, -0.0, 1e-30, -1e-30, 88., -88., -87.3, -100., -1e4, 1e4])]).astype(f32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float64) / (1 + np.exp(-x.astype(np.float64)))
        ref = (x / (f32(1) + np.exp((-x).astype(np.float64)).astype(f32))).astype(f32)
    ulp = np.spacing(np.abs(want).astype(f32)).astype(np.float64)
    ok = np.abs(want) > 1e-30
    for name, got in (("epilogue sequence (v_exp_f32 compensated + v_rcp_f32 + Newton)", silu_fast(x)),
                      ("fp32 formula: correctly rounded exp, IEEE division", ref)):
        err = np.abs(got.astype(np.float64) - want) / np.maximum(ulp, 1e-45)
        print(f"{name}: max {err[ok].max():.2f} ulp at x = {x[ok][err[ok].argmax()]:.4f}, mean {err[ok].mean():.3f} ulp over {ok.sum()} arguments")
    with np.errstate(over="ignore"):
        big = silu_fast(np.array([-1e4, -200., 200., 1e4], f32))
    print("extremes (-1e4, -200, 200, 1e4) ->", big)


if __name__ == "__main__":
    main()
