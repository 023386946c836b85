#!/usr/bin/env python
"""CPU numerics study for the next-round conv path (DESIGN.md §5 "next"): fp32 operands split exactly into three
bf16 terms, cross products on the bf16 matrix pipe, fp32 accumulation.  Question: how many of the 9 cross products
are needed to stay at the fp32 kernel's error level?  (No GPU; emulates operand rounding exactly, accumulation order
approximately: sequential fp32 adds over k like one MFMA lane's chain.)

    python tools/bf16x3_study.py > profiles/bf16x3_study_r1.txt
"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest-even bf16, returned as fp32."""
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split3(x):
    a1 = bf16_round(x)
    a2 = bf16_round(x - a1)
    a3 = bf16_round(x - a1 - a2)
    return a1, a2, a3


def chain_fp32(terms):
    """sum over axis 0 with sequential fp32 adds."""
    acc = np.zeros(terms.shape[1:], np.float32)
    for t in terms:
        acc = (acc + t).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    M, N = 64, 48
    print("K      fp32-chain   fp32-blocked(18x16)   bf16x3: 3 terms   6 terms   9 terms   (RMS error / RMS of the exact result)")
    for K in (288, 864, 1728, 5184):
        A = rng.normal(0, 1, (M, K)).astype(np.float32) * (rng.random((M, K)) < 0.7)      # SiLU-like: many small values
        A = np.where(A < 0, A * 0.1, A).astype(np.float32)
        B = rng.normal(0, (2.0 / K) ** 0.5, (N, K)).astype(np.float32)
        exact = A.astype(np.float64) @ B.astype(np.float64).T
        scale = np.sqrt((exact ** 2).mean())
        prod = (A[:, None, :] * B[None, :, :])                                            # fp32 products (the MFMA fuses them: close enough)
        p64 = A.astype(np.float64)[:, None, :] * B.astype(np.float64)[None, :, :]
        # fp32 FMA chain: emulate exact product + one rounding per add
        acc = np.zeros((M, N), np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + p64[:, :, k]).astype(np.float32)
        e_chain = np.sqrt(((acc - exact) ** 2).mean()) / scale
        # blocked: 288-long partial sums added into the accumulator (the engine's two-level scheme)
        accb = np.zeros((M, N), np.float32)
        for k0 in range(0, K, 288):
            part = np.zeros((M, N), np.float32)
            for k in range(k0, min(k0 + 288, K)):
                part = (part.astype(np.float64) + p64[:, :, k]).astype(np.float32)
            accb = (accb + part).astype(np.float32)
        e_blk = np.sqrt(((accb - exact) ** 2).mean()) / scale
        a = split3(A)
        b = split3(B)
        out = []
        for pairs in ([(0, 0), (0, 1), (1, 0)],
                      [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)],
                      [(i, j) for i in range(3) for j in range(3)]):
            accs = np.zeros((M, N), np.float32)
            # small terms first inside each k (what a kernel would do by issuing the low-order MFMAs first)
            for k0 in range(0, K, 288):
                part = np.zeros((M, N), np.float32)
                for (i, j) in sorted(pairs, key=lambda t: -(t[0] + t[1])):
                    t64 = a[i].astype(np.float64)[:, None, k0:k0 + 288] * b[j].astype(np.float64)[None, :, k0:k0 + 288]
                    for k in range(t64.shape[2]):
                        part = (part.astype(np.float64) + t64[:, :, k]).astype(np.float32)
                accs = (accs + part).astype(np.float32)
            out.append(np.sqrt(((accs - exact) ** 2).mean()) / scale)
        print(f"{K:5d}  {e_chain:.3e}    {e_blk:.3e}             {out[0]:.3e}        {out[1]:.3e} {out[2]:.3e}")
    print("\nreading: 6 cross products reproduce the fp32 error level (the three dropped ones are <= 2^-24 relative);"
          " 3 products (bf16x2-like) are ~2^-16 relative: not admissible under the 1e-3 px parity bar.")


if __name__ == "__main__":
    main()
