#!/usr/bin/env python3
"""Prototype of a DENSE layer of a wide state (t = 9) on the matrix cores: generator of the tables and checker of the result
of tools/mfma_dense_proto.hip.  Not part of the product.

  out_i = sum_j c_ij * Z_j  (mod p)   for 64 states per wave, Z_j nine 29-bit limbs each (the internal form of pmx_field.hpp)

as an int8 GEMM: the state's elements are cut into bytes u (K = 9 elements x 36 bytes, 33 used), the table holds, for every
(element j, byte b) and output i, the residue  Y = c_ij * 2^(8 b + 58) mod p  in 32 BALANCED signed bytes; the 32 sums of an
output are recombined, carried and taken through two Montgomery steps (division by 2^58).  Bytes go in as u - 128 (one
v_xor per register); the correction 128 * sum_k Y_k is a constant per output, added with the carries.

  gen   <dir> <log2 states>   writes <dir>/proto_in.bin
  check <dir>                 reads <dir>/proto_in.bin and <dir>/proto_out.bin, verifies the first 512 states
"""
import os
import struct
import sys

import numpy as np

P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001   # BN254 Fr
T = 9
KN, KW = 9, 29
MASK = (1 << KW) - 1
BYTES_PER_ELEM = 36          # 33 used
K = T * BYTES_PER_ELEM       # 324
CHUNK = 32                   # v_mfma_i32_32x32x32_i8
NQ = (K + CHUNK - 1) // CHUNK   # 11
WORDS = NQ * CHUNK // 4      # 88


def balanced_bytes(y, n=32):
    d = []
    for _ in range(n):
        b = y & 0xff
        y >>= 8
        if b >= 128:
            b -= 256
            y += 1
        d.append(b)
    assert y == 0, "top byte of the modulus too large for 32 balanced digits"
    return d


def gen(d, log2n):
    rng = np.random.default_rng(0x5EED)
    n = 1 << log2n
    import random
    rnd = random.Random(1234)
    c = [[rnd.randrange(P) for _ in range(T)] for _ in range(T)]
    # table in A-operand order: [i][q][lane][16 bytes]; lane l: row e = l & 31, k = 32 q + 16 (l >> 5) + byte
    a = np.zeros((T, NQ, 64, 16), dtype=np.int8)
    corr = np.zeros((T, 8), dtype=np.int64)
    for i in range(T):
        colsum = [0] * 32
        for j in range(T):
            for b in range(33):
                y = c[i][j] * pow(2, 8 * b + 58, P) % P
                dig = balanced_bytes(y)
                k = j * BYTES_PER_ELEM + b
                q, r = divmod(k, CHUNK)
                h, byte = divmod(r, 16)
                for e in range(32):
                    a[i, q, 32 * h + e, byte] = dig[e]
                    colsum[e] += dig[e]
        # correction: + 128 * sum_k Y_k, as signed sums per 32-bit word (4 digits each)
        for w in range(8):
            corr[i, w] = sum(128 * colsum[4 * w + t] << (8 * t) for t in range(4))
    p29 = [(P >> (KW * k)) & MASK for k in range(KN)]
    pinv = (-pow(P, -1, 1 << KW)) % (1 << KW)
    # states: limbs < 2^29, top limb < 2^22 (value < 2^254 < p)
    st = rng.integers(0, 1 << KW, size=(n, T, KN), dtype=np.uint32)
    st[:, :, KN - 1] &= (1 << 22) - 1
    with open(os.path.join(d, "proto_in.bin"), "wb") as f:
        f.write(struct.pack("<8I", 0x50524F54, n, T, NQ, 0, 0, 0, 0))
        f.write(np.array(p29 + [pinv], dtype=np.uint32).tobytes())
        f.write(a.tobytes())
        f.write(corr.tobytes())
        f.write(st.tobytes())
    with open(os.path.join(d, "proto_consts.txt"), "w") as f:
        for row in c:
            f.write(" ".join(hex(x) for x in row) + "\n")
    print("wrote", os.path.join(d, "proto_in.bin"), "states", n, "table bytes", a.nbytes)


def check(d):
    raw = open(os.path.join(d, "proto_in.bin"), "rb").read()
    magic, n, t, nq = struct.unpack_from("<4I", raw, 0)
    assert magic == 0x50524F54 and t == T and nq == NQ
    off = 32 + 40 + T * NQ * 64 * 16 + T * 8 * 8
    st = np.frombuffer(raw, dtype=np.uint32, count=n * T * KN, offset=off).reshape(n, T, KN)
    out = np.fromfile(os.path.join(d, "proto_out.bin"), dtype=np.uint32).reshape(n, T, KN)
    c = [[int(x, 16) for x in line.split()] for line in open(os.path.join(d, "proto_consts.txt"))]
    bad = 0
    worst = 0
    for s in list(range(256)) + list(range(n - 256, n)):
        z = [sum(int(st[s, j, k]) << (KW * k) for k in range(KN)) for j in range(T)]
        for i in range(T):
            o = sum(int(out[s, i, k]) << (KW * k) for k in range(KN))
            want = sum(c[i][j] * z[j] for j in range(T)) % P
            limbs_ok = all(int(out[s, i, k]) <= MASK for k in range(KN))
            worst = max(worst, o // P)
            if o % P != want or not limbs_ok:
                bad += 1
                if bad <= 5:
                    print("MISMATCH state", s, "row", i, hex(o % P), hex(want), "limbs ok", limbs_ok)
    print("checked 512 states x 9 rows: %d mismatches; largest result / p = %d" % (bad, worst))
    return bad == 0


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(sys.argv[2], int(sys.argv[3]))
    else:
        sys.exit(0 if check(sys.argv[2]) else 1)
