"""Restatement of the reference's input encodings (src/absorb.rs) for checking the host mirrors - TEST INFRASTRUCTURE.

Values are tagged tuples so that this file shares no code with sponge_amd/absorb.py:
    ("u8", 7) ("u16", n) ("u32", n) ("u64", n) ("u128", n) ("usize", n)
    ("i8", n) ... ("i128", n) ("isize", n)            ("bool", b)       ("fp", canonical_int)   (native element)
    ("vec", elem_tag, [python values])                 ("opt", None | tagged value)
    ("with_len", ("vec", ...))
    ("te", x, y)  ("sw", x, y, infinity)               affine points over the sponge's own field (absorb.rs:232-254)
`field_elements(p, bits, v)` -> canonical integers, `sponge_bytes(p_bits, v)` -> bytes.

Unpinned by reference KATs: the reference's tests for this layer are differential / self-consistency only
(src/poseidon/tests.rs:26-117); two encodings depend on third-party ark-ff / ark-serialize behaviour that is not in
the reference tree (byte packing 31 B per element after a u64-LE length, src/absorb.rs:135-139; Fp as
ceil(bits/8) little-endian bytes, :153-155; curve points as ark-ec's ToConstraintField vectors - [x, y] for
twisted Edwards, [x, y, infinity] for short Weierstrass - serialised with a u64-LE element count, :232-254).
"""
UNSIGNED = {"u8": 1, "u16": 2, "u32": 4, "u64": 8, "u128": 16, "usize": 8}
SIGNED = {"i8": 1, "i16": 2, "i32": 4, "i64": 8, "i128": 16, "isize": 8}


def _pack(bits, data):
    step = (bits - 1) // 8
    return [int.from_bytes(data[i:i + step], "little") for i in range(0, len(data), step)]


def field_elements(p, bits, v):
    tag = v[0]
    if tag in UNSIGNED:
        return [v[1] % p]                                    # F::from(x)                   absorb.rs:127,173
    if tag in SIGNED:
        return [(p - (-v[1]) % p) % p if v[1] < 0 else v[1] % p]   # -F::from(|x|)             absorb.rs:190-196
    if tag == "bool":
        return [1 if v[1] else 0]                            # absorb.rs:147-149
    if tag == "fp":
        return [v[1] % p]                                    # native field_cast             absorb.rs:156-158
    if tag == "vec":
        _, et, items = v
        if et == "u8":                                       # absorb.rs:135-139
            data = bytes(items)
            return _pack(bits, len(data).to_bytes(8, "little") + data)
        out = []
        for it in items:                                     # default batch: element by element, no length
            out += field_elements(p, bits, it if isinstance(it, tuple) else (et, it))
        return out
    if tag == "opt":                                         # absorb.rs:298-303
        return [0] if v[1] is None else [1] + field_elements(p, bits, v[1])
    if tag == "with_len":                                    # absorb.rs:96-99
        return [len(v[1][2]) % p] + field_elements(p, bits, v[1])
    if tag == "te":                                          # absorb.rs:240-241
        return [v[1] % p, v[2] % p]
    if tag == "sw":                                          # absorb.rs:252-253
        return [v[1] % p, v[2] % p, 1 if v[3] else 0]
    raise ValueError(tag)


def sponge_bytes(bits, v):
    tag = v[0]
    if tag in UNSIGNED:
        return v[1].to_bytes(UNSIGNED[tag], "little")
    if tag in SIGNED:
        return v[1].to_bytes(SIGNED[tag], "little", signed=True)
    if tag == "bool":
        return bytes([1 if v[1] else 0])
    if tag == "fp":
        return v[1].to_bytes((bits + 7) // 8, "little")
    if tag == "vec":
        _, et, items = v
        return b"".join(sponge_bytes(bits, it if isinstance(it, tuple) else (et, it)) for it in items)
    if tag == "opt":
        return b"\x00" if v[1] is None else b"\x01" + sponge_bytes(bits, v[1])
    if tag == "with_len":
        return len(v[1][2]).to_bytes(8, "little") + sponge_bytes(bits, v[1])
    if tag in ("te", "sw"):                                  # absorb.rs:233-238, 245-250
        coords = [v[1], v[2]] + ([1 if v[3] else 0] if tag == "sw" else [])
        width = (bits + 7) // 8
        return len(coords).to_bytes(8, "little") + b"".join(c.to_bytes(width, "little") for c in coords)
    raise ValueError(tag)


def fork_input(domain: bytes):
    """CryptographicSponge::fork (src/lib.rs:149-157): the Vec<u8> that gets absorbed."""
    return ("vec", "u8", list(len(domain).to_bytes(8, "little") + domain))
