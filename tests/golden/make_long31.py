"""A zstd frame whose one match reaches back 2^31 - 2 bytes -- offset code 31, the largest the format's 2^31 window allows
(ennaf --long 31: ennaf/src/ennaf.c:247-273; unnaf raises its decoder's limit to match, unnaf/src/input.c:271).  libzstd's own
match finders do not produce such an offset on demand, so the frame is assembled by hand from RFC 8878: a Raw block of 64 KiB of
noise, RLE blocks of zeros up to the right distance, and a compressed block of ten raw literals and ONE sequence whose three code
tables are in RLE mode (no FSE state bits: the bit-stream is just the extra bits).  Checked in this container against the real
libzstd (`zstd -d --long=31`): run this file to repeat the check; tests/test_gpu_decode.py pins the digests below."""
import hashlib
import struct
import sys

import numpy as np

FRAME_SHA256 = "62ac176409bcd274a119487665fb596586e3daaaae7172fc559cbe018ee2577a"
TEXT_SHA256 = "4639e302e84f0e7211d844d5998b2f1989c34322b9b1ffd1d1e3ab0635be6689"
SRC = 1000                        # where in the noise the match starts
ML = 50000
OFF = (1 << 31) - 2


def parts():
    a = np.random.default_rng(31).integers(0, 256, 65536, dtype=np.uint8).tobytes()
    lits = b"0123456789"
    zeros = OFF + SRC - 65536 - len(lits)           # so that (position of the match) - OFF == SRC
    return a, zeros, lits


def frame():
    a, zeros, lits = parts()
    out = bytearray(b"\x28\xb5\x2f\xfd")
    out += bytes([0x00, (31 - 10) << 3])             # no content size, no checksum; Window_Descriptor: 2^31
    def bh(last, typ, size):
        return struct.pack("<I", (size << 3) | (typ << 1) | last)[:3]
    out += bh(0, 0, len(a)) + a
    left = zeros
    while left:
        n = min(left, 128 * 1024)
        out += bh(0, 1, n) + b"\x00"
        left -= n
    # compressed block: raw literals, one sequence (ll = 10, ml = 50000, offset value = OFF + 3)
    ov = OFF + 3
    ofc = ov.bit_length() - 1
    assert ofc == 31
    of_extra = ov - (1 << ofc)
    ml_code, ml_base, ml_bits = 51, 32771, 15
    ml_extra = ML - ml_base
    assert 0 <= ml_extra < (1 << ml_bits)
    bits = ml_extra | (of_extra << ml_bits) | (1 << (ml_bits + ofc))       # read backwards: offset bits first, then match length, then (none) literals length
    nbytes = (ml_bits + ofc + 1 + 7) // 8
    seq = bytes([1, (1 << 6) | (1 << 4) | (1 << 2), len(lits), ofc, ml_code]) + bits.to_bytes(nbytes, "little")
    body = bytes([(len(lits) << 3) | 0]) + lits + seq
    out += bh(1, 2, len(body)) + body
    return bytes(out)


def text_digest():
    a, zeros, lits = parts()
    h = hashlib.sha256()
    h.update(a)
    z = bytes(1 << 24)
    left = zeros
    while left:
        n = min(left, len(z)); h.update(z[:n]); left -= n
    h.update(lits); h.update(a[SRC:SRC + ML])
    return h.hexdigest(), len(a) + zeros + len(lits) + ML


if __name__ == "__main__":
    import subprocess
    f = frame()
    open("/dev/shm/long31_hand.zst", "wb").write(f)
    p = subprocess.run(["/opt/conda/bin/zstd", "-d", "--long=31", "-c", "/dev/shm/long31_hand.zst"], stdout=subprocess.PIPE, check=True)
    want, n = text_digest()
    got = hashlib.sha256(p.stdout).hexdigest()
    print("frame bytes", len(f), "sha256", hashlib.sha256(f).hexdigest())
    print("text bytes", len(p.stdout), n, "libzstd output sha256", got, "expected", want, "OK" if got == want and len(p.stdout) == n else "MISMATCH")
    sys.exit(0 if got == want else 1)
