"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain-Python restatements of the two TIFF byte codecs the product decodes in C
(cerberus_amd/csrc/host_codecs.c, include/cerberus_host.h), after the TIFF 6.0 specification, sections 9 (PackBits), 13 (LZW) and 14 (predictor).
The reference has no code for this: it reads slides through tiatoolbox's WSIReader -> OpenSlide / libtiff (infer/wsi.py:521-531), neither of
which is in this image.  Pins: (1) files written by PIL's bundled libtiff (an independent ENCODER: tests/test_reader_host.py reads them through the
product and through PIL); (2) these decoders against the C ones on libtiff's streams and on the encoders below (random, smooth, constant and
table-overflow inputs).  The decoders were the product's own path until round 6 (1.3 Mpx/s with the interpreter lock held)."""


def lzw_decode(data, expected):
    """TIFF 6.0 section 13 LZW: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the code width grows one code EARLY
    (when the table reaches 511 / 1023 / 2047 entries).  Pure Python (a 256 x 256 RGB tile takes ~50 ms)."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    nbits, bitbuf, bitcnt, prev = 9, 0, 0, None
    for byte in data:
        bitbuf = (bitbuf << 8) | byte
        bitcnt += 8
        while bitcnt >= nbits:
            code = (bitbuf >> (bitcnt - nbits)) & ((1 << nbits) - 1)
            bitcnt -= nbits
            if code == 256:
                table = table[:258]
                nbits, prev = 9, None
                continue
            if code == 257:
                return bytes(out[:expected])
            if prev is None:
                entry = table[code]
            else:
                if code < len(table):
                    entry = table[code]
                elif code == len(table):
                    entry = prev + prev[:1]
                else:
                    raise ValueError("corrupt LZW stream in a TIFF strip / tile")
                table.append(prev + entry[:1])
            out += entry
            prev = entry
            n = len(table)
            nbits = 12 if n >= 2047 else 11 if n >= 1023 else 10 if n >= 511 else 9
            if len(out) >= expected:
                return bytes(out[:expected])
    return bytes(out[:expected])


def packbits_decode(data, expected):
    """TIFF 6.0 section 9 PackBits."""
    out = bytearray()
    i, n = 0, len(data)
    while i < n and len(out) < expected:
        h = data[i]
        i += 1
        if h < 128:
            out += data[i:i + h + 1]
            i += h + 1
        elif h > 128:
            out += data[i:i + 1] * (257 - h)
            i += 1
    return bytes(out[:expected])


# The ENCODERS the tests feed the decoders with are the product's own writer utilities (oracle -> product direction: cerberus_amd/reader.py writes LZW /
# PackBits tiles with them, `write_tiled_tiff(compress="lzw")`); what pins THEM is libtiff reading the files (tests/test_reader_host.py).
from cerberus_amd.reader import tiff_lzw_encode as lzw_encode  # noqa: E402,F401
from cerberus_amd.reader import tiff_packbits_encode as packbits_encode  # noqa: E402,F401
