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


def lzw_encode(data):
    """TIFF 6.0 section 13 encoder (MSB-first, early change, ClearCode first, a ClearCode when the table holds 4094 entries, EndOfInformation last):
    what the tests feed the decoders beside libtiff's own streams -- tiles of a TILED file, which PIL cannot write."""
    out = bytearray()
    bitbuf, bitcnt = 0, 0
    nbits = 9

    def put(code):
        nonlocal bitbuf, bitcnt
        bitbuf = (bitbuf << nbits) | code
        bitcnt += nbits
        while bitcnt >= 8:
            out.append((bitbuf >> (bitcnt - 8)) & 0xFF)
            bitcnt -= 8
        bitbuf &= (1 << bitcnt) - 1

    table = {bytes((i,)): i for i in range(256)}
    nxt = 258
    put(256)
    w = b""
    for byte in bytes(data):
        wc = w + bytes((byte,))
        if wc in table:
            w = wc
            continue
        put(table[w])
        table[wc] = nxt
        nxt += 1
        # the DECODER's table is one entry behind the encoder's, and it widens its codes when ITS table holds 511 / 1023 / 2047 entries
        # (one code early): seen from here that is nxt = 512 / 1024 / 2048
        if nxt == 4094:
            put(256)
            table = {bytes((i,)): i for i in range(256)}
            nxt, nbits = 258, 9
        else:
            nbits = 12 if nxt >= 2048 else 11 if nxt >= 1024 else 10 if nxt >= 512 else 9
        w = bytes((byte,))
    if w:
        put(table[w])
        nxt += 1  # (the decoder adds an entry for this code too, and widens on it)
        if nxt != 4094:
            nbits = 12 if nxt >= 2048 else 11 if nxt >= 1024 else 10 if nxt >= 512 else 9
    put(257)
    if bitcnt:
        out.append((bitbuf << (8 - bitcnt)) & 0xFF)
    return bytes(out)


def packbits_encode(data):
    """TIFF 6.0 section 9 encoder: runs of 3 and more as replicate packets, everything else as literal packets of up to 128 bytes."""
    data = bytes(data)
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 127:
            j += 1
        if j - i >= 2:
            out += bytes((257 - (j - i + 1), data[i]))
            i = j + 1
            continue
        k = i
        while k < n and k - i < 128 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
            k += 1
        out += bytes((k - i - 1,)) + data[i:k]
        i = k
    return bytes(out)
