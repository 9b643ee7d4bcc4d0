"""LZ4 frame decoder in plain Python — rosbag's `compression=lz4` chunks (roslz4 writes the LZ4 frame format: magic
0x184D2204, FLG/BD descriptor, size-prefixed blocks, end mark).  The image has no lz4 module, and a bag is decoded once
before the hot path, so a byte-loop decoder is enough.  Checksums (xxh32) are verified when `xxhash` is importable.
"""
import struct

_MAGIC = 0x184D2204


def decode_block(src, out: bytearray) -> None:
    """One LZ4 block appended to `out`; matches may reach back into what `out` already holds (linked blocks)."""
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        if i + lit > n:
            raise ValueError("lz4: literal run past the end of the block")
        out += src[i:i + lit]; i += lit
        if i >= n:
            break  # the last sequence is literals only
        if i + 2 > n:
            raise ValueError("lz4: truncated match offset")
        off = src[i] | (src[i + 1] << 8); i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4: match offset outside the window")
        mlen = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                b = src[i]; i += 1
                mlen += b
                if b != 255:
                    break
        start = len(out) - off
        if off >= mlen:
            out += out[start:start + mlen]
        else:  # overlapping copy = the last `off` bytes repeated
            pat = bytes(out[start:])
            out += (pat * (mlen // off + 1))[:mlen]


def decompress(data, expected_size=None) -> bytes:
    data = bytes(data)
    if len(data) < 7 or struct.unpack_from("<I", data, 0)[0] != _MAGIC:
        raise ValueError("lz4: not an LZ4 frame")
    flg, pos = data[4], 6
    if (flg >> 6) != 1:
        raise ValueError("lz4: unsupported frame version")
    block_checksum, has_size, content_checksum, has_dict = bool(flg & 0x10), bool(flg & 0x08), bool(flg & 0x04), bool(flg & 0x01)
    content_size = None
    if has_size:
        (content_size,) = struct.unpack_from("<Q", data, pos); pos += 8
    if has_dict:
        raise ValueError("lz4: frames with a dictionary are not supported")
    try:
        import xxhash
    except ImportError:  # checksums are then skipped
        xxhash = None
    if xxhash is not None and ((xxhash.xxh32(data[4:pos], seed=0).intdigest() >> 8) & 0xFF) != data[pos]:
        raise ValueError("lz4: frame header checksum mismatch")
    pos += 1
    out = bytearray()
    while True:
        if pos + 4 > len(data):
            raise ValueError("lz4: truncated frame (no end mark)")
        (bs,) = struct.unpack_from("<I", data, pos); pos += 4
        if bs == 0:
            break
        raw, bs = bool(bs & 0x80000000), bs & 0x7FFFFFFF
        if pos + bs > len(data):
            raise ValueError("lz4: truncated block")
        block = data[pos:pos + bs]; pos += bs
        if block_checksum:
            if xxhash is not None and xxhash.xxh32(block, seed=0).intdigest() != struct.unpack_from("<I", data, pos)[0]:
                raise ValueError("lz4: block checksum mismatch")
            pos += 4
        if raw:
            out += block
        else:
            try:
                decode_block(block, out)
            except IndexError:
                raise ValueError("lz4: block ends inside a sequence") from None
    if content_checksum:
        if pos + 4 > len(data):
            raise ValueError("lz4: truncated content checksum")
        if xxhash is not None and xxhash.xxh32(bytes(out), seed=0).intdigest() != struct.unpack_from("<I", data, pos)[0]:
            raise ValueError("lz4: content checksum mismatch")
    for want in (content_size, expected_size):
        if want is not None and want != len(out):
            raise ValueError(f"lz4: decoded {len(out)} bytes, header says {want}")
    return bytes(out)
