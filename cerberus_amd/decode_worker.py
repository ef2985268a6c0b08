"""A tile-decode worker process of cerberus_amd.reader (torch-free: numpy, PIL, zlib).  Started by reader._WorkerPool as
`python -c "... decode_worker.main()"` with pipes on stdin / stdout; every message is a 4-byte length + a pickle.  A task is the argument tuple of
reader._worker_decode (file, level, tile list, window, shared-memory name, shape): the worker opens the file itself, decodes its tiles and writes the
pixels into the shared block; the reply is ("ok", tile count) or ("err", traceback).  End of input ends the worker."""
import os
import pickle
import struct
import sys
import traceback


def _read_exact(f, n):
    buf = b""
    while len(buf) < n:
        part = f.read(n - len(buf))
        if not part:
            return None
        buf += part
    return buf


def main():
    os.environ["CERB_DECODE_WORKER"] = "1"  # a worker never starts pools of its own
    os.environ["CERB_DECODE_THREADS"] = "1"
    from cerberus_amd import reader

    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr  # nothing but replies on the pipe
    while True:
        head = _read_exact(fin, 4)
        if head is None:
            return
        body = _read_exact(fin, struct.unpack("<I", head)[0])
        if body is None:
            return
        try:
            reply = ("ok", reader._worker_decode(*pickle.loads(body)))
        except BaseException:  # noqa: BLE001
            reply = ("err", traceback.format_exc())
        out = pickle.dumps(reply)
        fout.write(struct.pack("<I", len(out)) + out)
        fout.flush()


if __name__ == "__main__":
    main()
