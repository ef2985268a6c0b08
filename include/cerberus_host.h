/* cerberus_host.h -- C ABI of libcerberus_host.so: host-side byte codecs of the slide reader (plain C, no HIP, no device memory).
 *
 * SURVEY.md par.8(f)2, real-slide ingest.  The reference reads slides through tiatoolbox's WSIReader -- OpenSlide / libtiff under twelve DataLoader
 * worker processes (infer/wsi.py:521-531, 936-950); this build parses the TIFF container itself (cerberus_amd/reader.py) and decodes tiles on a pool
 * ahead of the GPU.  JPEG and deflate tiles go to libjpeg / zlib; the two TIFF compressions without a library behind Python live here:
 * LZW (compression 5) and PackBits (32773), plus the horizontal predictor that LZW / deflate files usually carry.
 *
 * Conventions: plain pointers + sizes, re-entrant (no global state), never more than dst_cap bytes written; the decoders return the number of bytes
 * written (a stream may end early: the caller compares with what the tile needs) or a negative code for a corrupt / unsupported stream.
 * Called through ctypes (cerberus_amd/_hostlib.py) with the interpreter lock released, one tile or strip per call.
 */
#ifndef CERBERUS_HOST_H
#define CERBERUS_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int cerb_host_version(void);

/* TIFF 6.0 section 13 LZW of one strip / tile (MSB-first 9..12-bit codes, early change).
 * -> bytes written; -1 corrupt stream (a code beyond the table); -2 pre-6.0 "old-style" LSB-first stream (not supported).
 * Replaces the libtiff decode behind tiatoolbox's TIFFWSIReader / OpenSlide generic-tiff reads (infer/wsi.py:521-531). */
int64_t cerb_host_lzw_decode(const uint8_t* src, int64_t n_src, uint8_t* dst, int64_t dst_cap);

/* TIFF 6.0 section 9 PackBits of one strip / tile.  -> bytes written; -1 when a run or literal reads past the input. */
int64_t cerb_host_packbits_decode(const uint8_t* src, int64_t n_src, uint8_t* dst, int64_t dst_cap);

/* TIFF 6.0 section 14 Predictor = 2 on 8-bit chunky samples, in place: rows x cols pixels of `samples` bytes, every row the running sum (mod 256)
 * of its stored differences, per sample. */
void cerb_host_unpredict_u8(uint8_t* px, int64_t rows, int64_t cols, int samples);

/* A whole window of one TIFF level in ONE call: tile / strip i (file bytes [offsets[i], offsets[i] + counts[i]), read with pread on `fd`; decoded
 * geometry rows[i] x tile_cols pixels of `samples` 8-bit chunky samples, first pixel at (gx0[i], gy0[i]) of the level) is read, decoded (codec = the
 * TIFF Compression tag: 1 raw, 8 / 32946 deflate through zlib, 5 LZW, 32773 PackBits), un-predicted (predictor 2) and its part inside the window
 * [x0, x1) x [y0, y1) written as RGB bytes to out + (y - y0) * out_stride + (x - x0) * 3 -- on n_threads pthreads (at most 64) that take tiles off a
 * shared counter; tiles do not overlap, so every thread writes its own pixels.  JPEG tiles (7) stay with libjpeg behind the reader.
 * This is the reader's equivalent of the reference's DataLoader workers around tiatoolbox's read_bounds (infer/wsi.py:936-950).
 * -> 0, or the first failure with *bad_tile = its index: -1 corrupt LZW / PackBits stream, -2 old-style LZW, -3 short file read, -4 fewer decoded
 * bytes than the tile's pixels, -5 zlib error, -6 out of memory, -7 unsupported codec, -8 bad arguments. */
int cerb_host_tiff_read_tiles(int fd, int codec, int predictor, int samples, int tile_cols, int n_tiles, const int64_t* offsets, const int64_t* counts,
                              const int32_t* rows, const int32_t* gx0, const int32_t* gy0, int x0, int y0, int x1, int y1, uint8_t* out,
                              int64_t out_stride, int n_threads, int32_t* bad_tile);

#ifdef __cplusplus
}
#endif
#endif
