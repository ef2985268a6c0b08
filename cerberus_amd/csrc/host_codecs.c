/* libcerberus_host.so -- host-side byte codecs of the slide reader (cerberus_amd/reader.py), plain C, no HIP, no Python.
 *
 * The reference opens slides through tiatoolbox's WSIReader (infer/wsi.py:521-531, 936-950: OpenSlide / libtiff under 12 DataLoader workers); this
 * build reads TIFF containers itself and hands tiles to a decode pool ahead of the GPU.  JPEG and deflate tiles go to libjpeg / zlib (both release the
 * interpreter lock); LZW (TIFF compression 5) and PackBits (32773) have no such library behind Python, and a pure-Python LZW decoder runs at
 * ~1.3 Mpx/s WITH the lock held -- a hundredth of what one GPU infers.  These entry points are called through ctypes (lock released), one tile or
 * strip per call -- and, for whole reads, cerb_host_tiff_read_tiles: ONE call per window that reads, decodes (raw / deflate / LZW / PackBits),
 * un-predicts and places every tile on its own pthreads: per-tile calls from Python threads stop scaling at two (a thread coming back from a
 * 0.5 ms decode waits for the interpreter lock up to a 5 ms switch interval).  include/cerberus_host.h declares them.
 *
 * Every function is re-entrant (no global state) and writes at most `dst_cap` bytes.
 */
#define _XOPEN_SOURCE 700
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#define CERB_HOST_VERSION 2

#define LZW_CLEAR 256
#define LZW_EOI 257
#define LZW_FIRST 258
#define LZW_TABLE 4096

int cerb_host_version(void) { return CERB_HOST_VERSION; }

/* TIFF 6.0 section 13: codes of 9..12 bits packed MSB first, ClearCode 256, EndOfInformation 257, the code width grows one code EARLY (when the
 * table holds 511 / 1023 / 2047 entries).  A strip / tile is one independent stream.
 *
 * The table holds no strings and no prefix chains: the string of entry k = string(previous code) + first byte of string(current code), and those
 * bytes sit NEXT TO EACH OTHER in the output already (the current string was written right behind the previous one) -- an entry is (position in
 * dst, length), and emitting a code is one forward copy inside dst (8 bytes at a time where the tail slack allows; byte by byte for the
 * "KwKwK" code that names the entry being defined, whose last byte is the first one this very copy writes).
 *
 * -> bytes written (<= dst_cap; the stream may end early), -1 for a code beyond the table (corrupt stream), -2 for the pre-6.0 "old-style" bit
 * order (LSB first; libtiff's compatibility mode -- not supported, said so instead of decoding noise). */
int64_t cerb_host_lzw_decode(const uint8_t* src, int64_t n_src, uint8_t* dst, int64_t dst_cap) {
    int64_t pos[LZW_TABLE];
    int32_t length[LZW_TABLE];
    if (n_src >= 2 && src[0] == 0 && (src[1] & 1)) return -2;
    int next = LZW_FIRST, nbits = 9, prev = -1;
    int64_t prev_pos = 0;
    int32_t prev_len = 0;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    int64_t ip = 0, op = 0;
    while (op < dst_cap) {
        while (bitcnt < nbits) {
            if (ip >= n_src) return op;
            bitbuf = (bitbuf << 8) | src[ip++];
            bitcnt += 8;
        }
        const int code = (int)((bitbuf >> (bitcnt - nbits)) & ((1u << nbits) - 1u));
        bitcnt -= nbits;
        if (code == LZW_CLEAR) {
            next = LZW_FIRST;
            nbits = 9;
            prev = -1;
            continue;
        }
        if (code == LZW_EOI) return op;
        if (prev < 0) {  /* first code after a clear: a literal, no table entry */
            if (code >= 256) return -1;
            dst[op] = (uint8_t)code;
            prev = code;
            prev_pos = op++;
            prev_len = 1;
            continue;
        }
        if (code > next || (code == next && next >= LZW_TABLE)) return -1;
        if (next < LZW_TABLE) {  /* entry `next` = string(prev) + the byte this iteration writes at dst[op]: contiguous from prev_pos */
            pos[next] = prev_pos;
            length[next] = prev_len + 1;
            ++next;
        }
        int32_t len;
        if (code < 256) {
            dst[op] = (uint8_t)code;
            len = 1;
        } else {
            const uint8_t* s = dst + pos[code];
            uint8_t* q = dst + op;
            len = length[code];
            int64_t n = len;
            if (op + n > dst_cap) n = dst_cap - op;
            if (code != next - 1 && op + n + 8 <= dst_cap) {  /* a finished entry: its bytes end at or before q, whole words may spill into the slack */
                for (int64_t k = 0; k < n; k += 8) {  /* (through a register: source and destination words may overlap past the string's end) */
                    uint64_t w;
                    memcpy(&w, s + k, 8);
                    memcpy(q + k, &w, 8);
                }
            } else {
                for (int64_t k = 0; k < n; ++k) q[k] = s[k];
            }
        }
        prev = code;
        prev_pos = op;
        prev_len = len;
        op += len;
        nbits = next >= 2047 ? 12 : next >= 1023 ? 11 : next >= 511 ? 10 : 9;
    }
    return op < dst_cap ? op : dst_cap;
}

/* TIFF 6.0 section 9 PackBits: header n in 0..127 -> n + 1 literal bytes, 129..255 -> the next byte 257 - n times, 128 -> no operation.
 * -> bytes written (<= dst_cap), -1 for a run or literal that reads past the input. */
int64_t cerb_host_packbits_decode(const uint8_t* src, int64_t n_src, uint8_t* dst, int64_t dst_cap) {
    int64_t ip = 0, op = 0;
    while (ip < n_src && op < dst_cap) {
        const int h = src[ip++];
        if (h < 128) {
            int64_t n = h + 1;
            if (ip + n > n_src) return -1;
            if (op + n > dst_cap) n = dst_cap - op;
            memcpy(dst + op, src + ip, (size_t)n);
            ip += h + 1;
            op += n;
        } else if (h > 128) {
            int64_t n = 257 - h;
            if (ip >= n_src) return -1;
            if (op + n > dst_cap) n = dst_cap - op;
            memset(dst + op, src[ip], (size_t)n);
            ++ip;
            op += n;
        }
    }
    return op;
}

/* TIFF 6.0 section 14, Predictor 2 on 8-bit chunky samples: every row stores differences to the pixel on its left, per sample, modulo 256.
 * In place over rows x cols pixels of `samples` bytes. */
void cerb_host_unpredict_u8(uint8_t* px, int64_t rows, int64_t cols, int samples) {
    const int64_t stride = cols * samples;
    for (int64_t r = 0; r < rows; ++r) {
        uint8_t* row = px + r * stride;
        for (int64_t i = samples; i < stride; ++i) row[i] = (uint8_t)(row[i] + row[i - samples]);
    }
}

/* ---- a whole window of a TIFF level in one call ------------------------------------------------------------------------------------------------ */

typedef struct {
    int fd, codec, predictor, samples, tile_cols, n_tiles;
    const int64_t *offsets, *counts;
    const int32_t *rows, *gx0, *gy0;
    int x0, y0, x1, y1;
    uint8_t* out;
    int64_t out_stride;
    atomic_int next;
    atomic_int err;       /* first error kind (negative), 0 while fine */
    atomic_int bad_tile;
} tile_job;

static int64_t pread_all(int fd, uint8_t* buf, int64_t n, int64_t off) {
    int64_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, buf + got, (size_t)(n - got), (off_t)(off + got));
        if (r <= 0) break;
        got += r;
    }
    return got;
}

static void* tile_worker(void* arg) {
    tile_job* j = (tile_job*)arg;
    uint8_t *raw = NULL, *pix = NULL;
    int64_t raw_cap = 0, pix_cap = 0;
    for (;;) {
        const int i = atomic_fetch_add(&j->next, 1);
        if (i >= j->n_tiles || atomic_load(&j->err)) break;
        int kind = 0;
        const int64_t need = (int64_t)j->rows[i] * j->tile_cols * j->samples;
        const int64_t cnt = j->counts[i];
        if (need > pix_cap) {
            free(pix);
            pix = (uint8_t*)malloc((size_t)need + 8);
            pix_cap = pix ? need : 0;
        }
        if (cnt > raw_cap) {
            free(raw);
            raw = (uint8_t*)malloc((size_t)cnt + 8);
            raw_cap = raw ? cnt : 0;
        }
        if ((need > 0 && !pix) || (cnt > 0 && !raw)) {
            kind = -6;
        } else if (pread_all(j->fd, raw, cnt, j->offsets[i]) != cnt) {
            kind = -3;
        } else if (j->codec == 1) {
            if (cnt < need) kind = -4;
            else memcpy(pix, raw, (size_t)need);
        } else if (j->codec == 8 || j->codec == 32946) {
            uLongf dl = (uLongf)need;
            const int z = uncompress(pix, &dl, raw, (uLong)cnt);
            if (z == Z_BUF_ERROR && (int64_t)dl == need) { /* a stream longer than the tile: its first `need` bytes are what the tile holds */
            } else if (z != Z_OK) kind = -5;
            else if ((int64_t)dl < need) kind = -4;
        } else if (j->codec == 5 || j->codec == 32773) {
            const int64_t n = j->codec == 5 ? cerb_host_lzw_decode(raw, cnt, pix, need) : cerb_host_packbits_decode(raw, cnt, pix, need);
            if (n < 0) kind = (int)n;                    /* -1 corrupt, -2 old-style LZW */
            else if (n < need) memset(pix + n, 0, (size_t)(need - n));
        } else {
            kind = -7;
        }
        if (kind) {
            int zero = 0;
            if (atomic_compare_exchange_strong(&j->err, &zero, kind)) atomic_store(&j->bad_tile, i);
            break;
        }
        if (j->predictor == 2) cerb_host_unpredict_u8(pix, j->rows[i], j->tile_cols, j->samples);
        const int gy0 = j->gy0[i], gx0 = j->gx0[i];
        const int a0 = j->y0 > gy0 ? j->y0 : gy0, a1 = j->y1 < gy0 + j->rows[i] ? j->y1 : gy0 + j->rows[i];
        const int b0 = j->x0 > gx0 ? j->x0 : gx0, b1 = j->x1 < gx0 + j->tile_cols ? j->x1 : gx0 + j->tile_cols;
        if (a1 <= a0 || b1 <= b0) continue;
        const int sp = j->samples;
        for (int y = a0; y < a1; ++y) {
            const uint8_t* srow = pix + ((int64_t)(y - gy0) * j->tile_cols + (b0 - gx0)) * sp;
            uint8_t* drow = j->out + (int64_t)(y - j->y0) * j->out_stride + (int64_t)(b0 - j->x0) * 3;
            if (sp == 3) {
                memcpy(drow, srow, (size_t)(b1 - b0) * 3);
            } else {
                for (int x = 0; x < b1 - b0; ++x) {
                    drow[3 * x] = srow[sp * x];
                    drow[3 * x + 1] = srow[sp * x + 1];
                    drow[3 * x + 2] = srow[sp * x + 2];
                }
            }
        }
    }
    free(raw);
    free(pix);
    return NULL;
}

int cerb_host_tiff_read_tiles(int fd, int codec, int predictor, int samples, int tile_cols, int n_tiles, const int64_t* offsets, const int64_t* counts,
                              const int32_t* rows, const int32_t* gx0, const int32_t* gy0, int x0, int y0, int x1, int y1, uint8_t* out,
                              int64_t out_stride, int n_threads, int32_t* bad_tile) {
    if (samples < 3 || tile_cols <= 0 || n_tiles < 0 || x1 < x0 || y1 < y0 || out_stride < (int64_t)(x1 - x0) * 3) return -8;
    if (!(codec == 1 || codec == 5 || codec == 8 || codec == 32946 || codec == 32773)) return -7;
    tile_job j;
    j.fd = fd; j.codec = codec; j.predictor = predictor; j.samples = samples; j.tile_cols = tile_cols; j.n_tiles = n_tiles;
    j.offsets = offsets; j.counts = counts; j.rows = rows; j.gx0 = gx0; j.gy0 = gy0;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1; j.out = out; j.out_stride = out_stride;
    atomic_init(&j.next, 0);
    atomic_init(&j.err, 0);
    atomic_init(&j.bad_tile, -1);
    if (n_threads > n_tiles) n_threads = n_tiles;
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    int started = 0;
    for (int t = 1; t < n_threads; ++t) {
        if (pthread_create(&th[started], NULL, tile_worker, &j) != 0) break;  /* fewer threads than asked for: the others take the tiles */
        ++started;
    }
    tile_worker(&j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (bad_tile) *bad_tile = atomic_load(&j.bad_tile);
    return atomic_load(&j.err);
}
