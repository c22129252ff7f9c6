/* Host half of the split JPEG decode: entropy decoding only.
 *
 * The tiles the detector sweeps are baseline JPEGs written by GDAL's JPEG driver (reference src/load_data/tile_tifs.py:66-74: 8-bit,
 * 3 bands, YCbCr, 4:2:0, Huffman) and [UPSTREAM detect.py LoadImages -> cv2.imread] decodes them with libjpeg(-turbo) on one host
 * thread.  A full libjpeg decode of a 640-px tile costs ~2-3 ms of a core, most of it in the inverse DCT, the chroma upsampling and the
 * colour conversion -- dense integer arithmetic that belongs on the GPU (csrc/jpeg_idct.hip restates libjpeg's ISLOW IDCT, fancy
 * h2v2 upsampling and YCbCr -> RGB bit for bit).  What cannot go there cheaply is the serial Huffman bit stream: this file decodes it into
 * quantised DCT coefficient blocks and nothing else.
 *
 * Plain C, no libjpeg (its headers are not in the image), no HIP: built by aquaculture_amd/build.py into libaqjpeg.so with gcc and loaded
 * by the decode worker PROCESSES through ctypes (they never touch the GPU).
 *
 * Supported: baseline sequential DCT (SOF0) or extended sequential with 8-bit samples (SOF1), Huffman coding, 3 components YCbCr with
 * sampling 2x2 / 1x1 / 1x1 (4:2:0) or 1 component (greyscale), restart intervals, one interleaved scan.  Everything else (progressive,
 * arithmetic, 4:2:2, 4:4:4, CMYK, 12-bit) returns AQJ_UNSUPPORTED and the caller falls back to the full software decode.
 *
 * Output layout (int16, quantised coefficients in NATURAL (row-major, de-zigzagged) order, 64 per block):
 *   Y  blocks [by = 0 .. Hb-1][bx = 0 .. Wb-1], Hb = 2 * mcu_rows, Wb = 2 * mcu_cols (4:2:0) -- i.e. the padded image in 8x8 blocks
 *   Cb blocks [mcu_rows][mcu_cols], then Cr blocks [mcu_rows][mcu_cols]
 * plus the three quantisation tables (uint16 [3][64], natural order) -- the GPU multiplies.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define AQJ_OK 0
#define AQJ_UNSUPPORTED (-1)
#define AQJ_CORRUPT (-2)
#define AQJ_SPACE (-3)

typedef struct aq_jpeg_info {
    int32_t width, height;        /* image size in pixels */
    int32_t ncomp;                /* 1 or 3 */
    int32_t mcu_cols, mcu_rows;   /* 16x16 MCUs (3 components) or 8x8 (greyscale) */
    int32_t y_blocks_w, y_blocks_h;
    int32_t total_blocks;         /* blocks written to coef_out */
    uint16_t qt[3][64];           /* quantisation table of each component, natural order */
} aq_jpeg_info;

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

#define LOOK 9
typedef struct {
    uint16_t look[1 << LOOK];     /* (length << 8) | symbol for codes of <= LOOK bits, 0 = longer */
    int16_t fast[1 << LOOK];      /* AC tables: (value << 8) | (run << 4) | (code + magnitude bits) when both fit in LOOK bits, else 0 */
    int32_t maxcode[18];          /* largest code of each length (-1: none), [17] = sentinel */
    int32_t valoff[17];           /* huffval index of the first code of a length minus that code */
    uint8_t vals[256];
    int present;
} HuffTab;

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc;                 /* bit accumulator, MSB first in the low `nbits` bits */
    int nbits;
    int marker;                   /* a marker was met: only zero bits follow */
    int fake;                     /* zero bits appended after a marker / the end of the data; nbits < fake <=> the decoder has consumed
                                     bits the file does not contain (truncated or marker-interrupted scan) */
} Bits;

static void fill(Bits* b) {
    while (b->nbits <= 56) {
        unsigned c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0x00) {
                    b->p += 2;                                  /* stuffed zero */
                } else {
                    b->marker = 1;                              /* a real marker (RSTn, EOI): leave it for the caller */
                    c = 0;
                }
            } else {
                b->p += 1;
            }
        } else if (!b->marker) {
            b->marker = 1;
        }
        if (b->marker) b->fake += 8;
        b->acc = (b->acc << 8) | c;
        b->nbits += 8;
    }
}

static int build(HuffTab* t, const uint8_t* counts, const uint8_t* vals, int nvals) {
    int code = 0, k = 0;
    memset(t->look, 0, sizeof t->look);
    memset(t->vals, 0, sizeof t->vals);                         /* (the GPU path hashes and uploads whole tables: no stack bytes in them) */
    memcpy(t->vals, vals, (size_t)nvals);
    for (int len = 1; len <= 16; ++len) {
        t->valoff[len] = k - code;
        /* over-subscribed length: refuse BEFORE the look-up table is written (codes >= 2^len index past look[]) */
        if (code + counts[len - 1] > (1 << len)) return AQJ_CORRUPT;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (k >= nvals) return AQJ_CORRUPT;
            if (len <= LOOK) {
                const int first = code << (LOOK - len), n = 1 << (LOOK - len);
                for (int j = 0; j < n; ++j) t->look[first + j] = (uint16_t)((len << 8) | vals[k]);
            }
        }
        t->maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    t->present = 1;
    for (int i = 0; i < (1 << LOOK); ++i) {                     /* (only read through AC tables; harmless for DC ones) */
        t->fast[i] = 0;
        const unsigned e = t->look[i];
        if (!e) continue;
        const int len = (int)(e >> 8), run = (int)(e & 0xff) >> 4, mag = (int)(e & 15);
        if (mag == 0 || len + mag > LOOK) continue;
        int v = (i >> (LOOK - len - mag)) & ((1 << mag) - 1);   /* the magnitude bits that follow the code */
        if (v < (1 << (mag - 1))) v += 1 - (1 << mag);          /* (receive / extend) */
        if (v < -128 || v > 127) continue;
        t->fast[i] = (int16_t)(v * 256 + run * 16 + len + mag);
    }
    return AQJ_OK;
}

/* One block.  The bit reader lives in locals for the whole block (the compiler keeps `b->acc` / `b->nbits` in memory otherwise: every
 * symbol then costs a load and a store of each), and most AC coefficients take ONE table look-up: fast[] maps the next FAST bits to
 * (value, run, total bits) whenever a code and its magnitude bits fit in them.  17.5 -> about 6 ns per non-zero coefficient. */
#define REFILL()                                                                                                   \
    do {                                                                                                           \
        if (nbits <= 32) {                                                                                         \
            uint32_t w_;                                                                                           \
            if (!b->marker && p + 4 <= b->end && (memcpy(&w_, p, 4), !(((~w_) - 0x01010101u) & w_ & 0x80808080u))) { \
                acc = (acc << 32) | __builtin_bswap32(w_);                                                         \
                nbits += 32;                                                                                       \
                p += 4;                                                                                            \
            } else {                                                                                               \
                b->p = p; b->acc = acc; b->nbits = nbits;                                                          \
                fill(b);                                                                                           \
                p = b->p; acc = b->acc; nbits = b->nbits;                                                          \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
#define PEEK(n) ((unsigned)((acc >> (nbits - (n))) & ((1u << (n)) - 1u)))

static int decode_block(Bits* b, const HuffTab* dc, const HuffTab* ac, int* pred, int16_t* out) {      /* out: 64 zeros on entry */
    const uint8_t* p = b->p;
    uint64_t acc = b->acc;
    int nbits = b->nbits;
    int rc = AQJ_OK;
    REFILL();                              /* a symbol (<= 16 bits) and its magnitude bits (<= 15) fit the 32 guaranteed bits */
    int s;
    {
        const unsigned e = dc->look[PEEK(LOOK)];
        if (e) {
            nbits -= (int)(e >> 8);
            s = (int)(e & 0xff);
        } else {
            int len = LOOK + 1, code = (int)PEEK(len);
            while (len <= 16 && code > dc->maxcode[len]) { ++len; code = (int)PEEK(len); }
            if (len > 16) { rc = AQJ_CORRUPT; goto done; }
            nbits -= len;
            s = dc->vals[(code + dc->valoff[len]) & 0xff];
        }
    }
    if (s > 11) { rc = AQJ_CORRUPT; goto done; }
    if (s) {
        const int v = (int)PEEK(s);
        nbits -= s;
        *pred += v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    out[0] = (int16_t)*pred;
    for (int k = 1; k < 64;) {
        REFILL();
        const unsigned idx = PEEK(LOOK);
        const int f = ac->fast[idx];
        if (f) {                           /* code + magnitude bits inside the look-ahead: (value << 8) | (run << 4) | bits */
            k += (f >> 4) & 15;
            if (k > 63) { rc = AQJ_CORRUPT; goto done; }
            nbits -= f & 15;
            out[kZigzag[k++]] = (int16_t)(f >> 8);
            continue;
        }
        int rs;
        const unsigned e = ac->look[idx];
        if (e) {
            nbits -= (int)(e >> 8);
            rs = (int)(e & 0xff);
        } else {
            int len = LOOK + 1, code = (int)PEEK(len);
            while (len <= 16 && code > ac->maxcode[len]) { ++len; code = (int)PEEK(len); }
            if (len > 16) { rc = AQJ_CORRUPT; goto done; }
            nbits -= len;
            rs = ac->vals[(code + ac->valoff[len]) & 0xff];
        }
        const int r = rs >> 4;
        s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) { rc = AQJ_CORRUPT; goto done; }
            const int v = (int)PEEK(s);
            nbits -= s;
            out[kZigzag[k++]] = (int16_t)(v < (1 << (s - 1)) ? v - (1 << s) + 1 : v);
        } else {
            if (r != 15) break;            /* EOB */
            k += 16;                       /* ZRL */
        }
    }
done:
    b->p = p; b->acc = acc; b->nbits = nbits;
    /* bits past a marker or the end of the file were consumed: the scan is truncated or interrupted.  (fill() pads with zeros so that the
     * bit reader never reads out of bounds; a decoder that went on would turn them into plausible grey blocks -- PIL raises here.) */
    if (rc == AQJ_OK && nbits < b->fake) rc = AQJ_CORRUPT;
    return rc;
}
#undef REFILL
#undef PEEK

static inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

/* Decodes the entropy-coded data of one JPEG file into coef_out (capacity `cap` int16 values).  Returns AQJ_OK or a negative code;
 * info is filled as far as the headers were read. */
int aq_jpeg_decode_coeffs(const uint8_t* data, size_t n, int16_t* coef_out, size_t cap, aq_jpeg_info* info) {
    if (!data || !info || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return AQJ_CORRUPT;
    memset(info, 0, sizeof *info);
    uint16_t qt[4][64];
    int qt_present[4] = {0, 0, 0, 0};
    HuffTab hdc[4], hac[4];
    for (int i = 0; i < 4; ++i) hdc[i].present = hac[i].present = 0;
    int comp_id[3] = {0, 0, 0}, comp_h[3] = {0, 0, 0}, comp_v[3] = {0, 0, 0}, comp_q[3] = {0, 0, 0};
    int restart = 0, have_sof = 0;
    int saw_jfif = 0, saw_adobe = 0, adobe_transform = 0;      /* colour-space signalling, as libjpeg's default_decompress_parms reads it */
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) return AQJ_CORRUPT;
        const unsigned m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }                     /* fill byte */
        pos += 2;
        if (m == 0xD9) return AQJ_CORRUPT;                      /* EOI before any scan */
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;    /* standalone markers */
        if (pos + 2 > n) return AQJ_CORRUPT;
        const unsigned len = be16(data + pos);
        if (len < 2 || pos + len > n) return AQJ_CORRUPT;
        const uint8_t* seg = data + pos + 2;
        const unsigned sl = len - 2;
        if (m == 0xDB) {                                        /* DQT */
            unsigned o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (tq > 3 || pq > 1) return AQJ_CORRUPT;
                ++o;
                if (o + (pq ? 128u : 64u) > sl) return AQJ_CORRUPT;
                for (int k = 0; k < 64; ++k) {
                    qt[tq][kZigzag[k]] = (uint16_t)(pq ? be16(seg + o + 2 * k) : seg[o + k]);
                }
                o += pq ? 128 : 64;
                qt_present[tq] = 1;
            }
        } else if (m == 0xC4) {                                 /* DHT */
            unsigned o = 0;
            while (o + 17 <= sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3) return AQJ_CORRUPT;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += seg[o + 1 + k];
                if (cnt > 256 || o + 17 + (unsigned)cnt > sl) return AQJ_CORRUPT;
                const int rc = build(tc ? &hac[th] : &hdc[th], seg + o + 1, seg + o + 17, cnt);
                if (rc) return rc;
                o += 17 + (unsigned)cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {                    /* SOF0 / SOF1: sequential, Huffman */
            if (sl < 6 || seg[0] != 8) return AQJ_UNSUPPORTED;
            info->height = (int32_t)be16(seg + 1);
            info->width = (int32_t)be16(seg + 3);
            info->ncomp = seg[5];
            if ((info->ncomp != 1 && info->ncomp != 3) || sl < 6u + 3u * (unsigned)info->ncomp || info->width <= 0 || info->height <= 0)
                return AQJ_UNSUPPORTED;
            for (int c = 0; c < info->ncomp; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                comp_h[c] = seg[7 + 3 * c] >> 4;
                comp_v[c] = seg[7 + 3 * c] & 15;
                comp_q[c] = seg[8 + 3 * c];
                if (comp_q[c] > 3) return AQJ_CORRUPT;
            }
            if (info->ncomp == 3 && !(comp_h[0] == 2 && comp_v[0] == 2 && comp_h[1] == 1 && comp_v[1] == 1 && comp_h[2] == 1 && comp_v[2] == 1))
                return AQJ_UNSUPPORTED;                          /* only 4:2:0 */
            if (info->ncomp == 1) comp_h[0] = comp_v[0] = 1;     /* a single component is never interleaved */
            have_sof = 1;
        } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return AQJ_UNSUPPORTED;                              /* progressive, lossless, arithmetic */
        } else if (m == 0xE0) {                                  /* APP0: "JFIF\0" means YCbCr whatever else the file says */
            if (sl >= 5 && !memcmp(seg, "JFIF", 5)) saw_jfif = 1;
        } else if (m == 0xEE) {                                  /* APP14: "Adobe" + version, flags0, flags1, transform (0 = RGB / CMYK as stored) */
            if (sl >= 12 && !memcmp(seg, "Adobe", 5)) { saw_adobe = 1; adobe_transform = seg[11]; }
        } else if (m == 0xDD) {                                  /* DRI */
            if (sl < 2) return AQJ_CORRUPT;
            restart = (int)be16(seg);
        } else if (m == 0xDA) {                                  /* SOS: the one scan */
            if (!have_sof || sl < 1 || seg[0] != info->ncomp || sl < 1u + 2u * (unsigned)info->ncomp + 3u) return AQJ_UNSUPPORTED;
            if (info->ncomp == 3 && !saw_jfif) {
                /* libjpeg (jdapimin.c) and therefore Pillow / OpenCV treat the three components as RGB -- no colour conversion -- when an
                 * Adobe marker says transform 0, or when there is no JFIF / Adobe marker and the component ids spell "RGB".  The device half
                 * always converts YCbCr -> RGB, so such files go to the software decoder. */
                if (saw_adobe ? adobe_transform == 0 : (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B')) return AQJ_UNSUPPORTED;
            }
            int tdc[3], tac[3];
            for (int c = 0; c < info->ncomp; ++c) {
                if (seg[1 + 2 * c] != comp_id[c]) return AQJ_UNSUPPORTED;
                tdc[c] = seg[2 + 2 * c] >> 4;
                tac[c] = seg[2 + 2 * c] & 15;
                if (tdc[c] > 3 || tac[c] > 3 || !hdc[tdc[c]].present || !hac[tac[c]].present || !qt_present[comp_q[c]]) return AQJ_CORRUPT;
                memcpy(info->qt[c], qt[comp_q[c]], sizeof info->qt[c]);
            }
            const int mcu_px = info->ncomp == 3 ? 16 : 8;
            info->mcu_cols = (info->width + mcu_px - 1) / mcu_px;
            info->mcu_rows = (info->height + mcu_px - 1) / mcu_px;
            const int yb = info->ncomp == 3 ? 2 : 1;
            info->y_blocks_w = info->mcu_cols * yb;
            info->y_blocks_h = info->mcu_rows * yb;
            const size_t ny = (size_t)info->y_blocks_w * info->y_blocks_h, nc = (size_t)info->mcu_cols * info->mcu_rows;
            const size_t total = ny + (info->ncomp == 3 ? 2 * nc : 0);
            info->total_blocks = (int32_t)total;
            if (!coef_out || total * 64 > cap) return AQJ_SPACE;
            memset(coef_out, 0, total * 64 * sizeof(int16_t));     /* one pass over the whole image: the blocks only write what is non-zero */
            Bits b;
            b.p = data + pos + len; b.end = data + n; b.acc = 0; b.nbits = 0; b.marker = 0; b.fake = 0;
            int pred[3] = {0, 0, 0};
            int left = restart, next_rst = 0;
            int16_t* cb = coef_out + ny * 64;
            int16_t* cr = cb + nc * 64;
            for (int my = 0; my < info->mcu_rows; ++my)
                for (int mx = 0; mx < info->mcu_cols; ++mx) {
                    if (restart && left == 0) {                 /* expect RSTn: byte align, skip the marker, reset the predictors */
                        b.nbits = 0; b.acc = 0; b.fake = 0;
                        const uint8_t* q = b.p;
                        while (q + 1 < b.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                            if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return AQJ_CORRUPT;
                            ++q;
                        }
                        if (q + 1 >= b.end || q[1] != (uint8_t)(0xD0 + next_rst)) return AQJ_CORRUPT;
                        b.p = q + 2; b.marker = 0;
                        next_rst = (next_rst + 1) & 7;
                        pred[0] = pred[1] = pred[2] = 0;
                        left = restart;
                    }
                    int rc;
                    if (info->ncomp == 3) {
                        for (int v = 0; v < 2; ++v)
                            for (int h = 0; h < 2; ++h) {
                                rc = decode_block(&b, &hdc[tdc[0]], &hac[tac[0]], &pred[0],
                                                  coef_out + ((size_t)(2 * my + v) * info->y_blocks_w + (2 * mx + h)) * 64);
                                if (rc) return rc;
                            }
                        rc = decode_block(&b, &hdc[tdc[1]], &hac[tac[1]], &pred[1], cb + ((size_t)my * info->mcu_cols + mx) * 64);
                        if (rc) return rc;
                        rc = decode_block(&b, &hdc[tdc[2]], &hac[tac[2]], &pred[2], cr + ((size_t)my * info->mcu_cols + mx) * 64);
                        if (rc) return rc;
                    } else {
                        rc = decode_block(&b, &hdc[tdc[0]], &hac[tac[0]], &pred[0], coef_out + ((size_t)my * info->y_blocks_w + mx) * 64);
                        if (rc) return rc;
                    }
                    if (restart) --left;
                }
            /* the scan must be followed by EOI (fill() stops b.p at the first marker): a file cut right after its last MCU is as
             * truncated for libjpeg / Pillow ("image file is truncated") as one cut earlier */
            for (const uint8_t* q = b.p; q + 1 < b.end; ++q)
                if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return q[1] == 0xD9 ? AQJ_OK : AQJ_CORRUPT;
            return AQJ_CORRUPT;
        }
        pos += len;
    }
    return AQJ_CORRUPT;
}

/* Size query / header scan without decoding: fills info (total_blocks included) and returns AQJ_SPACE-free status. */
int aq_jpeg_scan(const uint8_t* data, size_t n, aq_jpeg_info* info) {
    const int rc = aq_jpeg_decode_coeffs(data, n, NULL, 0, info);
    return rc == AQJ_SPACE ? AQJ_OK : rc;
}


/* ---------------------------------------------------------------------------------------------------------------------------------------
 * GPU entropy decode (round 4; VERDICT r03 item 8b, SURVEY.md 8f rank 2 "optional GPU Huffman later"): the host keeps only what is
 * byte-serial and cheap -- header parsing and the removal of byte stuffing / restart markers while the file is copied into the upload
 * buffer -- and the Huffman decoding itself runs on the device (csrc/jpeg_huff.hip: one lane per restart segment, i.e. per image for
 * the files GDAL writes, which have no restart markers).  H2D then carries the 40-140 KB of a tile's entropy-coded data instead of its
 * 3 MB coefficient image, and a tile costs the host tens of microseconds instead of 1.0-2.4 ms.
 *
 * aq_jpeg_prepare: parses the headers exactly as aq_jpeg_decode_coeffs does (same refusals: anything but baseline / extended sequential
 * 8-bit YCbCr 4:2:0 is AQJ_UNSUPPORTED; colour signalling as libjpeg reads it), then writes to stream_out the scan's entropy-coded bytes
 * with every stuffed 0x00 removed, cut at the restart markers: segment i (MCUs i * restart .. of the scan; one segment without DRI) starts at
 * seg_off[i], a multiple of 16, and is followed by >= 8 zero bytes.  An unexpected marker inside the scan, a wrong RSTn sequence or a
 * missing EOI is AQJ_CORRUPT (what the bit reader's overrun check catches in the host decoder is caught by the device the same way).
 * tabs: the scan's six Huffman tables (DC / AC of the three components) in the device decoder's format. */
typedef struct aq_jpeg_gpu_tab {
    uint16_t look[1 << LOOK];     /* (length << 8) | symbol for codes of <= 9 bits, 0 = longer */
    int32_t maxcode[18];
    int32_t valoff[18];           /* (17 used) */
    uint8_t vals[256];
} aq_jpeg_gpu_tab;                /* 1024 + 72 + 72 + 256 = 1424 bytes */

typedef struct aq_jpeg_stream_info {
    int32_t width, height, mcu_cols, mcu_rows;
    int32_t restart;              /* MCUs per restart interval, 0 = none */
    int32_t nseg;                 /* segments written */
    uint32_t stream_bytes;        /* bytes used in stream_out */
    uint32_t pad;
    uint64_t tab_hash;            /* FNV-1a of tabs: equal hashes = equal tables (callers share one device copy) */
    uint16_t qt[3][64];
    aq_jpeg_gpu_tab tabs[6];      /* [component][dc, ac] */
} aq_jpeg_stream_info;

int aq_jpeg_prepare(const uint8_t* data, size_t n, uint8_t* stream_out, size_t cap, uint32_t* seg_off, uint32_t* seg_len, int seg_cap,
                    aq_jpeg_stream_info* si) {
    if (!data || !si || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return AQJ_CORRUPT;
    memset(si, 0, sizeof *si);
    uint16_t qt[4][64];
    int qt_present[4] = {0, 0, 0, 0};
    HuffTab hdc[4], hac[4];
    for (int i = 0; i < 4; ++i) hdc[i].present = hac[i].present = 0;
    int comp_id[3] = {0, 0, 0}, comp_h[3] = {0, 0, 0}, comp_v[3] = {0, 0, 0}, comp_q[3] = {0, 0, 0};
    int restart = 0, have_sof = 0, ncomp = 0;
    int saw_jfif = 0, saw_adobe = 0, adobe_transform = 0;
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) return AQJ_CORRUPT;
        const unsigned m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD9) return AQJ_CORRUPT;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) return AQJ_CORRUPT;
        const unsigned len = be16(data + pos);
        if (len < 2 || pos + len > n) return AQJ_CORRUPT;
        const uint8_t* seg = data + pos + 2;
        const unsigned sl = len - 2;
        if (m == 0xDB) {
            unsigned o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (tq > 3 || pq > 1) return AQJ_CORRUPT;
                ++o;
                if (o + (pq ? 128u : 64u) > sl) return AQJ_CORRUPT;
                for (int k = 0; k < 64; ++k) qt[tq][kZigzag[k]] = (uint16_t)(pq ? be16(seg + o + 2 * k) : seg[o + k]);
                o += pq ? 128 : 64;
                qt_present[tq] = 1;
            }
        } else if (m == 0xC4) {
            unsigned o = 0;
            while (o + 17 <= sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3) return AQJ_CORRUPT;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += seg[o + 1 + k];
                if (cnt > 256 || o + 17 + (unsigned)cnt > sl) return AQJ_CORRUPT;
                const int rc = build(tc ? &hac[th] : &hdc[th], seg + o + 1, seg + o + 17, cnt);
                if (rc) return rc;
                o += 17 + (unsigned)cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || seg[0] != 8) return AQJ_UNSUPPORTED;
            si->height = (int32_t)be16(seg + 1);
            si->width = (int32_t)be16(seg + 3);
            ncomp = seg[5];
            if (ncomp != 3 || sl < 6u + 9u || si->width <= 0 || si->height <= 0) return AQJ_UNSUPPORTED;
            for (int c = 0; c < 3; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                comp_h[c] = seg[7 + 3 * c] >> 4;
                comp_v[c] = seg[7 + 3 * c] & 15;
                comp_q[c] = seg[8 + 3 * c];
                if (comp_q[c] > 3) return AQJ_CORRUPT;
            }
            if (!(comp_h[0] == 2 && comp_v[0] == 2 && comp_h[1] == 1 && comp_v[1] == 1 && comp_h[2] == 1 && comp_v[2] == 1)) return AQJ_UNSUPPORTED;
            have_sof = 1;
        } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return AQJ_UNSUPPORTED;
        } else if (m == 0xE0) {
            if (sl >= 5 && !memcmp(seg, "JFIF", 5)) saw_jfif = 1;
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(seg, "Adobe", 5)) { saw_adobe = 1; adobe_transform = seg[11]; }
        } else if (m == 0xDD) {
            if (sl < 2) return AQJ_CORRUPT;
            restart = (int)be16(seg);
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1 || seg[0] != 3 || sl < 1u + 6u + 3u) return AQJ_UNSUPPORTED;
            if (!saw_jfif && (saw_adobe ? adobe_transform == 0 : (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B'))) return AQJ_UNSUPPORTED;
            for (int c = 0; c < 3; ++c) {
                if (seg[1 + 2 * c] != comp_id[c]) return AQJ_UNSUPPORTED;
                const int td = seg[2 + 2 * c] >> 4, ta = seg[2 + 2 * c] & 15;
                if (td > 3 || ta > 3 || !hdc[td].present || !hac[ta].present || !qt_present[comp_q[c]]) return AQJ_CORRUPT;
                memcpy(si->qt[c], qt[comp_q[c]], sizeof si->qt[c]);
                const HuffTab* src[2] = {&hdc[td], &hac[ta]};
                for (int k = 0; k < 2; ++k) {
                    aq_jpeg_gpu_tab* t = &si->tabs[2 * c + k];
                    memcpy(t->look, src[k]->look, sizeof t->look);
                    memcpy(t->maxcode, src[k]->maxcode, sizeof t->maxcode);
                    memcpy(t->valoff, src[k]->valoff, 17 * sizeof(int32_t));
                    t->valoff[0] = 0; t->valoff[17] = 0;
                    t->maxcode[0] = -1;
                    memcpy(t->vals, src[k]->vals, 256);
                }
            }
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < sizeof si->tabs; ++i) { h ^= ((const uint8_t*)si->tabs)[i]; h *= 1099511628211ull; }
            si->tab_hash = h;
            si->mcu_cols = (si->width + 15) / 16;
            si->mcu_rows = (si->height + 15) / 16;
            si->restart = restart;
            const long long total_mcu = (long long)si->mcu_cols * si->mcu_rows;
            const long long want_seg = restart ? (total_mcu + restart - 1) / restart : 1;
            if (!stream_out || !seg_off || !seg_len || want_seg > seg_cap) { si->nseg = (int32_t)want_seg; return AQJ_SPACE; }
            /* the scan: copy, dropping stuffed zeros; cut at RSTn (which must count 0..7 in order); stop at EOI */
            const uint8_t* p = data + pos + len;
            const uint8_t* end = data + n;
            size_t o = 0;
            int nseg = 0, next_rst = 0, saw_eoi = 0;
            seg_off[0] = 0;
            while (p < end) {
                const uint8_t* ff = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
                const size_t run = (size_t)((ff ? ff : end) - p);
                if (o + run + 48 > cap) return AQJ_SPACE;
                memcpy(stream_out + o, p, run);
                o += run;
                if (!ff) { p = end; break; }
                if (ff + 1 >= end) return AQJ_CORRUPT;
                const unsigned mk = ff[1];
                if (mk == 0x00) { stream_out[o++] = 0xFF; p = ff + 2; continue; }          /* stuffed: a data byte 0xFF */
                if (mk == 0xFF) { p = ff + 1; continue; }                                  /* fill byte */
                if (mk >= 0xD0 && mk <= 0xD7) {
                    if (!restart || mk != (unsigned)(0xD0 + next_rst) || nseg + 1 >= want_seg) return AQJ_CORRUPT;
                    next_rst = (next_rst + 1) & 7;
                    seg_len[nseg] = (uint32_t)(o - seg_off[nseg]);
                    if (o + 40 > cap) return AQJ_SPACE;
                    memset(stream_out + o, 0, 24);
                    o = (o + 8 + 15) & ~(size_t)15;                                        /* >= 8 zero bytes, next segment 16-byte aligned */
                    seg_off[++nseg] = (uint32_t)o;
                    p = ff + 2;
                    continue;
                }
                if (mk == 0xD9) { saw_eoi = 1; break; }
                return AQJ_CORRUPT;                                                        /* any other marker inside the scan */
            }
            if (!saw_eoi || nseg + 1 != want_seg) return AQJ_CORRUPT;
            seg_len[nseg] = (uint32_t)(o - seg_off[nseg]);
            if (o + 40 > cap) return AQJ_SPACE;
            memset(stream_out + o, 0, 24);
            o = (o + 8 + 15) & ~(size_t)15;
            si->nseg = nseg + 1;
            si->stream_bytes = (uint32_t)o;
            return AQJ_OK;
        }
        pos += len;
    }
    return AQJ_CORRUPT;
}


/* A whole super-batch at once, without the interpreter in the loop: reads `n` files and prepares each into its slot of the upload buffer
 * (slot i = per_image bytes at streams + i * per_image), on `nthreads` POSIX threads (files i = t, t + nthreads, ...).  Per image: status[i]
 * (AQJ_*; a file of another size than W x H is AQJ_UNSUPPORTED with nseg[i] = -1), nseg[i] segments in segs[i * seg_cap ..] (stream_off
 * relative to `streams`, coef_off = i * coef_per_image, tabset left 0 for the caller), qt[i], hash[i] and tabs[i] (the caller keeps one copy
 * per distinct hash).  Returns the number of files whose status is not AQJ_OK. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct aq_jpeg_seg {
    uint32_t stream_off, stream_len;
    uint64_t coef_off;
    uint32_t mcu0, n_mcu;
    uint16_t mcu_cols, mcu_rows;
    uint32_t tabset;
} aq_jpeg_seg;

typedef struct {
    const char* const* paths; int n, H, W; uint8_t* streams; size_t per_image; aq_jpeg_seg* segs; int seg_cap; uint64_t coef_per_image;
    int32_t* status; int32_t* nseg; uint16_t* qt; uint64_t* hash; aq_jpeg_gpu_tab* tabs; int t, nthreads;
} PrepJob;

static void* prep_thread(void* arg) {
    PrepJob* j = (PrepJob*)arg;
    uint8_t* buf = NULL;
    size_t cap = 0;
    uint32_t* off = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)j->seg_cap);
    uint32_t* len = off ? off + j->seg_cap : NULL;
    aq_jpeg_stream_info* si = (aq_jpeg_stream_info*)malloc(sizeof *si);
    for (int i = j->t; i < j->n; i += j->nthreads) {
        j->status[i] = AQJ_CORRUPT; j->nseg[i] = 0;
        if (!off || !si) continue;
        FILE* f = fopen(j->paths[i], "rb");
        if (!f) continue;
        size_t got = 0;
        for (;;) {
            if (got + 65536 > cap) {
                const size_t ncap = cap ? cap * 2 : (size_t)1 << 18;
                uint8_t* nb = (uint8_t*)realloc(buf, ncap);
                if (!nb) { got = 0; break; }
                buf = nb; cap = ncap;
            }
            const size_t r = fread(buf + got, 1, cap - got, f);
            got += r;
            if (r == 0) break;
        }
        fclose(f);
        if (!got) continue;
        const size_t base = (size_t)i * j->per_image;
        const int rc = aq_jpeg_prepare(buf, got, j->streams + base, j->per_image, off, len, j->seg_cap, si);
        j->status[i] = rc;
        if (rc != AQJ_OK) continue;
        if (si->width != j->W || si->height != j->H) { j->status[i] = AQJ_UNSUPPORTED; j->nseg[i] = -1; continue; }
        j->nseg[i] = si->nseg;
        const uint32_t per_seg = si->restart ? (uint32_t)si->restart : (uint32_t)(si->mcu_cols * si->mcu_rows);
        const uint32_t total = (uint32_t)(si->mcu_cols * si->mcu_rows);
        for (int k = 0; k < si->nseg; ++k) {
            aq_jpeg_seg* sg = &j->segs[(size_t)i * j->seg_cap + k];
            sg->stream_off = (uint32_t)(base + off[k]);
            sg->stream_len = len[k];
            sg->coef_off = (uint64_t)i * j->coef_per_image;
            sg->mcu0 = (uint32_t)k * per_seg;
            sg->n_mcu = sg->mcu0 + per_seg <= total ? per_seg : total - sg->mcu0;
            sg->mcu_cols = (uint16_t)si->mcu_cols; sg->mcu_rows = (uint16_t)si->mcu_rows;
            sg->tabset = 0;
        }
        memcpy(j->qt + (size_t)i * 192, si->qt, 384);
        j->hash[i] = si->tab_hash;
        memcpy(&j->tabs[(size_t)i * 6], si->tabs, sizeof si->tabs);
    }
    free(buf); free(off); free(si);
    return NULL;
}

int aq_jpeg_prepare_files(const char* const* paths, int n, int H, int W, uint8_t* streams, size_t per_image, void* segs, int seg_cap,
                          uint64_t coef_per_image, int32_t* status, int32_t* nseg, uint16_t* qt, uint64_t* hash, void* tabs, int nthreads) {
    if (!paths || n <= 0 || !streams || !segs || seg_cap <= 0 || !status || !nseg || !qt || !hash || !tabs || (uint64_t)n * per_image >= ((uint64_t)1 << 32))
        return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > n) nthreads = n;
    PrepJob jobs[64];
    pthread_t th[64];
    int started[64];
    for (int t = 0; t < nthreads; ++t) {
        PrepJob jb = {paths, n, H, W, streams, per_image, (aq_jpeg_seg*)segs, seg_cap, coef_per_image, status, nseg, qt, hash, (aq_jpeg_gpu_tab*)tabs, t, nthreads};
        jobs[t] = jb;
        started[t] = pthread_create(&th[t], NULL, prep_thread, &jobs[t]) == 0;
        if (!started[t]) prep_thread(&jobs[t]);             /* no thread: do its share here */
    }
    for (int t = 0; t < nthreads; ++t)
        if (started[t]) pthread_join(th[t], NULL);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += status[i] != AQJ_OK;
    return bad;
}
