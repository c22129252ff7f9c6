/* libaqjpeg.so -- host half of the split JPEG decode (aquaculture_amd/csrc/jpeg_coef.c; plain C, no GPU dependency).
 *
 * Replaces, for the tile sweep of reference README.md:77, the entropy-decoding stage of the libjpeg(-turbo) call inside
 * [UPSTREAM utils/dataloaders.py LoadImages.__next__ -> cv2.imread]; the rest of that call (IDCT, upsampling, colour conversion) runs on
 * the GPU: aq_jpeg_idct_rgb in include/aq_engine.h.  Input files: what reference src/load_data/tile_tifs.py:66-74 writes (GDAL JPEG
 * driver: baseline, 8-bit, YCbCr 4:2:0). */
#ifndef AQ_JPEG_H
#define AQ_JPEG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AQJ_OK 0
#define AQJ_UNSUPPORTED (-1)   /* progressive / arithmetic / not 4:2:0 / not 8-bit: use the full software decoder */
#define AQJ_CORRUPT (-2)
#define AQJ_SPACE (-3)         /* coef_out too small: info->total_blocks * 64 values are needed */

typedef struct aq_jpeg_info {
    int32_t width, height, ncomp;
    int32_t mcu_cols, mcu_rows;           /* 16 x 16 MCUs (three components) */
    int32_t y_blocks_w, y_blocks_h;
    int32_t total_blocks;
    uint16_t qt[3][64];                   /* quantisation tables per component, natural (row-major) order */
} aq_jpeg_info;

/* Quantised DCT coefficients of one file, int16, natural order, 64 per block: Y blocks [2 mcu_rows][2 mcu_cols], then Cb [mcu_rows][mcu_cols],
 * then Cr.  cap: capacity of coef_out in int16 values. */
int aq_jpeg_decode_coeffs(const uint8_t* data, size_t n, int16_t* coef_out, size_t cap, aq_jpeg_info* info);
/* Headers only: fills info (sizes, tables). */
int aq_jpeg_scan(const uint8_t* data, size_t n, aq_jpeg_info* info);


/* ---- GPU entropy decode (round 4): the host parses the headers and strips byte stuffing / restart markers while copying the scan into
 * the upload buffer; the Huffman decoding runs on the device (aq_jpeg_huffman_decode, include/aq_engine.h), one lane per restart segment.
 * Same refusals as aq_jpeg_decode_coeffs (AQJ_UNSUPPORTED: use the software decoder; AQJ_CORRUPT: truncated scan, stray marker, missing EOI). */
typedef struct aq_jpeg_gpu_tab {
    uint16_t look[512];                   /* (length << 8) | symbol for codes of <= 9 bits, 0 = longer */
    int32_t maxcode[18];                  /* largest code of each length 1..16 (-1: none), [17] = sentinel */
    int32_t valoff[18];                   /* vals index of the first code of a length minus that code */
    uint8_t vals[256];
} aq_jpeg_gpu_tab;

typedef struct aq_jpeg_stream_info {
    int32_t width, height, mcu_cols, mcu_rows;
    int32_t restart;                      /* MCUs per restart interval (0: none) = MCUs per segment */
    int32_t nseg;                         /* segments written (AQJ_SPACE: segments needed) */
    uint32_t stream_bytes;                /* bytes used in stream_out */
    uint32_t pad;
    uint64_t tab_hash;                    /* equal hashes = equal Huffman tables: callers keep one device copy per distinct hash */
    uint16_t qt[3][64];                   /* quantisation tables per component, natural order */
    aq_jpeg_gpu_tab tabs[6];              /* [component][dc, ac] */
} aq_jpeg_stream_info;

/* stream_out (capacity cap): the scan's bytes without stuffed zeros, segment i at seg_off[i] (a multiple of 16), seg_len[i] bytes long,
 * followed by >= 8 zero bytes.  seg_cap: capacity of seg_off / seg_len. */
int aq_jpeg_prepare(const uint8_t* data, size_t n, uint8_t* stream_out, size_t cap, uint32_t* seg_off, uint32_t* seg_len, int seg_cap,
                    aq_jpeg_stream_info* si);

/* A super-batch without the interpreter in the loop: reads n files and prepares each into slot i of the upload buffer (per_image bytes at
 * streams + i * per_image) on nthreads POSIX threads.  segs: 32-byte descriptors as aq_jpeg_huffman_decode reads them, seg_cap per image
 * (nseg[i] used; tabset left 0); status[i] = AQJ_* (another size than W x H: AQJ_UNSUPPORTED with nseg[i] = -1); qt uint16 [n][3][64];
 * hash[i], tabs [n][6] aq_jpeg_gpu_tab (keep one per distinct hash).  Returns how many files are not AQJ_OK (-1: bad argument). */
int aq_jpeg_prepare_files(const char* const* paths, int n, int H, int W, uint8_t* streams, size_t per_image, void* segs, int seg_cap,
                          uint64_t coef_per_image, int32_t* status, int32_t* nseg, uint16_t* qt, uint64_t* hash, void* tabs, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
