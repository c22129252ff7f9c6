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

#ifdef __cplusplus
}
#endif
#endif
