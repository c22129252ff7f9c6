/* Fuzz harness for the host entropy decoder (aquaculture_amd/csrc/jpeg_coef.c), built by tests/test_jpeg.py with
 * -fsanitize=address,undefined: mutates valid baseline JPEGs (byte flips, truncations, marker-length edits, spliced garbage) and decodes
 * them; any out-of-bounds access or undefined shift aborts the process.  usage: fuzz_jpeg <iterations> <file.jpg>... */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/aq_jpeg.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int iters = atoi(argv[1]);
    int ok = 0, bad = 0;
    for (int f = 2; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) return 2;
        fseek(fp, 0, SEEK_END);
        const long n = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        uint8_t* orig = (uint8_t*)malloc((size_t)n);
        if (fread(orig, 1, (size_t)n, fp) != (size_t)n) return 2;
        fclose(fp);
        aq_jpeg_info info;
        if (aq_jpeg_scan(orig, (size_t)n, &info) != AQJ_OK) return 3;
        const size_t cap = (size_t)info.total_blocks * 64;          /* EXACT capacity: one value too many written = a heap overflow ASan sees */
        for (int it = 0; it < iters; ++it) {
            size_t m = (size_t)n;
            uint8_t* buf = (uint8_t*)malloc(m + 64);
            memcpy(buf, orig, m);
            const int kind = (int)(rnd() % 6);
            if (kind == 0) {                                        /* flip a few bytes anywhere */
                for (int k = 0, c = 1 + (int)(rnd() % 8); k < c; ++k) buf[rnd() % m] ^= (uint8_t)(1u << (rnd() % 8));
            } else if (kind == 1) {                                 /* truncate */
                m = 2 + rnd() % (m - 2);
            } else if (kind == 2) {                                 /* damage the headers (first 700 bytes: tables, lengths, sampling factors) */
                for (int k = 0, c = 1 + (int)(rnd() % 4); k < c; ++k) buf[rnd() % (m < 700 ? m : 700)] = (uint8_t)rnd();
            } else if (kind == 3) {                                 /* garbage in the entropy-coded data, incl. stray markers */
                const size_t at = 700 + rnd() % (m > 800 ? m - 750 : 1);
                for (size_t k = at; k < at + 1 + rnd() % 40 && k < m; ++k) buf[k] = (rnd() & 3) ? (uint8_t)rnd() : 0xFF;
            } else if (kind == 4) {                                 /* rewrite a DHT's sixteen counts, keeping their SUM (so the segment length stays
                                                                       consistent and the table builder is reached with over- or under-subscribed
                                                                       lengths; ADVICE r03: random flips never got past the length check) */
                size_t at = 2;
                int nth = (int)(rnd() % 4);
                while (at + 4 < m) {
                    if (buf[at] != 0xFF) break;
                    const unsigned mk = buf[at + 1], ln = ((unsigned)buf[at + 2] << 8) | buf[at + 3];
                    if (mk == 0xDA) break;
                    if (mk == 0xC4 && ln >= 19 && at + 2 + ln <= m && nth-- <= 0) {
                        uint8_t* c = buf + at + 5;
                        int sum = 0;
                        for (int k = 0; k < 16; ++k) sum += c[k];
                        memset(c, 0, 16);
                        while (sum > 0) {                           /* pile the codes onto a few (mostly short) lengths */
                            const int k = (rnd() & 3) ? (int)(rnd() % 4) : (int)(rnd() % 16);
                            const int take = 1 + (int)(rnd() % (unsigned)sum);
                            const int put = c[k] + take > 255 ? 255 - c[k] : take;
                            c[k] = (uint8_t)(c[k] + put);
                            sum -= put;
                        }
                        break;
                    }
                    at += 2 + ln;
                }
            } else {                                                /* all ones / all zeros tail */
                const size_t at = 2 + rnd() % (m - 2);
                memset(buf + at, (rnd() & 1) ? 0xFF : 0x00, m - at);
            }
            uint8_t* exact = (uint8_t*)malloc(m);                   /* exact-size input buffer: a read past the end is caught too */
            memcpy(exact, buf, m);
            int16_t* coef = (int16_t*)malloc(cap * sizeof(int16_t));
            aq_jpeg_info inf2;
            const int rc = aq_jpeg_decode_coeffs(exact, m, coef, cap, &inf2);
            if (rc == AQJ_OK) ++ok; else ++bad;
            const int rs = aq_jpeg_scan(exact, m, &inf2);
            (void)rs;
            {   /* the GPU path's host preparation on the same bytes: exactly-sized output buffers of varying capacity */
                static aq_jpeg_stream_info si;
                const size_t scap = 64 + rnd() % (m + 64);
                uint8_t* sbuf = (uint8_t*)malloc(scap);
                const int segcap = 1 + (int)(rnd() % 40);
                uint32_t* so = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)segcap);
                uint32_t* sl = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)segcap);
                const int rp = aq_jpeg_prepare(exact, m, sbuf, scap, so, sl, segcap, &si);
                if (rp == AQJ_OK) {
                    if (si.nseg < 1 || si.nseg > segcap || si.stream_bytes > scap) abort();
                    for (int k = 0; k < si.nseg; ++k)
                        if ((so[k] & 15) || (size_t)so[k] + sl[k] + 8 > scap) abort();
                }
                free(sbuf); free(so); free(sl);
            }
            free(coef); free(exact); free(buf);
        }
        free(orig);
    }
    printf("fuzz_jpeg: %d decoded, %d rejected\n", ok, bad);
    return 0;
}
