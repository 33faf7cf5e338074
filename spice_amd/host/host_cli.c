/*
 * Drives red_compress_host.c from a script file, for the parity tests.
 *
 * input : "HOST1" i32 image_compression, u32 window_px, i32 enable_jpeg, i32 jpeg_quality,
 *         i32 zlib_wrap, i32 zlib_level, u32 nbitmaps, then per bitmap:
 *         i32 format, top_down, unstable, can_lossy; u32 x, y, stride, nchunks; u32 rows[nchunks];
 *         the chunks' bytes
 * output: per bitmap i32 ok, type; u32 size, glz_data_size, jpeg_size, nfreed; u64 freed[nfreed];
 *         the compressed bytes
 * All integers little-endian, as the host writes them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "red_compress_host.h"

static int rd(FILE *f, void *p, size_t n)
{
    return fread(p, 1, n, f) == n;
}

int main(int argc, char **argv)
{
    FILE *in, *out;
    char magic[5];
    int32_t cfg[6];
    uint32_t window, nbitmaps, b;
    HostDisplayClient *dcc;

    if (argc != 3) {
        fprintf(stderr, "usage: %s script.bin result.bin\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    out = fopen(argv[2], "wb");
    if (!in || !out || !rd(in, magic, 5) || memcmp(magic, "HOST1", 5) || !rd(in, &cfg[0], 4) ||
        !rd(in, &window, 4) || !rd(in, &cfg[1], 16) || !rd(in, &nbitmaps, 4)) {
        fprintf(stderr, "bad script\n");
        return 2;
    }
    if (spice_gpu_init(0) != 0) {
        fprintf(stderr, "spice_gpu_init: %s\n", spice_gpu_last_error());
        return 3;
    }
    dcc = host_display_client_new(cfg[0], window, cfg[1], cfg[2], cfg[3], cfg[4]);
    if (!dcc) {
        fprintf(stderr, "client: %s\n", spice_gpu_last_error());
        return 3;
    }
    for (b = 0; b < nbitmaps; b++) {
        int32_t h4[4];
        uint32_t g4[4], c, size = 0, nfreed;
        int32_t ok;
        uint32_t *rows;
        HostBitmap bm;
        HostCompressed res;
        uintptr_t freed[256];
        uint64_t freed64;
        uint8_t *bytes = NULL;
        int k;

        if (!rd(in, h4, 16) || !rd(in, g4, 16)) {
            fprintf(stderr, "truncated script\n");
            return 2;
        }
        memset(&bm, 0, sizeof(bm));
        bm.format = h4[0];
        bm.top_down = h4[1];
        bm.unstable = h4[2];
        bm.x = g4[0];
        bm.y = g4[1];
        bm.stride = g4[2];
        bm.num_chunks = g4[3];
        rows = (uint32_t *)malloc(sizeof(uint32_t) * bm.num_chunks);
        bm.chunk = (HostChunk *)calloc(bm.num_chunks, sizeof(HostChunk));
        if (!rd(in, rows, sizeof(uint32_t) * bm.num_chunks)) {
            return 2;
        }
        for (c = 0; c < bm.num_chunks; c++) {
            bm.chunk[c].len = rows[c] * bm.stride;
            bm.chunk[c].data = (uint8_t *)malloc(bm.chunk[c].len ? bm.chunk[c].len : 1);
            if (!rd(in, bm.chunk[c].data, bm.chunk[c].len)) {
                return 2;
            }
        }
        ok = host_compress_image(dcc, &bm, h4[3], BITMAP_GRADUAL_INVALID, (uintptr_t)(1000 + b), &res);
        if (ok == 1 && res.comp_buf) {
            bytes = (uint8_t *)malloc(res.comp_buf_size);
            size = (uint32_t)host_gather_compressed(&res, bytes, res.comp_buf_size);
        }
        nfreed = (uint32_t)host_pop_freed_images(dcc, freed, 256);
        fwrite(&ok, 4, 1, out);
        fwrite(&res.type, 4, 1, out);
        fwrite(&size, 4, 1, out);
        fwrite(&res.glz_data_size, 4, 1, out);
        fwrite(&res.jpeg_size, 4, 1, out);
        fwrite(&nfreed, 4, 1, out);
        for (k = 0; k < (int)nfreed; k++) {
            freed64 = (uint64_t)freed[k];
            fwrite(&freed64, 8, 1, out);
        }
        if (size) {
            fwrite(bytes, 1, size, out);
        }
        free(bytes);
        host_reset_compress_bufs(dcc); /* the image has been "sent" */
        for (c = 0; c < bm.num_chunks; c++) {
            free(bm.chunk[c].data);
        }
        free(bm.chunk);
        free(rows);
    }
    host_display_client_free(dcc);
    fclose(out);
    fclose(in);
    return 0;
}
