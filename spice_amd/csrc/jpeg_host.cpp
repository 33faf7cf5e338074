/* placeholder replaced by the real JPEG host path */
#include "runtime.h"
using namespace spicegpu;
extern "C" {
MJpegEncoder *mjpeg_encoder_new(int, uint64_t, MJpegEncoderRateControlCbs *, void *) { set_error("jpeg: not built yet"); return nullptr; }
void mjpeg_encoder_destroy(MJpegEncoder *) {}
uint8_t mjpeg_encoder_get_bytes_per_pixel(MJpegEncoder *) { return 0; }
int mjpeg_encoder_start_frame(MJpegEncoder *, int, int, int, uint8_t **, size_t *, uint32_t) { return -1; }
int mjpeg_encoder_encode_scanline(MJpegEncoder *, uint8_t *, size_t) { return 0; }
size_t mjpeg_encoder_end_frame(MJpegEncoder *) { return 0; }
uint64_t mjpeg_encoder_get_bit_rate(MJpegEncoder *) { return 0; }
void mjpeg_encoder_get_stats(MJpegEncoder *, MJpegEncoderStats *) {}
JpegEncoderContext *jpeg_encoder_create(JpegEncoderUsrContext *) { return nullptr; }
void jpeg_encoder_destroy(JpegEncoderContext *) {}
int jpeg_encode(JpegEncoderContext *, int, JpegEncoderImageType, int, int, uint8_t *, unsigned int, int, uint8_t *, unsigned int) { return 0; }
size_t spice_gpu_jpeg_out_bound(int, int) { return 0; }
int spice_gpu_jpeg_encode_batch(SpiceGpuJpegJob *, int) { return -1; }
int spice_gpu_jpeg_submit(SpiceGpuJpegJob *, int) { return -1; }
int spice_gpu_jpeg_finish(SpiceGpuJpegJob *, int) { return -1; }
}
