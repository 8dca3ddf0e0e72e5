/* liblbdrn_jp2.so -- the MSB-plane payload as a REAL JPEG 2000 stream.
 *
 * The reference stores the high-bit planes by shelling out to GDAL's JP2OpenJPEG driver:
 *     gdal_translate -of JP2OpenJPEG -co QUALITY=100 -co REVERSIBLE=YES base.tif base.jp2     (encode.py:137)
 *     gdal_translate -of GTiff base.jp2 recon.tif                                             (decode.py:69-73)
 * i.e. a multi-component, reversible (5/3 wavelet), single-layer lossless JP2 written by OpenJPEG.  GDAL is absent from
 * this image; the OpenJPEG library it wraps is not (libopenjp2 2.4, /opt/conda).  This shim binds it through memory
 * streams: host code only, plain C ABI, no GPU call -- the GPU codec of the package (LBB2, lbdrn_hip.h) stays the default
 * payload; this one is selected with LBDRN_BASE_CODEC=jp2 and recognised on decode by the JP2 signature box (or the raw
 * codestream marker), so that a base payload written by the reference decodes too.
 * Parity: lossless by construction (pixel values pinned); byte identity with a GDAL-written file is UNPINNED (GDAL's
 * tiling / box choices are not reproducible here without GDAL).
 *
 * Return value: 0 ok, negative = error (message via lbdrn_jp2_last_error()). */
#ifndef LBDRN_JP2_H
#define LBDRN_JP2_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

const char *lbdrn_jp2_last_error(void);

/* planes: [C][H][W] uint16 host array; bits: 8 or 16 = the precision written into the stream (8 when every value is
 * <= 255: the reference's MSB raster is Byte in that case, LBDRNdataset.py:100).  On success *out is a malloc'ed
 * buffer of *out_bytes bytes holding a complete .jp2 file (release with lbdrn_jp2_free). */
int lbdrn_jp2_encode(const uint16_t *planes, int32_t C, int32_t H, int32_t W, int32_t bits, uint8_t **out, size_t *out_bytes);

/* Geometry of a .jp2 file / raw codestream held in memory (header only). */
int lbdrn_jp2_info(const uint8_t *buf, size_t bytes, int32_t *C, int32_t *H, int32_t *W, int32_t *bits);

/* Decode into a caller-owned [C][H][W] uint16 array of the geometry lbdrn_jp2_info reported. */
int lbdrn_jp2_decode(const uint8_t *buf, size_t bytes, uint16_t *planes, int32_t C, int32_t H, int32_t W);

void lbdrn_jp2_free(uint8_t *p);

/* Worker threads OpenJPEG may use inside ONE encode / decode call from now on (process-wide; 0 or 1: none, the
 * library's default; at most 256).  The code blocks of a tile are coded independently, so the stream's bytes and the
 * decoded values do not depend on it.  Returns the previous setting. */
int lbdrn_jp2_set_threads(int32_t n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
