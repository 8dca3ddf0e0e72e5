/* liblbdrn_jp2.so: OpenJPEG (libopenjp2) behind the C ABI of include/lbdrn_jp2.h -- reversible multi-component JP2 in
 * memory, standing where the reference runs gdal_translate -of JP2OpenJPEG -co QUALITY=100 -co REVERSIBLE=YES
 * (encode.py:137; decode.py:69-73).  Host code; built by csrc/build.py when openjpeg.h is found. */
#include <openjpeg.h>
#include <stdarg.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lbdrn_jp2.h"

static __thread char g_err[512];
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *lbdrn_jp2_last_error(void) { return g_err; }
/* worker threads OpenJPEG may use inside one call (code blocks are independent: the bytes do not depend on it); read by
 * calls on any thread while another thread may set it */
static atomic_int g_threads = 0;
int lbdrn_jp2_set_threads(int32_t n)
{
    if (n >= 0 && n <= 256) return atomic_exchange(&g_threads, (int)n);
    return atomic_load(&g_threads);
}

/* OpenJPEG reports an error on whichever thread met it -- with worker threads, a code-block worker whose thread-local
 * g_err nobody reads.  Every call therefore hands the handler a context of its own: the FIRST message of
 * the call is kept (first writer wins, an atomic flag), and the calling thread copies it into its g_err when the call
 * ends (take_err). */
typedef struct { atomic_flag taken; char msg[480]; } err_ctx;
static void on_error(const char *msg, void *u)
{
    err_ctx *e = (err_ctx *)u;
    if (e && !atomic_flag_test_and_set(&e->taken)) {
        snprintf(e->msg, sizeof e->msg, "%s", msg ? msg : "");
        for (size_t k = strlen(e->msg); k > 0 && (e->msg[k - 1] == '\n' || e->msg[k - 1] == '\r'); --k) e->msg[k - 1] = 0;
    }
}
static void take_err(err_ctx *e, const char *fallback)   /* on the calling thread, after the codec's workers have been joined */
{
    if (e->msg[0]) set_err("openjpeg: %s", e->msg);
    else if (!g_err[0]) set_err("%s", fallback);
}
static void on_quiet(const char *msg, void *u) { (void)msg; (void)u; }

/* ---- a growable / read-only memory stream */
typedef struct { uint8_t *data; size_t size, cap, pos; int writable; } mem_t;

static OPJ_SIZE_T mem_read(void *dst, OPJ_SIZE_T n, void *u)
{
    mem_t *m = (mem_t *)u;
    if (m->pos >= m->size) return (OPJ_SIZE_T)-1;
    if (n > m->size - m->pos) n = m->size - m->pos;
    memcpy(dst, m->data + m->pos, n);
    m->pos += n;
    return n;
}
static OPJ_SIZE_T mem_write(void *src, OPJ_SIZE_T n, void *u)
{
    mem_t *m = (mem_t *)u;
    if (m->pos + n > m->cap) {
        size_t cap = m->cap ? m->cap : (1u << 20);
        while (cap < m->pos + n) cap *= 2;
        uint8_t *p = (uint8_t *)realloc(m->data, cap);
        if (!p) return (OPJ_SIZE_T)-1;
        m->data = p;
        m->cap = cap;
    }
    memcpy(m->data + m->pos, src, n);
    m->pos += n;
    if (m->pos > m->size) m->size = m->pos;
    return n;
}
static OPJ_OFF_T mem_skip(OPJ_OFF_T n, void *u)
{
    mem_t *m = (mem_t *)u;
    if (n < 0) return -1;
    if (m->writable) {   /* skipping forward while writing leaves a hole that is filled in later (box lengths) */
        const size_t end = m->pos + (size_t)n;
        if (end > m->size) {   /* whatever of [pos, end) lies beyond the written bytes becomes zeros, never heap contents */
            static const uint8_t zero[256] = {0};
            m->pos = m->size;
            while (m->pos < end) {
                const size_t k = end - m->pos > 256 ? 256 : end - m->pos;
                if (mem_write((void *)zero, k, u) == (OPJ_SIZE_T)-1) return -1;
            }
        }
        m->pos = end;
        return n;
    }
    if ((size_t)n > m->size - m->pos) n = (OPJ_OFF_T)(m->size - m->pos);
    m->pos += (size_t)n;
    return n;
}
static OPJ_BOOL mem_seek(OPJ_OFF_T p, void *u)
{
    mem_t *m = (mem_t *)u;
    if (p < 0 || (!m->writable && (size_t)p > m->size)) return OPJ_FALSE;
    if (m->writable && (size_t)p > m->size) return OPJ_FALSE;
    m->pos = (size_t)p;
    return OPJ_TRUE;
}
static opj_stream_t *open_stream(mem_t *m, int input)
{
    opj_stream_t *s = opj_stream_create(1 << 20, input ? OPJ_TRUE : OPJ_FALSE);
    if (!s) return NULL;
    opj_stream_set_user_data(s, m, NULL);
    opj_stream_set_user_data_length(s, input ? m->size : 0);
    opj_stream_set_read_function(s, mem_read);
    opj_stream_set_write_function(s, mem_write);
    opj_stream_set_skip_function(s, mem_skip);
    opj_stream_set_seek_function(s, mem_seek);
    return s;
}

int lbdrn_jp2_encode(const uint16_t *planes, int32_t C, int32_t H, int32_t W, int32_t bits, uint8_t **out, size_t *out_bytes)
{
    if (!planes || !out || !out_bytes || C < 1 || C > 16384 || H < 1 || W < 1 || (bits != 8 && bits != 16)) {
        set_err("lbdrn_jp2_encode: bad argument");
        return -1;
    }
    *out = NULL;
    *out_bytes = 0;
    opj_cparameters_t prm;
    opj_set_default_encoder_parameters(&prm);
    prm.irreversible = 0;          /* REVERSIBLE=YES: the 5/3 integer wavelet */
    prm.tcp_numlayers = 1;
    prm.tcp_rates[0] = 0;          /* QUALITY=100: one layer, everything in it */
    prm.cp_disto_alloc = 1;
    prm.tcp_mct = 0;               /* independent bands: no colour transform */
    prm.numresolution = 6;
    {   /* fewer resolutions for small rasters (the smallest level must keep at least one sample) */
        int m = H < W ? H : W, r = 1;
        while ((m >> r) > 0 && r < 6) ++r;
        prm.numresolution = r;
    }
    if (H > 1024 || W > 1024) {    /* GDAL's JP2OpenJPEG driver tiles at 1024 x 1024 by default */
        prm.tile_size_on = OPJ_TRUE;
        prm.cp_tdx = 1024;
        prm.cp_tdy = 1024;
    }
    opj_image_cmptparm_t *cp = (opj_image_cmptparm_t *)calloc((size_t)C, sizeof *cp);
    if (!cp) { set_err("out of memory"); return -2; }
    for (int c = 0; c < C; ++c) {
        cp[c].dx = cp[c].dy = 1;
        cp[c].w = (OPJ_UINT32)W;
        cp[c].h = (OPJ_UINT32)H;
        cp[c].prec = cp[c].bpp = (OPJ_UINT32)bits;
        cp[c].sgnd = 0;
    }
    opj_image_t *img = opj_image_create((OPJ_UINT32)C, cp, C == 3 ? OPJ_CLRSPC_SRGB : (C == 1 ? OPJ_CLRSPC_GRAY : OPJ_CLRSPC_UNSPECIFIED));
    free(cp);
    if (!img) { set_err("opj_image_create failed"); return -2; }
    img->x0 = img->y0 = 0;
    img->x1 = (OPJ_UINT32)W;
    img->y1 = (OPJ_UINT32)H;
    const size_t n = (size_t)H * W;
    const unsigned vmax = bits == 8 ? 255u : 65535u;
    for (int c = 0; c < C; ++c) {
        OPJ_INT32 *d = img->comps[c].data;
        const uint16_t *s = planes + (size_t)c * n;
        for (size_t k = 0; k < n; ++k) {
            if (s[k] > vmax) { opj_image_destroy(img); set_err("value %u does not fit %d bits", (unsigned)s[k], bits); return -1; }
            d[k] = (OPJ_INT32)s[k];
        }
    }
    int rc = -3;
    mem_t mem = {NULL, 0, 0, 0, 1};
    err_ctx ectx = {ATOMIC_FLAG_INIT, {0}};
    const int threads = atomic_load(&g_threads);
    opj_codec_t *codec = opj_create_compress(OPJ_CODEC_JP2);
    opj_stream_t *st = NULL;
    g_err[0] = 0;
    if (!codec) { set_err("opj_create_compress failed"); goto done; }
    opj_set_error_handler(codec, on_error, &ectx);
    opj_set_warning_handler(codec, on_quiet, NULL);
    opj_set_info_handler(codec, on_quiet, NULL);
    if (!opj_setup_encoder(codec, &prm, img)) { take_err(&ectx, "opj_setup_encoder failed"); goto done; }
    if (threads > 1 && opj_has_thread_support()) (void)opj_codec_set_threads(codec, threads);   /* (between setup and start) */
    st = open_stream(&mem, 0);
    if (!st) { set_err("opj_stream_create failed"); goto done; }
    if (!opj_start_compress(codec, img, st) || !opj_encode(codec, st) || !opj_end_compress(codec, st)) {
        rc = -4;   /* (the message is taken below, once the codec -- and with it its worker threads -- is gone) */
        goto done;
    }
    rc = 0;
done:
    if (st) opj_stream_destroy(st);
    if (codec) opj_destroy_codec(codec);
    if (rc == -4) { take_err(&ectx, "openjpeg: compression failed"); rc = -3; }
    opj_image_destroy(img);
    if (rc) { free(mem.data); return rc; }
    *out = mem.data;
    *out_bytes = mem.size;
    return 0;
}

static OPJ_CODEC_FORMAT sniff(const uint8_t *buf, size_t bytes)
{
    static const uint8_t jp2[12] = {0, 0, 0, 12, 'j', 'P', ' ', ' ', 13, 10, 0x87, 10};
    if (bytes >= 12 && !memcmp(buf, jp2, 12)) return OPJ_CODEC_JP2;
    if (bytes >= 4 && buf[0] == 0xFF && buf[1] == 0x4F && buf[2] == 0xFF && buf[3] == 0x51) return OPJ_CODEC_J2K;
    return OPJ_CODEC_UNKNOWN;
}

/* decode: header only (planes == NULL) or everything */
static int decode_impl(const uint8_t *buf, size_t bytes, uint16_t *planes, int32_t *C, int32_t *H, int32_t *W, int32_t *bits)
{
    const OPJ_CODEC_FORMAT fmt = sniff(buf, bytes);
    if (fmt == OPJ_CODEC_UNKNOWN) { set_err("not a JPEG 2000 stream (no JP2 signature box, no SOC marker)"); return -1; }
    mem_t mem = {(uint8_t *)buf, bytes, bytes, 0, 0};
    opj_dparameters_t prm;
    opj_set_default_decoder_parameters(&prm);
    opj_codec_t *codec = opj_create_decompress(fmt);
    opj_stream_t *st = NULL;
    opj_image_t *img = NULL;
    int rc = -3;
    err_ctx ectx = {ATOMIC_FLAG_INIT, {0}};
    const int threads = atomic_load(&g_threads);
    const char *fallback = NULL;   /* rc == -3 with a fallback: an OpenJPEG call failed; its message is taken behind `done` */
    if (!codec) { set_err("opj_create_decompress failed"); return -3; }
    opj_set_error_handler(codec, on_error, &ectx);
    opj_set_warning_handler(codec, on_quiet, NULL);
    opj_set_info_handler(codec, on_quiet, NULL);
    g_err[0] = 0;
    if (!opj_setup_decoder(codec, &prm)) { fallback = "opj_setup_decoder failed"; goto done; }
    if (threads > 1 && opj_has_thread_support()) (void)opj_codec_set_threads(codec, threads);
    st = open_stream(&mem, 1);
    if (!st) { set_err("opj_stream_create failed"); goto done; }
    if (!opj_read_header(st, codec, &img) || !img) { fallback = "openjpeg: cannot read the header"; goto done; }
    {
        const int32_t c = (int32_t)img->numcomps, w = (int32_t)(img->x1 - img->x0), h = (int32_t)(img->y1 - img->y0);
        int32_t prec = 0;
        for (int k = 0; k < c; ++k) {
            const opj_image_comp_t *q = &img->comps[k];
            if (q->dx != 1 || q->dy != 1 || q->sgnd || q->prec > 16) { set_err("component %d: sub-sampled, signed or deeper than 16 bits", k); goto done; }
            if ((int32_t)q->prec > prec) prec = (int32_t)q->prec;
        }
        if (!planes) {
            *C = c; *H = h; *W = w; *bits = prec;
            rc = 0;
            goto done;
        }
        if (c != *C || h != *H || w != *W) { set_err("stream is %d x %d x %d, caller expected %d x %d x %d", c, h, w, *C, *H, *W); rc = -1; goto done; }
        if (!opj_decode(codec, st, img) || !opj_end_decompress(codec, st)) { fallback = "openjpeg: decoding failed"; goto done; }
        const size_t n = (size_t)h * w;
        for (int k = 0; k < c; ++k) {
            const OPJ_INT32 *d = img->comps[k].data;
            if (!d || (int32_t)img->comps[k].w != w || (int32_t)img->comps[k].h != h) { set_err("component %d came back empty or resized", k); goto done; }
            uint16_t *o = planes + (size_t)k * n;
            for (size_t e = 0; e < n; ++e) o[e] = (uint16_t)d[e];
        }
        rc = 0;
    }
done:
    if (img) opj_image_destroy(img);
    if (st) opj_stream_destroy(st);
    opj_destroy_codec(codec);          /* joins the codec's worker threads: whatever they reported is in ectx by now */
    if (fallback) take_err(&ectx, fallback);
    return rc;
}

int lbdrn_jp2_info(const uint8_t *buf, size_t bytes, int32_t *C, int32_t *H, int32_t *W, int32_t *bits)
{
    if (!buf || !C || !H || !W || !bits) { set_err("lbdrn_jp2_info: bad argument"); return -1; }
    return decode_impl(buf, bytes, NULL, C, H, W, bits);
}

int lbdrn_jp2_decode(const uint8_t *buf, size_t bytes, uint16_t *planes, int32_t C, int32_t H, int32_t W)
{
    if (!buf || !planes) { set_err("lbdrn_jp2_decode: bad argument"); return -1; }
    int32_t bits = 0;
    return decode_impl(buf, bytes, planes, &C, &H, &W, &bits);
}

void lbdrn_jp2_free(uint8_t *p) { free(p); }
