/* tools/kpeg_synth.c -- deterministic baseline-JPEG generator for tests and bench.
 *
 * Neither the reference's encoder (README.md:21-23: "not yet functional") nor
 * Pillow can be relied on at the GPU box, so the synthetic 4:4:4 inputs named in
 * BASELINE.json / SURVEY.md 8(d) are produced by this small encoder:
 *   SOI, APP0, DQT(id 0), DQT(id 1), SOF0 (3 comps, 1x1), 4x DHT (Annex K),
 *   [DRI], SOS, entropy data, EOI
 * i.e. exactly the marker set the reference decoder accepts (DRI
 * only when a restart interval is asked for -- the reference rejects it).
 *
 * Test infrastructure only; nothing in the product path links to it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- Annex K tables ------------------------------------------------------ */
static const uint8_t K_LUMA_Q[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                     14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                     18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                     49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t K_CHROMA_Q[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                       24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static const uint8_t DC_LUMA_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_CHROMA_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_LUMA_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_LUMA_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_CHROMA_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_CHROMA_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

static const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    uint16_t code[256];
    uint8_t len[256];
} hcode;

static void build_hcode(const uint8_t bits[16], const uint8_t* vals, hcode* h)
{
    memset(h, 0, sizeof(*h));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i) {
            h->code[vals[k]] = (uint16_t)code;
            h->len[vals[k]] = (uint8_t)l;
            code++;
            k++;
        }
        code <<= 1;
    }
}

/* ---- bit writer (un-stuffed; stuffing is applied when chunks are joined) --- */
typedef struct {
    uint8_t* p;
    size_t cap, n; /* whole bytes written */
    uint64_t acc;
    int nacc; /* bits held in acc */
} bitw;

static void bw_init(bitw* b, size_t cap)
{
    b->p = (uint8_t*)malloc(cap);
    b->cap = cap;
    b->n = 0;
    b->acc = 0;
    b->nacc = 0;
}

static inline void bw_put(bitw* b, uint32_t v, int nb)
{
    b->acc = (b->acc << nb) | (v & ((1u << nb) - 1u));
    b->nacc += nb;
    while (b->nacc >= 8) {
        if (b->n + 1 > b->cap) {
            b->cap = b->cap * 2 + 64;
            b->p = (uint8_t*)realloc(b->p, b->cap);
        }
        b->p[b->n++] = (uint8_t)(b->acc >> (b->nacc - 8));
        b->nacc -= 8;
    }
}

static inline int category(int v)
{
    int a = v < 0 ? -v : v, c = 0;
    while (a) {
        c++;
        a >>= 1;
    }
    return c;
}

static void encode_block(bitw* b, const int16_t* zz, int* pred, const hcode* dc, const hcode* ac)
{
    int diff = zz[0] - *pred;
    *pred = zz[0];
    int c = category(diff);
    bw_put(b, dc->code[c], dc->len[c]);
    if (c) bw_put(b, (uint32_t)(diff < 0 ? diff - 1 : diff), c);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        int v = zz[k];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            bw_put(b, ac->code[0xF0], ac->len[0xF0]);
            run -= 16;
        }
        c = category(v);
        int sym = (run << 4) | c;
        bw_put(b, ac->code[sym], ac->len[sym]);
        bw_put(b, (uint32_t)(v < 0 ? v - 1 : v), c);
        run = 0;
    }
    if (run) bw_put(b, ac->code[0], ac->len[0]);
}

/* ---- pixel source ---------------------------------------------------------- */
static inline uint64_t splitmix(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* approx. N(0,1): sum of four 16-bit uniforms, variance-normalised */
static inline double gauss4(uint64_t* s)
{
    uint64_t r = splitmix(s);
    double u = (double)(r & 0xFFFF) + (double)((r >> 16) & 0xFFFF) + (double)((r >> 32) & 0xFFFF) +
               (double)((r >> 48) & 0xFFFF);
    /* mean 2*65535, var 4*(65536^2-1)/12 */
    return (u - 131070.0) / 37837.2;
}

static inline uint8_t clampu8(double v)
{
    int i = (int)floor(v + 0.5);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

/* SURVEY.md 8(d) field: R=127+90 sin(x/97)+30 cos(y/53), G=127+80 sin((x+y)/131),
 * B=127+100 cos(x/71) sin(y/89), plus N(0,sigma) noise per channel.
 * mode 1: pure uniform noise (dense stress input). */
static void field_row(uint32_t w, uint32_t y, uint64_t seed, double sigma, int mode, uint8_t* rgb)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)y * 0xD1B54A32D192ED03ull + 0x1234567ull;
    (void)splitmix(&s);
    if (mode == 1) {
        for (uint32_t x = 0; x < w * 3; x += 1) rgb[x] = (uint8_t)(splitmix(&s) >> 56);
        return;
    }
    double cy = 30.0 * cos(y / 53.0), sy = sin(y / 89.0);
    for (uint32_t x = 0; x < w; ++x) {
        double r = 127.0 + 90.0 * sin(x / 97.0) + cy;
        double g = 127.0 + 80.0 * sin((x + y) / 131.0);
        double b = 127.0 + 100.0 * cos(x / 71.0) * sy;
        if (sigma > 0) {
            r += sigma * gauss4(&s);
            g += sigma * gauss4(&s);
            b += sigma * gauss4(&s);
        }
        rgb[x * 3 + 0] = clampu8(r);
        rgb[x * 3 + 1] = clampu8(g);
        rgb[x * 3 + 2] = clampu8(b);
    }
}

void kpeg_synth_field(uint32_t w, uint32_t h, uint64_t seed, double sigma, int mode, uint8_t* rgb)
{
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) field_row(w, (uint32_t)y, seed, sigma, mode, rgb + (size_t)y * w * 3);
}

/* ---- forward path: RGB -> YCbCr -> FDCT -> quantise ------------------------- */
static double g_c[8][8];
static void init_dct(void)
{
    static int done = 0;
    if (done) return;
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) g_c[u][x] = (u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * M_PI / 16.0);
    done = 1;
}

static void fdct_quant(const double px[64], const uint16_t* q /*natural order*/, int16_t* zz)
{
    double tmp[64], F[64];
    for (int r = 0; r < 8; ++r)
        for (int v = 0; v < 8; ++v) {
            double s = 0;
            for (int x = 0; x < 8; ++x) s += px[r * 8 + x] * g_c[v][x];
            tmp[r * 8 + v] = s;
        }
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            double s = 0;
            for (int r = 0; r < 8; ++r) s += tmp[r * 8 + v] * g_c[u][r];
            F[u * 8 + v] = s;
        }
    for (int k = 0; k < 64; ++k) {
        int n = ZZ[k];
        double v = F[n] / q[n];
        zz[k] = (int16_t)(v < 0 ? -floor(-v + 0.5) : floor(v + 0.5));
    }
}

static void make_qtables(int quality, uint16_t ql[64], uint16_t qc[64])
{
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        int a = (K_LUMA_Q[i] * scale + 50) / 100, b = (K_CHROMA_Q[i] * scale + 50) / 100;
        ql[i] = (uint16_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        qc[i] = (uint16_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
}

/* 8 rows of RGB -> one MCU row of quantised zig-zag coefficients [mw][3][64] */
static void mcu_row_coefs(const uint8_t* rows /*8*w*3*/, uint32_t w, const uint16_t* ql, const uint16_t* qc, int16_t* out)
{
    uint32_t mw = w / 8;
    for (uint32_t m = 0; m < mw; ++m) {
        double px[3][64];
        for (int r = 0; r < 8; ++r)
            for (int x = 0; x < 8; ++x) {
                const uint8_t* p = rows + ((size_t)r * w + m * 8 + x) * 3;
                double R = p[0], G = p[1], B = p[2];
                px[0][r * 8 + x] = 0.299 * R + 0.587 * G + 0.114 * B - 128.0;
                px[1][r * 8 + x] = -0.168735892 * R - 0.331264108 * G + 0.5 * B;
                px[2][r * 8 + x] = 0.5 * R - 0.418687589 * G - 0.081312411 * B;
            }
        fdct_quant(px[0], ql, out + ((size_t)m * 3 + 0) * 64);
        fdct_quant(px[1], qc, out + ((size_t)m * 3 + 1) * 64);
        fdct_quant(px[2], qc, out + ((size_t)m * 3 + 2) * 64);
    }
}

/* ---- container ---------------------------------------------------------------- */
typedef struct {
    uint8_t* p;
    size_t cap, n;
    int overflow;
} obuf;

static void ob_put(obuf* o, const void* src, size_t k)
{
    if (o->n + k > o->cap) {
        o->overflow = 1;
        return;
    }
    memcpy(o->p + o->n, src, k);
    o->n += k;
}
static void ob_u8(obuf* o, int v)
{
    uint8_t b = (uint8_t)v;
    ob_put(o, &b, 1);
}
static void ob_u16(obuf* o, int v)
{
    ob_u8(o, v >> 8);
    ob_u8(o, v & 255);
}

static void put_dht(obuf* o, int cls, int id, const uint8_t bits[16], const uint8_t* vals, int nvals)
{
    ob_u16(o, 0xFFC4);
    ob_u16(o, 2 + 1 + 16 + nvals);
    ob_u8(o, (cls << 4) | id);
    ob_put(o, bits, 16);
    ob_put(o, vals, (size_t)nvals);
}

static void put_headers(obuf* o, uint32_t w, uint32_t h, const uint16_t* ql, const uint16_t* qc, uint32_t dri)
{
    static const uint8_t app0[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    ob_u16(o, 0xFFD8);
    ob_put(o, app0, sizeof(app0));
    for (int t = 0; t < 2; ++t) {
        ob_u16(o, 0xFFDB);
        ob_u16(o, 67);
        ob_u8(o, t);
        for (int k = 0; k < 64; ++k) ob_u8(o, (t ? qc : ql)[ZZ[k]]);
    }
    ob_u16(o, 0xFFC0);
    ob_u16(o, 17);
    ob_u8(o, 8);
    ob_u16(o, (int)h);
    ob_u16(o, (int)w);
    ob_u8(o, 3);
    for (int c = 0; c < 3; ++c) {
        ob_u8(o, c + 1);
        ob_u8(o, 0x11);
        ob_u8(o, c ? 1 : 0);
    }
    put_dht(o, 0, 0, DC_LUMA_BITS, DC_VALS, 12);
    put_dht(o, 1, 0, AC_LUMA_BITS, AC_LUMA_VALS, 162);
    put_dht(o, 0, 1, DC_CHROMA_BITS, DC_VALS, 12);
    put_dht(o, 1, 1, AC_CHROMA_BITS, AC_CHROMA_VALS, 162);
    if (dri) {
        ob_u16(o, 0xFFDD);
        ob_u16(o, 4);
        ob_u16(o, (int)dri);
    }
    ob_u16(o, 0xFFDA);
    ob_u16(o, 12);
    ob_u8(o, 3);
    ob_u8(o, 1);
    ob_u8(o, 0x00);
    ob_u8(o, 2);
    ob_u8(o, 0x11);
    ob_u8(o, 3);
    ob_u8(o, 0x11);
    ob_u8(o, 0);
    ob_u8(o, 63);
    ob_u8(o, 0);
}

/* Encode quantised coefficients coef[nmcu][3][64] (zig-zag, absolute DC) into a
 * JFIF.  restart_interval in MCUs (0 = none).  Returns the file size, 0 if `cap`
 * is too small. */
size_t kpeg_synth_encode_coefs(const int16_t* coef, uint32_t w, uint32_t h, const uint16_t ql[64],
                               const uint16_t qc[64], uint32_t restart_interval, uint8_t* out, size_t cap)
{
    hcode hdc[2], hac[2];
    build_hcode(DC_LUMA_BITS, DC_VALS, &hdc[0]);
    build_hcode(DC_CHROMA_BITS, DC_VALS, &hdc[1]);
    build_hcode(AC_LUMA_BITS, AC_LUMA_VALS, &hac[0]);
    build_hcode(AC_CHROMA_BITS, AC_CHROMA_VALS, &hac[1]);

    uint32_t mw = w / 8, mh = h / 8, nmcu = mw * mh;
    uint32_t chunk = restart_interval ? restart_interval : mw;
    uint32_t nchunks = (nmcu + chunk - 1) / chunk;
    bitw* bws = (bitw*)calloc(nchunks, sizeof(bitw));

    /* chunks are independent once the predictor at their start is known */
#pragma omp parallel for schedule(dynamic, 4)
    for (long ci = 0; ci < (long)nchunks; ++ci) {
        uint32_t m0 = (uint32_t)ci * chunk, m1 = m0 + chunk > nmcu ? nmcu : m0 + chunk;
        bitw* b = &bws[ci];
        bw_init(b, (size_t)(m1 - m0) * 96 + 64);
        int pred[3] = {0, 0, 0};
        if (!restart_interval && m0 > 0)
            for (int c = 0; c < 3; ++c) pred[c] = coef[((size_t)(m0 - 1) * 3 + c) * 64];
        for (uint32_t m = m0; m < m1; ++m)
            for (int c = 0; c < 3; ++c)
                encode_block(b, coef + ((size_t)m * 3 + c) * 64, &pred[c], &hdc[c ? 1 : 0], &hac[c ? 1 : 0]);
    }

    obuf o = {out, cap, 0, 0};
    put_headers(&o, w, h, ql, qc, restart_interval);

    /* join the chunks: bit-concatenate (or byte-align + RSTn), stuffing FF -> FF 00 */
    uint64_t acc = 0;
    int nacc = 0;
#define EMIT_BYTE(v)                  \
    do {                              \
        uint8_t _b = (uint8_t)(v);    \
        ob_u8(&o, _b);                \
        if (_b == 0xFF) ob_u8(&o, 0); \
    } while (0)
    for (uint32_t ci = 0; ci < nchunks; ++ci) {
        bitw* b = &bws[ci];
        if (nacc == 0) {
            for (size_t i = 0; i < b->n; ++i) EMIT_BYTE(b->p[i]);
        } else {
            for (size_t i = 0; i < b->n; ++i) {
                acc = (acc << 8) | b->p[i];
                EMIT_BYTE(acc >> nacc);
                acc &= (1u << nacc) - 1u;
            }
        }
        if (b->nacc) {
            acc = (acc << b->nacc) | (b->acc & ((1ull << b->nacc) - 1));
            nacc += b->nacc;
            if (nacc >= 8) {
                EMIT_BYTE(acc >> (nacc - 8));
                nacc -= 8;
                acc &= (1u << nacc) - 1u;
            }
        }
        free(b->p);
        int last = ci + 1 == nchunks;
        if (restart_interval || last) {
            if (nacc) { /* pad with 1-bits */
                EMIT_BYTE((acc << (8 - nacc)) | ((1u << (8 - nacc)) - 1u));
                nacc = 0;
                acc = 0;
            }
            if (restart_interval && !last) {
                ob_u8(&o, 0xFF);
                ob_u8(&o, 0xD0 + (ci & 7));
            }
        }
    }
#undef EMIT_BYTE
    free(bws);
    ob_u16(&o, 0xFFD9);
    return o.overflow ? 0 : o.n;
}

/* RGB (h*w*3) -> quantised coefficients [nmcu][3][64]; also returns the tables. */
void kpeg_synth_rgb_to_coefs(const uint8_t* rgb, uint32_t w, uint32_t h, int quality, int16_t* coef,
                             uint16_t ql[64], uint16_t qc[64])
{
    init_dct();
    make_qtables(quality, ql, qc);
    uint32_t mw = w / 8, mh = h / 8;
#pragma omp parallel for schedule(static)
    for (long tr = 0; tr < (long)mh; ++tr)
        mcu_row_coefs(rgb + (size_t)tr * 8 * w * 3, w, ql, qc, coef + (size_t)tr * mw * 192);
}

/* RGB -> JFIF */
size_t kpeg_synth_encode_rgb(const uint8_t* rgb, uint32_t w, uint32_t h, int quality, uint32_t restart_interval,
                             uint8_t* out, size_t cap)
{
    if ((w & 7) || (h & 7) || !w || !h) return 0;
    uint16_t ql[64], qc[64];
    int16_t* coef = (int16_t*)malloc((size_t)(w / 8) * (h / 8) * 192 * sizeof(int16_t));
    kpeg_synth_rgb_to_coefs(rgb, w, h, quality, coef, ql, qc);
    size_t n = kpeg_synth_encode_coefs(coef, w, h, ql, qc, restart_interval, out, cap);
    free(coef);
    return n;
}

/* The SURVEY.md 8(d) synthetic image, generated one MCU row at a time so that a
 * 16384x16384 input never needs its 805 MB RGB source in memory.
 * mode 0: smooth field + N(0,sigma) noise; mode 1: uniform noise. */
size_t kpeg_synth_jpeg_rows(uint32_t w, uint32_t h, uint32_t y0, uint64_t seed, int quality, uint32_t restart_interval,
                            double sigma, int mode, uint8_t* out, size_t cap);

size_t kpeg_synth_jpeg(uint32_t w, uint32_t h, uint64_t seed, int quality, uint32_t restart_interval, double sigma,
                       int mode, uint8_t* out, size_t cap)
{
    return kpeg_synth_jpeg_rows(w, h, 0, seed, quality, restart_interval, sigma, mode, out, cap);
}

/* Same image function, but the h rows start at row y0 of the (virtual) full image: with a
 * restart interval of whole MCU rows the result is byte-for-byte the stripe of the full
 * image's entropy data (restart intervals are independent), apart from the RSTn numbering. */
size_t kpeg_synth_jpeg_rows(uint32_t w, uint32_t h, uint32_t y0, uint64_t seed, int quality, uint32_t restart_interval,
                            double sigma, int mode, uint8_t* out, size_t cap)
{
    if ((w & 7) || (h & 7) || !w || !h) return 0;
    init_dct();
    uint16_t ql[64], qc[64];
    make_qtables(quality, ql, qc);
    uint32_t mw = w / 8, mh = h / 8;
    int16_t* coef = (int16_t*)malloc((size_t)mw * mh * 192 * sizeof(int16_t));
#pragma omp parallel
    {
        uint8_t* rows = (uint8_t*)malloc((size_t)8 * w * 3);
#pragma omp for schedule(static)
        for (long tr = 0; tr < (long)mh; ++tr) {
            for (int r = 0; r < 8; ++r) field_row(w, y0 + (uint32_t)tr * 8 + r, seed, sigma, mode, rows + (size_t)r * w * 3);
            mcu_row_coefs(rows, w, ql, qc, coef + (size_t)tr * mw * 192);
        }
        free(rows);
    }
    size_t n = kpeg_synth_encode_coefs(coef, w, h, ql, qc, restart_interval, out, cap);
    free(coef);
    return n;
}
