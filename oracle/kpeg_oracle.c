/* oracle/kpeg_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of libKPEG's decode-to-PPM path (see kpeg_oracle.h).  It is
 * the arbiter for the HIP path on the GPU box, where /root/reference does not
 * exist.  It is pinned against the real reference (oracle/_ref/kpeg_ref) in
 * this container by tests/test_oracle.py and tests/golden/make_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math): every
 * float/double operation below must be a single IEEE operation in the order
 * written, because the reference's pixels depend on that order.
 */
#include "kpeg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Marker parser                                                       */
/* ------------------------------------------------------------------ */

typedef struct {
    const uint8_t* p;
    size_t n, pos;
    int eof; /* a read ran past the end (ifstream failbit)              */
} rd_t;

/* `m_imageFile >> std::noskipws >> byte`: on failure the variable keeps its
 * old value and the stream stays failed. */
static int rd_byte(rd_t* r, uint8_t* b)
{
    if (r->eof || r->pos >= r->n) {
        r->eof = 1;
        return 0;
    }
    *b = r->p[r->pos++];
    return 1;
}

/* m_imageFile.read(&u16, 2); htons() */
static uint16_t rd_be16(rd_t* r)
{
    uint8_t a = 0, b = 0;
    rd_byte(r, &a);
    rd_byte(r, &b);
    return (uint16_t)((a << 8) | b);
}

static void rd_skip(rd_t* r, size_t k)
{
    /* seekg(k, cur) */
    if (r->eof) return;
    if (r->pos + k > r->n) {
        r->pos = r->n;
    } else {
        r->pos += k;
    }
}

/* parseJFIFSegment, Decoder.cpp:164-228 */
static void parse_app0(rd_t* r)
{
    uint8_t xt = 0, yt = 0, b;
    (void)rd_be16(r);  /* length: read, never used to skip            */
    rd_skip(r, 5);     /* "JFIF\0"                                    */
    rd_byte(r, &b);    /* major                                       */
    rd_byte(r, &b);    /* minor                                       */
    rd_byte(r, &b);    /* density unit                                */
    (void)rd_be16(r);  /* x density                                   */
    (void)rd_be16(r);  /* y density                                   */
    rd_byte(r, &xt);
    rd_byte(r, &yt);
    rd_skip(r, (size_t)3 * xt * yt);
}

/* parseComment, Decoder.cpp:579-619 */
static void parse_com(rd_t* r)
{
    uint16_t len = rd_be16(r);
    uint8_t b = 0;
    for (int i = 0; i < (int)len - 2; ++i) {
        rd_byte(r, &b);
        if (b == 0xFF) return; /* "Unexpected start of marker": gives up mid-segment */
    }
}

/* parseQuantizationTable, Decoder.cpp:230-299 */
static int parse_dqt(rd_t* r, kpeg_oracle_jfif* j)
{
    uint16_t len = rd_be16(r);
    len = (uint16_t)(len - 2);
    for (int t = 0; t < (int)len / 65; ++t) {
        uint8_t pqtq = 0, q = 0;
        rd_byte(r, &pqtq);
        int id = pqtq & 0x0F;
        /* m_QTables.push_back({}); m_QTables[id].push_back(Qi) x64:
         * only "id 0 first, then id 1" stays inside the vector. */
        if (j->nqt >= 4) return KPEG_ORACLE_OUT_OF_CONTRACT;
        j->nqt++;
        if (id >= j->nqt || id != j->nqt - 1) return KPEG_ORACLE_OUT_OF_CONTRACT;
        for (int i = 0; i < 64; ++i) {
            rd_byte(r, &q);
            j->qt[id][i] = q;
        }
    }
    return KPEG_ORACLE_SUCCESS;
}

/* parseSOF0Segment, Decoder.cpp:301-364 */
static int parse_sof0(rd_t* r, kpeg_oracle_jfif* j)
{
    uint8_t b = 0, id = 0, samp = 0, tq = 0;
    (void)rd_be16(r);
    rd_byte(r, &b); /* precision */
    uint16_t h = rd_be16(r);
    uint16_t w = rd_be16(r);
    rd_byte(r, &b); /* component count: logged, not used */
    int non_sampled = 1;
    for (int i = 0; i < 3; ++i) { /* always three triples */
        rd_byte(r, &id);
        rd_byte(r, &samp);
        rd_byte(r, &tq);
        if ((samp >> 4) != 1 || (samp & 0x0F) != 1) non_sampled = 0;
    }
    if (!non_sampled) return KPEG_ORACLE_TERMINATE;
    j->width = w;
    j->height = h;
    return KPEG_ORACLE_SUCCESS;
}

/* parseHuffmanTable, Decoder.cpp:366-459 */
static int parse_dht(rd_t* r, kpeg_oracle_jfif* j)
{
    uint16_t len = rd_be16(r);
    size_t seg_end = r->pos + len - 2;
    while (!r->eof && r->pos < seg_end) {
        uint8_t info = 0, c = 0;
        rd_byte(r, &info);
        int cls = (info & 0x10) >> 4;
        int id = info & 0x0F;
        if (id > 1) return KPEG_ORACLE_OUT_OF_CONTRACT; /* m_huffmanTable[2][2] */
        kpeg_oracle_dht* t = &j->dht[cls][id];
        if (t->defined) return KPEG_ORACLE_OUT_OF_CONTRACT; /* reference would append to the old lists */
        int total = 0;
        for (int i = 0; i < 16; ++i) {
            rd_byte(r, &c);
            t->counts[i] = c;
            total += c;
        }
        if (total > 256) return KPEG_ORACLE_OUT_OF_CONTRACT;
        for (int s = 0; s < total; ++s) {
            rd_byte(r, &c);
            t->symbols[s] = c;
        }
        t->nsymbols = total;
        t->defined = 1;
    }
    return KPEG_ORACLE_SUCCESS;
}

/* scanImageData, Decoder.cpp:532-577 */
static void scan_image_data(rd_t* r, kpeg_oracle_jfif* j)
{
    size_t cap = (r->n - r->pos) + 16, len = 0;
    uint8_t* out = (uint8_t*)malloc(cap);
    uint8_t b = 0;
    while (rd_byte(r, &b)) {
        if (b == 0xFF) {
            uint8_t prev = b;
            rd_byte(r, &b); /* on EOF b stays FF */
            if (b == 0xD9) break;
            out[len++] = prev;
        }
        out[len++] = b;
    }
    j->scan = out;
    j->scan_len = len;
}

/* parseSOSSegment, Decoder.cpp:461-530 */
static void parse_sos(rd_t* r, kpeg_oracle_jfif* j)
{
    uint8_t n = 0, b = 0;
    (void)rd_be16(r);
    rd_byte(r, &n);
    if (n < 1 || n > 4) return; /* "Invalid component count": returns without scanning */
    for (int i = 0; i < n; ++i) (void)rd_be16(r);
    for (int i = 0; i < 3; ++i) rd_byte(r, &b);
    if (j->scan) { /* a second SOS would append to m_scanData */
        free(j->scan);
        j->scan = NULL;
        j->saw_sos = 2;
    }
    scan_image_data(r, j);
    if (j->saw_sos == 0) j->saw_sos = 1;
}

int kpeg_oracle_parse(const uint8_t* file, size_t n, kpeg_oracle_jfif* j)
{
    rd_t r = {file, n, 0, 0};
    memset(j, 0, sizeof(*j));
    int status = KPEG_ORACLE_DECODE_DONE;
    uint8_t b = 0;
    /* decodeImageFile, Decoder.cpp:105-133 */
    while (rd_byte(&r, &b)) {
        if (b != 0xFF) {
            status = KPEG_ORACLE_ERROR;
            break;
        }
        rd_byte(&r, &b);
        /* parseSegmentInfo, Decoder.cpp:53-75 */
        int code = KPEG_ORACLE_SUCCESS;
        if (b == 0x00 || b == 0xFF) {
            code = KPEG_ORACLE_ERROR; /* neither continues nor breaks the loop */
        } else {
            switch (b) {
                case 0xD8: break;
                case 0xE0: parse_app0(&r); break;
                case 0xFE: parse_com(&r); break;
                case 0xDB: code = parse_dqt(&r, j); break;
                case 0xC0: code = parse_sof0(&r, j); break;
                case 0xC1:
                case 0xC2: code = KPEG_ORACLE_TERMINATE; break;
                case 0xC4: code = parse_dht(&r, j); break;
                case 0xDA: parse_sos(&r, j); break;
                default: break; /* unknown marker: payload NOT skipped */
            }
        }
        if (code == KPEG_ORACLE_OUT_OF_CONTRACT) return code;
        if (code == KPEG_ORACLE_TERMINATE) {
            status = KPEG_ORACLE_TERMINATE;
            break;
        }
        /* SUCCESS -> continue; ERROR falls through the if/else chain and the
         * loop simply goes on (Decoder.cpp:113-124). */
    }
    if (j->saw_sos == 2) return KPEG_ORACLE_OUT_OF_CONTRACT;
    return status;
}

void kpeg_oracle_jfif_free(kpeg_oracle_jfif* j)
{
    free(j->scan);
    j->scan = NULL;
}

/* ------------------------------------------------------------------ */
/* Un-stuffing                                                         */
/* ------------------------------------------------------------------ */

/* byteStuffScanData, Decoder.cpp:631-650: one left-to-right pass over the
 * shrinking string; an FF at current index b deletes the following 00 only if
 * b < nbytes_current - 2. */
size_t kpeg_oracle_unstuff(const uint8_t* in, size_t n, uint8_t* out)
{
    /* The reference walks the string with index b and erases in place (O(n) per
     * erase).  Same walk with a read cursor r and a write cursor w: w is the
     * reference's current index b, n - (r - w) its current length. */
    size_t r = 0, w = 0;
    while (r < n) {
        uint8_t s = in[r];
        size_t len = n - (r - w); /* current length of the shrinking string */
        size_t b = w;
        out[w++] = s;
        r++;
        if (s == 0xFF && len >= 1 && b + 1 < len - 1) {
            if (in[r] == 0x00) r++; /* erase the next byte; the walk moves past it */
        }
    }
    return w;
}

/* Literal restatement (erase in place), kept for the equivalence test. */
size_t kpeg_oracle_unstuff_literal(const uint8_t* in, size_t n, uint8_t* out)
{
    uint8_t* s = (uint8_t*)malloc(n ? n : 1);
    memcpy(s, in, n);
    size_t len = n;
    for (size_t b = 0; len >= 1 && b <= len - 1; ++b) {
        if (s[b] == 0xFF) {
            if (b + 1 < len - 1) {
                if (s[b + 1] == 0x00) {
                    memmove(s + b + 1, s + b + 2, len - (b + 2));
                    len--;
                }
            }
        }
    }
    memcpy(out, s, len);
    free(s);
    return len;
}

/* ------------------------------------------------------------------ */
/* Entropy decode                                                      */
/* ------------------------------------------------------------------ */

typedef struct {
    /* canonical code book: HuffmanTree::constructHuffmanTree (HuffmanTree.cpp:106-157)
     * hands out leaves left to right, level by level = the JPEG Annex C codes. */
    int32_t mincode[17], maxcode[17], valptr[17];
    const uint8_t* symbols;
} codebook;

static void build_codebook(const kpeg_oracle_dht* t, codebook* cb)
{
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        int cnt = t->counts[len - 1];
        cb->valptr[len] = k;
        cb->mincode[len] = code;
        cb->maxcode[len] = cnt ? code + cnt - 1 : -1;
        code += cnt;
        k += cnt;
        code <<= 1;
    }
    cb->symbols = t->symbols;
}

typedef struct {
    const uint8_t* p;
    uint64_t nbits, k;
} bitrd;

static inline int get_bit(bitrd* b, int* bit)
{
    if (b->k >= b->nbits) return 0;
    *bit = (b->p[b->k >> 3] >> (7 - (b->k & 7))) & 1;
    b->k++;
    return 1;
}

/* the `while(1){ bitsScanned += bit; contains(bitsScanned) ... }` loops,
 * Decoder.cpp:704-748 / 759-803 + HuffmanTree::contains (HuffmanTree.cpp:164-193) */
static int decode_symbol(bitrd* b, const codebook* cb, int* sym)
{
    int code = 0, bit = 0;
    for (int len = 1; len <= 16; ++len) {
        if (!get_bit(b, &bit)) return 0;
        code = (code << 1) | bit;
        if (cb->maxcode[len] >= 0 && code >= cb->mincode[len] && code <= cb->maxcode[len]) {
            *sym = cb->symbols[cb->valptr[len] + code - cb->mincode[len]];
            return 1;
        }
    }
    return 0; /* no code of <= 16 bits: the reference would loop on / throw */
}

/* bitStringtoValue, Image.cpp:285-302 (JPEG EXTEND; "" -> 0) */
static int receive_extend(bitrd* b, int cat, int* val)
{
    if (cat == 0) {
        *val = 0;
        return 1;
    }
    int v = 0, bit = 0, first = 0;
    for (int i = 0; i < cat; ++i) {
        if (!get_bit(b, &bit)) return 0;
        if (i == 0) first = bit;
        v = (v << 1) | bit;
    }
    if (first) {
        *val = (int16_t)v;
    } else {
        *val = (int16_t)(-(((1 << cat) - 1) - v));
    }
    return 1;
}

/* One block: DC symbol, then AC symbols until EOB or ACCodesCount == 63
 * (Decoder.cpp:694-803), then constructMCU's RLE walk (MCU.cpp:91-108):
 * the walk stops at the first (0,0) pair -- including a DC pair of (0,0) (Q1).
 * zz receives the 64 quantised values with the *difference* in zz[0];
 * returns 0 on a truncated/invalid stream. */
static int decode_block(bitrd* b, const codebook* dc, const codebook* ac, int zz[64])
{
    int sym = 0, val = 0;
    int stopped = 0; /* RLE walk has hit a (0,0) pair */
    int j = -1;
    memset(zz, 0, 64 * sizeof(int));

    if (!decode_symbol(b, dc, &sym)) return 0;
    if (sym != 0) { /* value != "EOB" */
        int run = sym >> 4, cat = sym & 0x0F;
        if (!receive_extend(b, cat, &val)) return 0;
        if (run == 0 && val == 0) {
            stopped = 1;
        } else {
            j += run + 1;
            if (j > 63) return 0;
            zz[j] = val;
        }
    } else {
        stopped = 1; /* DC "EOB": pair (0,0), Q1 */
    }

    int count = 0;
    while (count != 63) {
        if (!decode_symbol(b, ac, &sym)) return 0;
        if (sym == 0) break; /* EOB */
        int run = sym >> 4, cat = sym & 0x0F;
        if (!receive_extend(b, cat, &val)) return 0;
        count += run + 1;
        if (!stopped) {
            if (run == 0 && val == 0) {
                stopped = 1; /* unreachable for sym != 0, kept for symmetry */
            } else {
                j += run + 1;
                if (j > 63) return 0; /* reference: out-of-bounds write */
                zz[j] = val;
            }
        }
        if (count > 63) return 0; /* reference would keep decoding until an EOB; out of contract */
    }
    return 1;
}

static int have_tables(const kpeg_oracle_jfif* j)
{
    return j->nqt >= 2 && j->dht[0][0].defined && j->dht[0][1].defined && j->dht[1][0].defined &&
           j->dht[1][1].defined;
}

int kpeg_oracle_entropy_decode(const kpeg_oracle_jfif* j, const uint8_t* bits, size_t nbytes,
                               uint32_t nmcu, int16_t* coef, uint64_t* bits_used)
{
    if (!have_tables(j)) return KPEG_ORACLE_OUT_OF_CONTRACT;
    codebook cb[2][2];
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 2; ++i) build_codebook(&j->dht[c][i], &cb[c][i]);
    bitrd b = {bits, (uint64_t)nbytes * 8, 0};
    int pred[3] = {0, 0, 0}; /* MCU::DCDiff, MCU.cpp:53 (fresh process) */
    int zz[64];
    for (uint32_t m = 0; m < nmcu; ++m) {
        for (int c = 0; c < 3; ++c) {
            int id = c == 0 ? 0 : 1; /* Decoder.cpp:704 */
            if (!decode_block(&b, &cb[0][id], &cb[1][id], zz)) return KPEG_ORACLE_OUT_OF_CONTRACT;
            pred[c] += zz[0]; /* MCU.cpp:107-108 */
            zz[0] = pred[c];
            int16_t* o = coef + ((size_t)m * 3 + c) * 64;
            for (int k = 0; k < 64; ++k) {
                if (zz[k] < -32768 || zz[k] > 32767) return KPEG_ORACLE_OUT_OF_CONTRACT;
                o[k] = (int16_t)zz[k];
            }
        }
    }
    if (bits_used) *bits_used = b.k;
    return 0;
}

int kpeg_oracle_entropy_decode_rst(const kpeg_oracle_jfif* j, const uint8_t* scan, size_t nbytes,
                                   uint32_t nmcu, uint32_t interval, int16_t* coef)
{
    if (!have_tables(j) || interval == 0) return KPEG_ORACLE_OUT_OF_CONTRACT;
    kpeg_oracle_jfif jj = *j;
    size_t pos = 0;
    uint32_t done = 0;
    uint8_t* tmp = (uint8_t*)malloc(nbytes + 4);
    int rc = 0;
    while (done < nmcu) {
        /* find the end of this interval: next FF D0..D7 */
        size_t e = pos;
        while (e < nbytes) {
            if (scan[e] == 0xFF && e + 1 < nbytes && scan[e + 1] >= 0xD0 && scan[e + 1] <= 0xD7) break;
            e++;
        }
        /* each interval is what a re-wrapped single-interval JFIF would hold in
         * m_scanData; pad two bytes so the tail rule of byteStuffScanData never
         * bites inside real data. */
        size_t seg = e - pos;
        memcpy(tmp, scan + pos, seg);
        tmp[seg] = 0;
        tmp[seg + 1] = 0;
        size_t ulen = kpeg_oracle_unstuff(tmp, seg + 2, tmp);
        uint32_t cnt = nmcu - done < interval ? nmcu - done : interval;
        rc = kpeg_oracle_entropy_decode(&jj, tmp, ulen, cnt, coef + (size_t)done * 192, NULL);
        if (rc) break;
        done += cnt;
        pos = e + 2;
    }
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Dequantise, IDCT, level shift, colour, tiling                        */
/* ------------------------------------------------------------------ */

int kpeg_oracle_zz_to_rowmajor(int k)
{
    /* zzOrderToMatIndices, Transform.cpp:5-27, generated instead of tabulated:
     * walk the anti-diagonals, alternating direction. */
    static int tab[64], init = 0;
    if (!init) {
        int idx = 0;
        for (int s = 0; s < 15; ++s) {
            if (s & 1) { /* odd diagonal: row increasing */
                for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); ++r) tab[idx++] = r * 8 + (s - r);
            } else { /* even diagonal: row decreasing */
                for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); --r) tab[idx++] = r * 8 + (s - r);
            }
        }
        init = 1;
    }
    return tab[k];
}

static double g_cos[8][8];
static float g_cc[8][8];
static int g_tab_init = 0;

/* libm must be called at run time (a constant-folded cos is MPFR's, not glibc's) */
static double cos_rt(volatile double x) { return cos(x); }

static void init_tables(void)
{
    if (g_tab_init) return;
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) g_cos[a][b] = cos_rt((2 * a + 1) * b * M_PI / 16.0); /* MCU.cpp:192-193 */
    volatile double s2 = 2.0;
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            float cu = u == 0 ? (float)(1.0 / sqrt(s2)) : (float)1.0; /* MCU.cpp:189-190 */
            float cv = v == 0 ? (float)(1.0 / sqrt(s2)) : (float)1.0;
            g_cc[u][v] = cu * cv;
        }
    g_tab_init = 1;
}

void kpeg_oracle_cos_table(double out[64])
{
    init_tables();
    memcpy(out, g_cos, sizeof(g_cos));
}

/* F: dequantised block, row-major (u = row, v = col). out[x*8+y] = icoeffs[x][y].
 * Only non-zero F are visited, in the reference's (u outer, v inner) order:
 * a zero coefficient contributes +-0 and leaves the float accumulator unchanged. */
static void idct_rowmajor(const int F[64], float out[64])
{
    int nz[64], nnz = 0;
    for (int i = 0; i < 64; ++i)
        if (F[i] != 0) nz[nnz++] = i;
    float fc[64];
    for (int i = 0; i < nnz; ++i) {
        int p = nz[i];
        fc[i] = g_cc[p >> 3][p & 7] * (float)F[p]; /* Cu * Cv * m_8x8block (float) */
    }
    for (int x = 0; x < 8; ++x) {
        for (int y = 0; y < 8; ++y) {
            float sum = 0.0f;
            for (int i = 0; i < nnz; ++i) {
                int p = nz[i];
                double t = ((double)fc[i] * g_cos[x][p >> 3]) * g_cos[y][p & 7];
                sum = (float)((double)sum + t); /* sum += ... (MCU.cpp:192) */
            }
            out[x * 8 + y] = (float)(0.25 * (double)sum); /* MCU.cpp:198 */
        }
    }
}

static void dequant(const int16_t* zz, const uint16_t* q, int F[64])
{
    for (int k = 0; k < 64; ++k) F[kpeg_oracle_zz_to_rowmajor(k)] = (int)zz[k] * (int)q[k]; /* MCU.cpp:110-120 */
}

void kpeg_oracle_idct_block(const int16_t* zz, const uint16_t* q, float out[64])
{
    init_tables();
    int F[64];
    dequant(zz, q, F);
    idct_rowmajor(F, out);
}

static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static void mcu_to_rgb(const int16_t* coef, const uint16_t qt[2][64], uint8_t* rgb, size_t pitch)
{
    int S[3][64];
    int F[64];
    float ic[64];
    for (int c = 0; c < 3; ++c) {
        dequant(coef + c * 64, qt[c == 0 ? 0 : 1], F); /* MCU.cpp:110 */
        idct_rowmajor(F, ic);
        for (int i = 0; i < 64; ++i) S[c][i] = (int)(roundl((long double)ic[i]) + 128); /* MCU.cpp:228 */
    }
    for (int r = 0; r < 8; ++r) {
        for (int x = 0; x < 8; ++x) {
            float Y = (float)S[0][r * 8 + x]; /* MCU.cpp:255-257 */
            float Cb = (float)S[1][r * 8 + x];
            float Cr = (float)S[2][r * 8 + x];
            int R = (int)floor(Y + 1.402 * (1.0 * Cr - 128.0));
            int G = (int)floor(Y - 0.344136 * (1.0 * Cb - 128.0) - 0.714136 * (1.0 * Cr - 128.0));
            int B = (int)floor(Y + 1.772 * (1.0 * Cb - 128.0));
            uint8_t* o = rgb + r * pitch + x * 3;
            o[0] = (uint8_t)clamp255(R);
            o[1] = (uint8_t)clamp255(G);
            o[2] = (uint8_t)clamp255(B);
        }
    }
}

void kpeg_oracle_idct_colour(const int16_t* coef, const uint16_t qt[2][64], uint32_t width,
                             uint32_t height, uint8_t* rgb, int nthreads)
{
    init_tables();
    (void)kpeg_oracle_zz_to_rowmajor(0);
    const uint32_t mw = width / 8, mh = height / 8;
    const size_t pitch = (size_t)width * 3;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long tr = 0; tr < (long)mh; ++tr) {
        for (uint32_t tc = 0; tc < mw; ++tc) {
            /* Image::createImageFromMCUs, Image.cpp:51-68: MCU n -> tile (n / mw, n % mw) */
            size_t n = (size_t)tr * mw + tc;
            mcu_to_rgb(coef + n * 192, qt, rgb + (size_t)tr * 8 * pitch + (size_t)tc * 24, pitch);
        }
    }
}

static int decode_impl(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads, int any_size)
{
    kpeg_oracle_jfif j;
    int st = kpeg_oracle_parse(file, n, &j);
    if (st != KPEG_ORACLE_DECODE_DONE) {
        kpeg_oracle_jfif_free(&j);
        return st;
    }
    /* contract: SURVEY.md A.1 */
    if (!j.scan || j.scan_len == 0 || j.width == 0 || j.height == 0 || (!any_size && ((j.width & 7) || (j.height & 7))) ||
        !have_tables(&j)) {
        kpeg_oracle_jfif_free(&j);
        return KPEG_ORACLE_OUT_OF_CONTRACT;
    }
    /* any_size (extension): the padded size Image::createImageFromMCUs tiles (Image.cpp:26-27) */
    const uint32_t pw = (j.width + 7) & ~7u, ph = (j.height + 7) & ~7u;
    uint32_t nmcu = any_size ? (pw / 8) * (ph / 8) : (j.width * j.height) / 64; /* Decoder.cpp:670 */
    uint8_t* bits = (uint8_t*)malloc(j.scan_len);
    size_t nb = kpeg_oracle_unstuff(j.scan, j.scan_len, bits);
    int16_t* coef = (int16_t*)malloc((size_t)nmcu * 192 * sizeof(int16_t));
    int rc = kpeg_oracle_entropy_decode(&j, bits, nb, nmcu, coef, NULL);
    free(bits);
    if (rc) {
        free(coef);
        kpeg_oracle_jfif_free(&j);
        return rc;
    }
    uint8_t* out = (uint8_t*)malloc((size_t)pw * ph * 3);
    kpeg_oracle_idct_colour(coef, (const uint16_t(*)[64])j.qt, pw, ph, out, nthreads);
    free(coef);
    if (pw != j.width || ph != j.height) {
        /* the columns and rows Image::createImageFromMCUs pops (Image.cpp:73-84) */
        for (uint32_t y = 0; y < j.height; ++y) memmove(out + (size_t)y * j.width * 3, out + (size_t)y * pw * 3, (size_t)j.width * 3);
    }
    *rgb = out;
    *width = j.width;
    *height = j.height;
    kpeg_oracle_jfif_free(&j);
    return KPEG_ORACLE_DECODE_DONE;
}

int kpeg_oracle_decode(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads)
{
    return decode_impl(file, n, rgb, width, height, nthreads, 0);
}

/* Extension: width / height that are not multiples of 8.  The reference computes the MCU count as (w * h) / 64
 * (Decoder.cpp:670), decodes that many and then tiles ceil(w / 8) * ceil(h / 8) of them (Image.cpp:26-68): it reads past
 * the end of its MCU vector -- undefined behaviour, nothing to pin to.  PARITY UNPINNED: here every MCU of the padded
 * picture is decoded (what a JFIF encoder writes) and the picture is cropped the way createImageFromMCUs crops it. */
int kpeg_oracle_decode_any_size(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads)
{
    return decode_impl(file, n, rgb, width, height, nthreads, 1);
}

size_t kpeg_oracle_ppm_header(uint32_t width, uint32_t height, char* buf, size_t cap)
{
    /* Image::dumpRawData, Image.cpp:124-127 */
    int k = snprintf(buf, cap,
                     "P6\n# PPM dump created using libKPEG: https://github.com/TheIllusionistMirage/libKPEG\n%u %u\n255\n",
                     width, height);
    return k < 0 ? 0 : (size_t)k;
}

/* ------------------------------------------------------------------ */
/* Extension: one-component (grayscale) baseline files                  */
/* ------------------------------------------------------------------ */
/* The reference cannot decode these: parseSOF0Segment reads three component triples whatever the
 * frame header says (Decoder.cpp:339) and runs into the next marker.  PARITY UNPINNED: what is
 * restated here is the reference's own per-block path -- decodeScanData's symbol loop with its
 * quirk Q1 (Decoder.cpp:694-803, MCU.cpp:97-108), dequantisation, computeIDCT, performLevelShift
 * (MCU.cpp:110-245) -- applied to the one component with table id 0, and convertYCbCrToRGB
 * (MCU.cpp:247-279) on Cb = Cr = 128, i.e. R = G = B = clamp(Y).  The marker walk is the
 * standard one (segments skipped by their length).  Checked against Pillow's decoder within the
 * tolerance two different IDCTs allow (tests/test_oracle.py), not against the reference. */
int kpeg_oracle_decode_gray(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads)
{
    kpeg_oracle_jfif j;
    memset(&j, 0, sizeof(j));
    size_t pos = 2;
    if (n < 4 || file[0] != 0xFF || file[1] != 0xD8) return KPEG_ORACLE_ERROR;
    int have_sof = 0, have_q = 0;
    uint32_t ri = 0;
    while (pos + 4 <= n) {
        if (file[pos] != 0xFF) return KPEG_ORACLE_ERROR;
        const uint8_t m = file[pos + 1];
        const size_t len = ((size_t)file[pos + 2] << 8) | file[pos + 3];
        if (len < 2 || pos + 2 + len > n) return KPEG_ORACLE_ERROR;
        const uint8_t* p = file + pos + 4;
        const size_t body = len - 2;
        if (m == 0xDB) {
            for (size_t o = 0; o + 65 <= body; o += 65)
                if ((p[o] & 0x0F) == 0 && (p[o] >> 4) == 0) {
                    for (int i = 0; i < 64; ++i) j.qt[0][i] = j.qt[1][i] = p[o + 1 + i];
                    have_q = 1;
                }
        } else if (m == 0xC0) {
            if (body < 9 || p[0] != 8 || p[5] != 1 || p[7] != 0x11) return KPEG_ORACLE_OUT_OF_CONTRACT;
            j.height = ((uint32_t)p[1] << 8) | p[2];
            j.width = ((uint32_t)p[3] << 8) | p[4];
            have_sof = 1;
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o + 17 <= body) {
                const int cls = (p[o] >> 4) & 1, id = p[o] & 0x0F;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += p[o + 1 + i];
                if (o + 17 + (size_t)total > body || total > 256) return KPEG_ORACLE_ERROR;
                if (id == 0)
                    for (int t = 0; t < 2; ++t) {   /* id 0 stands in for id 1: one component */
                        kpeg_oracle_dht* d = &j.dht[cls][t];
                        memset(d, 0, sizeof(*d));
                        memcpy(d->counts, p + o + 1, 16);
                        memcpy(d->symbols, p + o + 17, (size_t)total);
                        d->nsymbols = total;
                        d->defined = 1;
                    }
                o += 17 + (size_t)total;
            }
        } else if (m == 0xDD) {
            if (body >= 2) ri = ((uint32_t)p[0] << 8) | p[1];   /* every interval is decoded as a stream of its own, as for colour (kpeg_oracle_entropy_decode_rst) */
        } else if (m == 0xDA) {
            if (body < 1 || p[0] != 1) return KPEG_ORACLE_OUT_OF_CONTRACT;
            pos += 2 + len;
            break;
        } else if (m == 0xC1 || m == 0xC2) {
            return KPEG_ORACLE_TERMINATE;
        }
        pos += 2 + len;
    }
    j.nqt = 2;
    if (!have_sof || !have_q || !have_tables(&j) || j.width == 0 || j.height == 0 || (j.width & 7) || (j.height & 7))
        return KPEG_ORACLE_OUT_OF_CONTRACT;
    /* scanImageData's rule (Decoder.cpp:544-574) on the bytes after the SOS header */
    uint8_t* scan = (uint8_t*)malloc(n - pos + 2);
    size_t sl = 0;
    while (pos < n) {
        uint8_t b = file[pos++];
        if (b == 0xFF) {
            const uint8_t nx = pos < n ? file[pos++] : 0xFF;
            if (nx == 0xD9) break;
            scan[sl++] = 0xFF;
            b = nx;
        }
        scan[sl++] = b;
    }
    const uint32_t nmcu = (j.width * j.height) / 64;
    int16_t* coef = (int16_t*)calloc((size_t)nmcu * 192, sizeof(int16_t));   /* chroma blocks stay zero: samples 0 + 128 */
    codebook dc, ac;
    build_codebook(&j.dht[0][0], &dc);
    build_codebook(&j.dht[1][0], &ac);
    uint8_t* tmp = (uint8_t*)malloc(sl + 4);
    int rc = 0;
    size_t at = 0;
    for (uint32_t done = 0; done < nmcu && !rc;) {
        size_t e = sl, ulen;
        uint32_t cnt = nmcu - done;
        if (ri) {
            for (e = at; e < sl; ++e)
                if (scan[e] == 0xFF && e + 1 < sl && scan[e + 1] >= 0xD0 && scan[e + 1] <= 0xD7) break;
            memcpy(tmp, scan + at, e - at);
            tmp[e - at] = tmp[e - at + 1] = 0;
            ulen = kpeg_oracle_unstuff(tmp, e - at + 2, tmp);
            if (cnt > ri) cnt = ri;
        } else {
            ulen = kpeg_oracle_unstuff(scan, sl, tmp);
        }
        bitrd b = {tmp, (uint64_t)ulen * 8, 0};
        int pred = 0, zz[64];
        for (uint32_t m = done; m < done + cnt && !rc; ++m) {
            if (!decode_block(&b, &dc, &ac, zz)) {
                rc = KPEG_ORACLE_OUT_OF_CONTRACT;
                break;
            }
            pred += zz[0];
            zz[0] = pred;
            for (int k = 0; k < 64; ++k) {
                if (zz[k] < -32768 || zz[k] > 32767) rc = KPEG_ORACLE_OUT_OF_CONTRACT;
                coef[(size_t)m * 192 + k] = (int16_t)zz[k];
            }
        }
        done += cnt;
        at = e + 2;
    }
    free(tmp);
    free(scan);
    if (rc) {
        free(coef);
        return rc;
    }
    uint8_t* out = (uint8_t*)malloc((size_t)j.width * j.height * 3);
    kpeg_oracle_idct_colour(coef, (const uint16_t(*)[64])j.qt, j.width, j.height, out, nthreads);
    free(coef);
    *rgb = out;
    *width = j.width;
    *height = j.height;
    return KPEG_ORACLE_DECODE_DONE;
}

/* ------------------------------------------------------------------ */
/* Extension: 4:2:0 (luma sampled 2x2, both chroma components 1x1)      */
/* ------------------------------------------------------------------ */
/* The reference answers TERMINATE on any sampling factor other than 1x1 (parseSOF0Segment, Decoder.cpp:339-356; fixture
 * rej_420.jpg).  PARITY UNPINNED: what is restated here is the reference's own per-block path -- decodeScanData's symbol
 * loop with quirk Q1, dequantisation, computeIDCT, performLevelShift -- applied to the six blocks of a 16x16 MCU in T.81's
 * order (Y00 Y01 Y10 Y11 Cb Cr: table pair 0 for the four Y blocks, 1 for Cb and Cr, one DC predictor per component),
 * every chroma sample repeated over its 2x2 luma samples (no interpolation: the simplest upsampling T.81 allows and the
 * only one that adds no arithmetic of its own), then convertYCbCrToRGB (MCU.cpp:247-279) per pixel, the picture padded to
 * whole MCUs and cropped as Image::createImageFromMCUs crops.  Checked against Pillow's decoder (libjpeg, which
 * interpolates chroma) within what that difference allows (tests/test_420.py), not against the reference. */
int kpeg_oracle_decode_420(const uint8_t* file, size_t n, uint8_t** rgb, uint32_t* width, uint32_t* height, int nthreads)
{
    kpeg_oracle_jfif j;
    memset(&j, 0, sizeof(j));
    size_t pos = 2;
    if (n < 4 || file[0] != 0xFF || file[1] != 0xD8) return KPEG_ORACLE_ERROR;
    int have_sof = 0;
    uint32_t ri = 0;
    while (pos + 4 <= n) {
        if (file[pos] != 0xFF) return KPEG_ORACLE_ERROR;
        const uint8_t m = file[pos + 1];
        const size_t len = ((size_t)file[pos + 2] << 8) | file[pos + 3];
        if (len < 2 || pos + 2 + len > n) return KPEG_ORACLE_ERROR;
        const uint8_t* p = file + pos + 4;
        const size_t body = len - 2;
        if (m == 0xDB) {
            for (size_t o = 0; o + 65 <= body; o += 65) {
                if (p[o] >> 4) return KPEG_ORACLE_OUT_OF_CONTRACT;   /* 16-bit tables */
                const int id = p[o] & 0x0F;
                if (id > 1) return KPEG_ORACLE_OUT_OF_CONTRACT;
                for (int i = 0; i < 64; ++i) j.qt[id][i] = p[o + 1 + i];
                j.nqt |= 1 << id;
            }
        } else if (m == 0xC0) {
            if (body < 15 || p[0] != 8 || p[5] != 3 || p[7] != 0x22 || p[8] != 0 || p[10] != 0x11 || p[11] != 1 || p[13] != 0x11 || p[14] != 1)
                return KPEG_ORACLE_OUT_OF_CONTRACT;
            j.height = ((uint32_t)p[1] << 8) | p[2];
            j.width = ((uint32_t)p[3] << 8) | p[4];
            have_sof = 1;
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o + 17 <= body) {
                const int cls = (p[o] >> 4) & 1, id = p[o] & 0x0F;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += p[o + 1 + i];
                if (id > 1 || o + 17 + (size_t)total > body || total > 256) return KPEG_ORACLE_OUT_OF_CONTRACT;
                kpeg_oracle_dht* d = &j.dht[cls][id];
                memset(d, 0, sizeof(*d));
                memcpy(d->counts, p + o + 1, 16);
                memcpy(d->symbols, p + o + 17, (size_t)total);
                d->nsymbols = total;
                d->defined = 1;
                o += 17 + (size_t)total;
            }
        } else if (m == 0xDD) {
            if (body >= 2) ri = ((uint32_t)p[0] << 8) | p[1];
        } else if (m == 0xDA) {
            /* three components, Y with table pair 0, Cb and Cr with pair 1: the reference's hard-wired selection (Decoder.cpp:704) */
            if (body < 7 || p[0] != 3 || p[2] != 0x00 || p[4] != 0x11 || p[6] != 0x11) return KPEG_ORACLE_OUT_OF_CONTRACT;
            pos += 2 + len;
            break;
        } else if (m == 0xC1 || m == 0xC2) {
            return KPEG_ORACLE_TERMINATE;
        }
        pos += 2 + len;
    }
    if (j.nqt == 3) j.nqt = 2;
    if (!have_sof || j.nqt != 2 || !have_tables(&j) || j.width == 0 || j.height == 0) return KPEG_ORACLE_OUT_OF_CONTRACT;
    uint8_t* scan = (uint8_t*)malloc(n - pos + 2);
    size_t sl = 0;
    while (pos < n) {   /* scanImageData's rule (Decoder.cpp:544-574) */
        uint8_t b = file[pos++];
        if (b == 0xFF) {
            const uint8_t nx = pos < n ? file[pos++] : 0xFF;
            if (nx == 0xD9) break;
            scan[sl++] = 0xFF;
            b = nx;
        }
        scan[sl++] = b;
    }
    const uint32_t mw = (j.width + 15) / 16, mh = (j.height + 15) / 16, nmcu = mw * mh;
    int16_t* coef = (int16_t*)calloc((size_t)nmcu * 384, sizeof(int16_t));   /* [mcu][Y00 Y01 Y10 Y11 Cb Cr][64] */
    codebook cb[2][2];
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 2; ++i) build_codebook(&j.dht[c][i], &cb[c][i]);
    uint8_t* tmp = (uint8_t*)malloc(sl + 4);
    int rc = 0;
    size_t at = 0;
    for (uint32_t done = 0; done < nmcu && !rc;) {
        size_t e = sl, ulen;
        uint32_t cnt = nmcu - done;
        if (ri) {   /* every restart interval as a stream of its own, as kpeg_oracle_entropy_decode_rst does */
            for (e = at; e < sl; ++e)
                if (scan[e] == 0xFF && e + 1 < sl && scan[e + 1] >= 0xD0 && scan[e + 1] <= 0xD7) break;
            memcpy(tmp, scan + at, e - at);
            tmp[e - at] = tmp[e - at + 1] = 0;
            ulen = kpeg_oracle_unstuff(tmp, e - at + 2, tmp);
            if (cnt > ri) cnt = ri;
        } else {
            ulen = kpeg_oracle_unstuff(scan, sl, tmp);
        }
        bitrd b = {tmp, (uint64_t)ulen * 8, 0};
        int pred[3] = {0, 0, 0}, zz[64];
        for (uint32_t m = done; m < done + cnt && !rc; ++m)
            for (int blk = 0; blk < 6 && !rc; ++blk) {
                const int comp = blk < 4 ? 0 : blk - 3, id = blk < 4 ? 0 : 1;
                if (!decode_block(&b, &cb[0][id], &cb[1][id], zz)) {
                    rc = KPEG_ORACLE_OUT_OF_CONTRACT;
                    break;
                }
                pred[comp] += zz[0];
                zz[0] = pred[comp];
                for (int k = 0; k < 64; ++k) {
                    if (zz[k] < -32768 || zz[k] > 32767) rc = KPEG_ORACLE_OUT_OF_CONTRACT;
                    coef[((size_t)m * 6 + blk) * 64 + k] = (int16_t)zz[k];
                }
            }
        done += cnt;
        at = e + 2;
    }
    free(tmp);
    free(scan);
    if (rc) {
        free(coef);
        return rc;
    }
    init_tables();
    (void)kpeg_oracle_zz_to_rowmajor(0);
    uint8_t* out = (uint8_t*)malloc((size_t)j.width * j.height * 3);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long tr = 0; tr < (long)mh; ++tr)
        for (uint32_t tc = 0; tc < mw; ++tc) {
            int S[6][64], F[64];
            float ic[64];
            const int16_t* c6 = coef + ((size_t)tr * mw + tc) * 384;
            for (int blk = 0; blk < 6; ++blk) {
                dequant(c6 + blk * 64, j.qt[blk < 4 ? 0 : 1], F);
                idct_rowmajor(F, ic);
                for (int i = 0; i < 64; ++i) S[blk][i] = (int)(roundl((long double)ic[i]) + 128);
            }
            for (int py = 0; py < 16; ++py)
                for (int px = 0; px < 16; ++px) {
                    const uint32_t y = (uint32_t)tr * 16 + py, x = tc * 16 + px;
                    if (y >= j.height || x >= j.width) continue;   /* the rows and columns createImageFromMCUs pops */
                    const float Y = (float)S[(py >> 3) * 2 + (px >> 3)][(py & 7) * 8 + (px & 7)];
                    const float Cb = (float)S[4][(py >> 1) * 8 + (px >> 1)], Cr = (float)S[5][(py >> 1) * 8 + (px >> 1)];
                    const int R = (int)floor(Y + 1.402 * (1.0 * Cr - 128.0));
                    const int G = (int)floor(Y - 0.344136 * (1.0 * Cb - 128.0) - 0.714136 * (1.0 * Cr - 128.0));
                    const int B = (int)floor(Y + 1.772 * (1.0 * Cb - 128.0));
                    uint8_t* o = out + ((size_t)y * j.width + x) * 3;
                    o[0] = (uint8_t)clamp255(R);
                    o[1] = (uint8_t)clamp255(G);
                    o[2] = (uint8_t)clamp255(B);
                }
        }
    free(coef);
    *rgb = out;
    *width = j.width;
    *height = j.height;
    return KPEG_ORACLE_DECODE_DONE;
}
