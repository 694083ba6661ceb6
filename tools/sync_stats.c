/* tools/sync_stats.c -- how far does a Huffman decoder started at an arbitrary bit run before it is in
 * step with the true decode?  (design study for K1's sub-sequence size; uses the oracle's parser) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../oracle/kpeg_oracle.h"

typedef struct { int32_t mincode[17], maxcode[17], valptr[17]; const uint8_t* sym; } cb_t;
static void build(const kpeg_oracle_dht* t, cb_t* cb) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) { int c = t->counts[l-1]; cb->valptr[l] = k; cb->mincode[l] = code; cb->maxcode[l] = c ? code + c - 1 : -1; code += c; k += c; code <<= 1; }
    cb->sym = t->symbols;
}
static const uint8_t* B; static uint64_t NB;
static inline int bit(uint64_t p) { return p < NB ? (B[p >> 3] >> (7 - (p & 7))) & 1 : 0; }
static int sym(const cb_t* cb, uint64_t* p) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) { code = (code << 1) | bit((*p)++); if (cb->maxcode[l] >= 0 && code >= cb->mincode[l] && code <= cb->maxcode[l]) return cb->sym[cb->valptr[l] + code - cb->mincode[l]]; }
    return 0;
}
typedef struct { uint64_t p; int c, k; } st_t;
static void step(const cb_t cb[2][2], st_t* s) {
    int id = s->c ? 1 : 0;
    if (s->k == 0) { int v = sym(&cb[0][id], &s->p); s->p += v & 15; s->k = 1; }
    else { int v = sym(&cb[1][id], &s->p); if (v == 0) s->k = 64; else { s->p += v & 15; s->k += (v >> 4) + 1; }
           if (s->k >= 64) { s->k = 0; s->c = (s->c + 1) % 3; } }
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* d = malloc(n); fread(d, 1, n, f); fclose(f);
    int L = argc > 2 ? atoi(argv[2]) : 512;
    kpeg_oracle_jfif j; if (kpeg_oracle_parse(d, n, &j) != 4) return 1;
    uint8_t* u = malloc(j.scan_len); size_t nu = kpeg_oracle_unstuff(j.scan, j.scan_len, u);
    B = u; NB = (uint64_t)nu * 8;
    cb_t cb[2][2]; for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) build(&j.dht[c][i], &cb[c][i]);
    uint32_t nmcu = j.width * j.height / 64;
    /* true decode: state code at every codeword boundary */
    uint8_t* mark = calloc(NB + 64, 1);  /* 0 = not a boundary, else 1 + c*64 + k */
    st_t s = {0, 0, 0}; uint64_t blocks = 0;
    while (blocks < (uint64_t)nmcu * 3) { mark[s.p] = (uint8_t)(1 + s.c * 64 + s.k); int wask = s.k; step(cb, &s); if (s.k == 0 && wask != 0) blocks++; else if (0) {} }
    uint64_t end = s.p;
    uint64_t hist[16] = {0}, cnt = 0, sum = 0, mx = 0;
    for (uint64_t p0 = L; p0 + 4096 < end; p0 += L) {
        st_t t = {p0, 0, 0};
        while (t.p < end && mark[t.p] != (uint8_t)(1 + t.c * 64 + t.k)) step(cb, &t);
        uint64_t dist = t.p - p0; cnt++; sum += dist; if (dist > mx) mx = dist;
        int b = 0; while ((dist >> b) > 0 && b < 15) b++; hist[b]++;
    }
    printf("L=%d starts=%llu mean sync distance %.1f bits, max %llu\n", L, (unsigned long long)cnt, (double)sum / cnt, (unsigned long long)mx);
    for (int b = 0; b < 16; ++b) if (hist[b]) printf("  < 2^%-2d bits: %llu (%.2f%%)\n", b, (unsigned long long)hist[b], 100.0 * hist[b] / cnt);
    return 0;
}
