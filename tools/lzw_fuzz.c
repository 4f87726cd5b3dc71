/* Mutation fuzzer of sch_tiff_lzw_decode (scarplet_amd/csrc/sc_host.c), built by `make asan` with
 * -fsanitize=address,undefined (CPU only): the decoder parses bytes of untrusted files.
 *
 *   lzw_fuzz <cases> <seed> <strip file> [<strip file> ...]
 *
 * Every strip file holds one LZW-compressed strip or tile of a TIFF fixture.  Each case takes one
 * of them, applies 1 .. 8 mutations (bit flips, byte sets, truncation, a duplicated or zeroed run,
 * a spliced piece of another strip) and decodes it into a heap block of EXACTLY the capacity
 * passed to the decoder - one byte beyond it is a sanitizer report - with capacities of 0, 1, a
 * random size, the true decoded size and one more than that.  The unmutated strips must decode to
 * the same bytes at every capacity that holds them.  Exit code 0: no report, every invariant held. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/scarplet_host.h"

static unsigned long long rng_state;
static unsigned int rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (unsigned int)(rng_state >> 11);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: lzw_fuzz <cases> <seed> <strip files>\n"); return 2; }
    const long cases = atol(argv[1]);
    rng_state = strtoull(argv[2], 0, 10) * 0x9E3779B97F4A7C15ull + 1;
    const int ns = argc - 3;
    unsigned char** strip = calloc(ns, sizeof *strip);
    size_t* len = calloc(ns, sizeof *len);
    long long* full = calloc(ns, sizeof *full);
    for (int i = 0; i < ns; ++i) {
        FILE* f = fopen(argv[3 + i], "rb");
        if (!f) { perror(argv[3 + i]); return 2; }
        fseek(f, 0, SEEK_END); len[i] = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
        strip[i] = malloc(len[i] ? len[i] : 1);
        if (fread(strip[i], 1, len[i], f) != len[i]) return 2;
        fclose(f);
        /* the decoded size of the intact strip, found with a generous block */
        size_t cap = len[i] * 3000 + 16;
        unsigned char* d = malloc(cap);
        full[i] = sch_tiff_lzw_decode(strip[i], len[i], d, cap);
        if (full[i] <= 0) { fprintf(stderr, "seed strip %s does not decode (%lld)\n", argv[3 + i], full[i]); return 1; }
        /* exact capacity: the same bytes; one byte less: -2 */
        unsigned char* e = malloc((size_t)full[i]);
        if (sch_tiff_lzw_decode(strip[i], len[i], e, (size_t)full[i]) != full[i] || memcmp(d, e, (size_t)full[i])) {
            fprintf(stderr, "seed strip %s: exact-capacity decode differs\n", argv[3 + i]); return 1;
        }
        if (full[i] > 1 && sch_tiff_lzw_decode(strip[i], len[i], e, (size_t)full[i] - 1) != -2) {
            fprintf(stderr, "seed strip %s: a block one byte short is not refused\n", argv[3 + i]); return 1;
        }
        free(d); free(e);
    }
    long ok = 0, malformed = 0, toosmall = 0;
    for (long c = 0; c < cases; ++c) {
        const int i = (int)(rnd() % ns);
        size_t n = len[i];
        unsigned char* src = malloc(n + 64);
        memcpy(src, strip[i], n);
        const int muts = 1 + (int)(rnd() % 8);
        for (int m = 0; m < muts && n > 0; ++m) {
            const size_t at = rnd() % n;
            switch (rnd() % 6) {
            case 0: src[at] ^= (unsigned char)(1u << (rnd() % 8)); break;
            case 1: src[at] = (unsigned char)rnd(); break;
            case 2: n = at + 1; break;                                           /* truncate */
            case 3: {                                                            /* a run of 0x00 or 0xFF */
                size_t r = rnd() % 32;
                if (at + r > n) r = n - at;
                memset(src + at, (rnd() & 1) ? 0 : 0xFF, r);
            } break;
            case 4: {                                                            /* a run copied from elsewhere */
                size_t r = rnd() % 32, from = rnd() % n;
                if (at + r > n) r = n - at;
                if (from + r > n) r = n - from;
                memmove(src + at, src + from, r);
            } break;
            default: {                                                           /* a piece of another strip */
                const int j = (int)(rnd() % ns);
                size_t r = rnd() % 48, from = len[j] ? rnd() % len[j] : 0;
                if (at + r > n) r = n - at;
                if (from + r > len[j]) r = len[j] - from;
                memcpy(src + at, strip[j] + from, r);
            } break;
            }
        }
        /* the source in a block of exactly n bytes: a read past the end is a report too */
        unsigned char* tight = malloc(n ? n : 1);
        memcpy(tight, src, n);
        size_t cap;
        switch (rnd() % 5) {
        case 0: cap = 0; break;
        case 1: cap = 1; break;
        case 2: cap = (size_t)full[i]; break;
        case 3: cap = (size_t)full[i] + 1; break;
        default: cap = rnd() % ((size_t)full[i] * 2 + 2); break;
        }
        unsigned char* dst = malloc(cap ? cap : 1);
        const long long r = sch_tiff_lzw_decode(tight, n, dst, cap);
        if (r > (long long)cap || r < -2) { fprintf(stderr, "case %ld: return %lld with capacity %zu\n", c, r, cap); return 1; }
        if (r >= 0) ++ok; else if (r == -1) ++malformed; else ++toosmall;
        free(dst); free(tight); free(src);
    }
    printf("lzw_fuzz: %ld cases over %d strips: %ld decoded, %ld malformed, %ld refused for size; no sanitizer report\n",
           cases, ns, ok, malformed, toosmall);
    return 0;
}
