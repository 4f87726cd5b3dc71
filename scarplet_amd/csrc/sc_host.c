/* libscarplet_host.so: host-only helpers (include/scarplet_host.h).  gcc, no ROCm. */
#include "../../include/scarplet_host.h"

struct lzw_table {
    unsigned short prefix[4096], length[4096];
    unsigned char suffix[4096], first[4096];
};

/* the string of `code` to dst[*pos ..); 0 when it does not fit */
static int lzw_put(const struct lzw_table* t, int code, unsigned char* dst, size_t cap, size_t* pos) {
    const size_t len = t->length[code];
    size_t q;
    if (*pos + len > cap) return 0;
    q = *pos + len;
    while (code != 0xFFFF && q > *pos) {
        dst[--q] = t->suffix[code];
        code = t->prefix[code];
    }
    *pos += len;
    return 1;
}

long long sch_tiff_lzw_decode(const unsigned char* src, size_t n, unsigned char* dst, size_t cap) {
    struct lzw_table t;
    size_t pos = 0, bitpos = 0;
    const size_t nbits_total = n * 8;
    int nbits = 9, next = 258, old = -1, i;
    if (!src || !dst) return -1;
    for (i = 0; i < 4096; ++i) {
        t.prefix[i] = 0xFFFF;
        t.suffix[i] = t.first[i] = (unsigned char)i;
        t.length[i] = i < 256 ? 1 : 0;
    }
    while (bitpos + (size_t)nbits <= nbits_total) {
        /* MSB-first: the next nbits bits starting at bitpos */
        const size_t byte = bitpos >> 3;
        unsigned int w = (unsigned int)src[byte] << 16;
        int code;
        if (byte + 1 < n) w |= (unsigned int)src[byte + 1] << 8;
        if (byte + 2 < n) w |= (unsigned int)src[byte + 2];
        code = (int)((w >> (24 - nbits - (int)(bitpos & 7))) & ((1u << nbits) - 1));
        bitpos += (size_t)nbits;
        if (code == 257) break;                          /* EndOfInformation */
        if (code == 256) { nbits = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code > 255) return -1;
            if (!lzw_put(&t, code, dst, cap, &pos)) return -2;
            old = code;
            continue;
        }
        if (code < next) {
            if (code >= 258 && t.length[code] == 0) return -1;
            if (!lzw_put(&t, code, dst, cap, &pos)) return -2;
            if (next < 4096) {
                t.prefix[next] = (unsigned short)old; t.suffix[next] = t.first[code];
                t.first[next] = t.first[old]; t.length[next] = (unsigned short)(t.length[old] + 1);
                ++next;
            }
        } else if (code == next && next < 4096) {
            t.prefix[next] = (unsigned short)old; t.suffix[next] = t.first[old];
            t.first[next] = t.first[old]; t.length[next] = (unsigned short)(t.length[old] + 1);
            ++next;
            if (!lzw_put(&t, code, dst, cap, &pos)) return -2;
        } else {
            return -1;
        }
        old = code;
        if (next >= (1 << nbits) - 1 && nbits < 12) ++nbits;        /* one code early */
    }
    return (long long)pos;
}
