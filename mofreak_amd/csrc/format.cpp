// .mofreak text rows (host only): the on-disk contract between feature extraction and the rest of the
// reference pipeline.  Writer = MoFREAKUtilities::writeMoFREAKFeaturesToFile (MoFREAKUtilities.cpp:691-719),
// reader = the row loop of MoFREAKUtilities::readMoFREAKFeatures (:1146-1190).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mofreak_hip.h"

namespace {

// `ostream << float` with default flags is printf("%g") (precision 6).  Small integral values -- pixel
// coordinates, sizes like 12 -- take a shortcut that prints the same characters.
inline int put_float(char *dst, float v)
{
    if (v >= 0.0f && v < 100000.0f && v == (float)(int)v && !(v == 0.0f && std::signbit(v))) {
        int n = 0;
        unsigned u = (unsigned)(int)v;
        char tmp[8];
        do {
            tmp[n++] = (char)('0' + u % 10);
            u /= 10;
        } while (u);
        for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
        return n;
    }
    return std::snprintf(dst, 32, "%g", (double)v);
}

inline int put_uint(char *dst, unsigned u)
{
    char tmp[12];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}

inline int put_int(char *dst, int v)
{
    if (v < 0) {
        dst[0] = '-';
        return 1 + put_uint(dst + 1, 0u - (unsigned)v);
    }
    return put_uint(dst, (unsigned)v);
}

// One row: "x y frame scale motion_x motion_y a0..a7 m0..m7 \n", every value followed by one space.
// motion_x and motion_y are always 0 (MoFREAKUtilities.cpp:476-477).
int format_one(const mofreak_row &r, char *b)
{
    int n = 0;
    n += put_float(b + n, r.x);
    b[n++] = ' ';
    n += put_float(b + n, r.y);
    b[n++] = ' ';
    n += put_int(b + n, r.frame_number);
    b[n++] = ' ';
    n += put_float(b + n, r.scale);
    b[n++] = ' ';
    b[n++] = '0';
    b[n++] = ' ';
    b[n++] = '0';
    b[n++] = ' ';
    for (int i = 0; i < MOFREAK_APPEARANCE_BYTES; ++i) {
        n += put_uint(b + n, r.appearance[i]);
        b[n++] = ' ';
    }
    for (int i = 0; i < MOFREAK_MOTION_BYTES; ++i) {
        n += put_uint(b + n, r.motion[i]);
        b[n++] = ' ';
    }
    b[n++] = '\n';
    return n;
}

inline const char *skip_ws(const char *p, const char *end)
{
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
    return p;
}

}  // namespace

extern "C" {

int mofreak_format_rows(const mofreak_row *rows, int64_t n_rows, char *buf, size_t cap, size_t *needed)
{
    if (n_rows < 0 || (n_rows > 0 && !rows)) return MOFREAK_ERR_BAD_ARG;
    size_t total = 0;
    char line[512];
    for (int64_t i = 0; i < n_rows; ++i) {
        const int n = format_one(rows[i], line);
        if (buf && total < cap) std::memcpy(buf + total, line, total + n <= cap ? (size_t)n : cap - total);
        total += (size_t)n;
    }
    if (needed) *needed = total;
    return (buf && total > cap) ? MOFREAK_ERR_CAPACITY : MOFREAK_OK;
}

int mofreak_parse_rows(const char *text, size_t len, mofreak_row *rows, int64_t rows_capacity, int64_t *n_rows_out)
{
    if (!text && len) return MOFREAK_ERR_BAD_ARG;
    // strtof/strtol need a terminator: work on a NUL-terminated copy
    char *copy = (char *)std::malloc(len + 1);
    if (!copy) return MOFREAK_ERR_OOM;
    std::memcpy(copy, text, len);
    copy[len] = 0;
    const char *p = copy, *end = copy + len;
    int64_t n = 0;
    int rc = MOFREAK_OK;
    for (;;) {
        p = skip_ws(p, end);
        if (p >= end) break;  // the reference leaves its loop at EOF after the six leading fields fail to read
        float f[5];
        long frame = 0;
        unsigned bytes[16];
        char *q;
        bool bad = false;
        auto next_float = [&](float &dst) {
            dst = std::strtof(p, &q);
            if (q == p) bad = true;
            p = q;
        };
        next_float(f[0]);
        next_float(f[1]);
        if (!bad) {
            frame = std::strtol(p, &q, 10);
            if (q == p) bad = true;
            p = q;
        }
        next_float(f[2]);
        next_float(f[3]);
        next_float(f[4]);
        for (int i = 0; i < 16 && !bad; ++i) {
            const unsigned long u = std::strtoul(p, &q, 10);
            if (q == p) bad = true;
            p = q;
            bytes[i] = (unsigned)u;
        }
        if (bad) {
            rc = MOFREAK_ERR_BAD_ARG;
            break;
        }
        if (rows) {
            if (n >= rows_capacity) {
                rc = MOFREAK_ERR_CAPACITY;
                break;
            }
            mofreak_row &r = rows[n];
            r.x = f[0];
            r.y = f[1];
            r.frame_number = (int32_t)frame;
            r.scale = f[2];
            for (int i = 0; i < 8; ++i) r.appearance[i] = (uint8_t)bytes[i];
            for (int i = 0; i < 8; ++i) r.motion[i] = (uint8_t)bytes[8 + i];
        }
        ++n;
    }
    std::free(copy);
    if (n_rows_out) *n_rows_out = n;
    return rc;
}

}  // extern "C"
