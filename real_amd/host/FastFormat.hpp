// FastFormat.hpp -- "%g" of a float, digit for digit what printf / operator<<(float) print, without printf.
//
// The output's score column is operator<<(std::ostream&, float) in the reference (matchUniqueImplementation.cpp:266-270):
// "%g" with six significant digits of the float's exact value, round-half-even on that exact value.  A float is
// m * 2^e with m < 2^24, so for 1e-5 <= |v| < 1e15 the value scaled to six digits is a quotient of two 64-bit
// integers and the correctly rounded digits come out of one division; everything else (0, tiny, huge, inf, nan) goes
// to snprintf.  host_selftest fmtcheck compares the two on 10^8 floats.
#pragma once
#include <stdint.h>
#include <cstdio>
#include <cstring>

namespace fastformat {

static const uint64_t kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                    10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                                    1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};

// writes at most 16 characters (no terminator), returns their number
inline int fmt_g6(float f, char *out)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t expo = (u >> 23) & 0xff;
    const float af = f < 0 ? -f : f;
    if (expo == 0 || expo == 0xff || !(af >= 1e-5f) || !(af < 1e15f)) return snprintf(out, 32, "%g", (double)f);
    char *p = out;
    if (u >> 31) *p++ = '-';
    const uint64_t m = (u & 0x7fffffu) | 0x800000u;
    const int e = (int)expo - 150; // |v| = m * 2^e
    // decimal exponent of the leading digit: estimate from the binary exponent, corrected by the digits themselves
    int X = (((int)expo - 127) * 1233) >> 12;
    uint64_t D = 0;
    for (int guard = 0; guard < 4; ++guard) {
        const int s = 5 - X; // |v| * 10^s has six digits in front of the point
        uint64_t A = m, B = 1;
        if (s >= 0) A *= kPow10[s]; else B *= kPow10[-s];
        if (e >= 0) A <<= e; else B <<= -e;
        D = A / B;
        const uint64_t r = A - D * B;
        if (2 * r > B || (2 * r == B && (D & 1))) D++;
        if (D >= 1000000) { X++; continue; } // (also when the rounding carried into the next decade: 999999.5 -> 1.00000e+06)
        if (D < 100000) { X--; continue; }
        break;
    }
    char d[6];
    for (int i = 5; i >= 0; --i) { d[i] = (char)('0' + D % 10); D /= 10; }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') nd--; // %g strips trailing zeros
    if (X < -4 || X >= 6) { // scientific
        *p++ = d[0];
        if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = d[i]; }
        *p++ = 'e';
        int ax = X;
        if (ax < 0) { *p++ = '-'; ax = -ax; } else *p++ = '+';
        if (ax >= 100) { *p++ = (char)('0' + ax / 100); ax %= 100; }
        *p++ = (char)('0' + ax / 10); *p++ = (char)('0' + ax % 10);
    } else if (X >= 0) { // X + 1 digits in front of the point
        for (int i = 0; i <= X; ++i) *p++ = i < nd ? d[i] : '0';
        if (nd > X + 1) { *p++ = '.'; for (int i = X + 1; i < nd; ++i) *p++ = d[i]; }
    } else { // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -X - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = d[i];
    }
    return (int)(p - out);
}

} // namespace fastformat
