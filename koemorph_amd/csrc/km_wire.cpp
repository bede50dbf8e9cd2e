// km_wire.cpp -- host-side wire encoding of blendshape frames (no GPU involved).
//
// The reference emits one UTF-8 JSON text per output frame, for UDP datagrams, JSONL files (scripts/rt.py:209-231)
// and dataset labels (src/data/io.py:119-131):
//     json.dumps({"timestamp": <float>, "blendshapes": <52 floats as a list>})
// At 1024 streams x 30 fps that is 30 720 json.dumps calls per second on the host thread that also drives the GPU.
// km_format_frames produces byte-identical text for a whole tick in one call: CPython's float repr is the shortest
// decimal string that round-trips the double (here: the float32 value widened to double, as ndarray.tolist() does),
// printed positionally for 1e-4 <= |x| < 1e16 and as d.ddde+XX otherwise; std::to_chars yields the same shortest
// digits, the layout rules are applied below (CPython Objects/floatobject.c float_repr -> PyOS_double_to_string 'r').
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "koemorph.h"

namespace {

// Python repr(float) / json.dumps of a finite or non-finite double; returns the number of bytes written (<= 32).
int py_float_repr(double v, char* out) {
    if (std::isnan(v)) { std::memcpy(out, "NaN", 3); return 3; }                       // json.dumps spelling
    if (std::isinf(v)) {
        if (v < 0) { std::memcpy(out, "-Infinity", 9); return 9; }
        std::memcpy(out, "Infinity", 8); return 8;
    }
    char* p = out;
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0) { std::memcpy(p, "0.0", 3); return (int)(p + 3 - out); }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int exp10 = 0;
    {
        ++q;                                  // past 'e'
        const bool neg = *q == '-';
        ++q;
        for (; q < r.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
        if (neg) exp10 = -exp10;
    }
    const int decpt = exp10 + 1;              // value = 0.d1d2...dn * 10^decpt
    if (decpt <= -4 || decpt > 16) {          // exponent form: d[.ddd]e+XX with at least two exponent digits
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; std::memcpy(p, digits + 1, nd - 1); p += nd - 1; }
        *p++ = 'e';
        int e = exp10;
        if (e < 0) { *p++ = '-'; e = -e; } else { *p++ = '+'; }
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *p++ = eb[--ne];
    } else if (decpt <= 0) {                  // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        std::memcpy(p, digits, nd); p += nd;
    } else if (decpt >= nd) {                 // ddd000.0
        std::memcpy(p, digits, nd); p += nd;
        for (int i = 0; i < decpt - nd; ++i) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else {                                  // dd.ddd
        std::memcpy(p, digits, decpt); p += decpt;
        *p++ = '.';
        std::memcpy(p, digits + decpt, nd - decpt); p += nd - decpt;
    }
    return (int)(p - out);
}

}  // namespace

extern "C" int64_t km_format_frames(const float* frames_host, int64_t n_frames, int32_t n_values, const double* timestamps,
                                    int32_t newline, char* out, int64_t capacity, int64_t* offsets) {
    if (!frames_host || !timestamps || n_frames < 0 || n_values < 0 || (!out && capacity > 0)) return KM_ERR_INVALID_ARG;
    static const char k1[] = "{\"timestamp\": ", k2[] = ", \"blendshapes\": [";
    const int64_t per_frame_max = (int64_t)sizeof(k1) + sizeof(k2) + 32 + (int64_t)n_values * 34 + 4;
    int64_t pos = 0;
    for (int64_t f = 0; f < n_frames; ++f) {
        if (offsets) offsets[f] = pos;
        if (pos + per_frame_max > capacity) return -(pos + (n_frames - f) * per_frame_max);   // -(bytes that certainly suffice)
        char* p = out + pos;
        std::memcpy(p, k1, sizeof(k1) - 1); p += sizeof(k1) - 1;
        p += py_float_repr(timestamps[f], p);
        std::memcpy(p, k2, sizeof(k2) - 1); p += sizeof(k2) - 1;
        const float* row = frames_host + f * n_values;
        for (int32_t i = 0; i < n_values; ++i) {
            if (i) { *p++ = ','; *p++ = ' '; }
            p += py_float_repr((double)row[i], p);
        }
        *p++ = ']'; *p++ = '}';
        if (newline) *p++ = '\n';
        pos = p - out;
    }
    if (offsets) offsets[n_frames] = pos;
    return pos;
}
