// Host build of minnow_amd/csrc/mnw_text_num.h for tests/test_text_num_cpu.py: the same source the device compiles,
// exported with a C interface so that Python can compare every conversion with int() / float().
// g++ -O2 -ffp-contract=off -shared -fPIC
#include <cstring>
#include "../../minnow_amd/csrc/mnw_text_num.h"

static void window(const uint8_t* tok, int len, uint32_t (&w)[4]) {   // the 16-byte window as the kernel builds it: the token, then its terminator
    uint8_t b[16];
    for (int i = 0; i < 16; ++i) b[i] = i < len ? tok[i] : (uint8_t)' ';
    std::memcpy(w, b, 16);
}

extern "C" {
// generic forms: rc of text_atoi (1 ok / 0 error), of text_atof (0 value / 1 host list / 2 error)
int tn_atoi(const uint8_t* tok, int len, int64_t* out) { return text_atoi(tok, len, out) ? 1 : 0; }
int tn_atof(const uint8_t* tok, int len, double* out) { return text_atof(tok, len, out); }
// register forms (len <= 16): 1 ok / 0 gave up; 0 value / 1 gave up
int tn_fast_atoi(const uint8_t* tok, int len, int64_t* out) { uint32_t w[4]; window(tok, len, w); return fast_atoi16(w, len, out) ? 1 : 0; }
int tn_fast_atof(const uint8_t* tok, int len, double* out) { uint32_t w[4]; window(tok, len, w); double p10[23]; for (int k = 0; k < 23; ++k) p10[k] = text_pow10(k); return fast_atof16(w, len, p10, out); }
}
