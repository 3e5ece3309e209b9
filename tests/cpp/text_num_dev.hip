// Device check (run by tests/test_text_gpu.py::test_number_conversions_on_the_device) of minnow_amd/csrc/mnw_text_num.h: every token of argv[1] (one per line) through text_atof / fast_atof16 /
// text_atoi / fast_atoi16 on the GPU, from global memory and from LDS, printed beside the host's results of the same source.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tests/cpp/text_num_dev.hip -o tests/cpp/text_num_dev
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../minnow_amd/csrc/mnw_text_num.h"

struct Res { int rc_f, rc_ff, rc_i, rc_fi, rc_f_lds; double f, ff, f_lds; long long i, fi; };

__global__ void k(const uint8_t* text, const int* off, const int* len, int n, Res* out) {
    __shared__ uint8_t lds[4096];
    __shared__ double p10[23];
    for (int q = threadIdx.x; q < 4096; q += blockDim.x) lds[q] = ' ';
    if (threadIdx.x < 23) p10[threadIdx.x] = text_pow10((int)threadIdx.x);
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint8_t* tok = text + off[t];
    const int l = len[t];
    Res r = {};
    r.rc_f = text_atof(tok, l, &r.f);
    int64_t x = 0;
    r.rc_i = text_atoi(tok, l, &x) ? 1 : 0; r.i = x;
    uint8_t* mine = lds + 32 * threadIdx.x + (threadIdx.x % 3);            // unaligned on purpose
    if (l <= 24 && threadIdx.x < 120) { for (int q = 0; q < l; ++q) mine[q] = tok[q]; r.rc_f_lds = text_atof(mine, l, &r.f_lds); } else r.rc_f_lds = -1;
    if (l <= 16) {
        uint8_t b[16];
        for (int q = 0; q < 16; ++q) b[q] = q < l ? tok[q] : (uint8_t)' ';
        uint32_t w[4];
        for (int q = 0; q < 4; ++q) w[q] = b[4 * q] | (b[4 * q + 1] << 8) | (b[4 * q + 2] << 16) | ((uint32_t)b[4 * q + 3] << 24);
        r.rc_ff = fast_atof16(w, l, p10, &r.ff);
        int64_t y = 0;
        r.rc_fi = fast_atoi16(w, l, &y) ? 1 : 0; r.fi = y;
    } else { r.rc_ff = -1; r.rc_fi = -1; }
    out[t] = r;
}

int main(int argc, char** argv) {
    std::ifstream f(argv[1]);
    std::string line, text;
    std::vector<int> off, len;
    while (std::getline(f, line)) { off.push_back((int)text.size()); len.push_back((int)line.size()); text += line; text += ' '; }
    const int n = (int)off.size();
    uint8_t* dt; int *doff, *dlen; Res* dres;
    hipMalloc(&dt, text.size() + 64); hipMalloc(&doff, n * 4); hipMalloc(&dlen, n * 4); hipMalloc(&dres, n * sizeof(Res));
    hipMemcpy(dt, text.data(), text.size(), hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dlen, len.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 127) / 128), dim3(128), 0, 0, dt, doff, dlen, n, dres);
    std::vector<Res> res(n);
    hipMemcpy(res.data(), dres, n * sizeof(Res), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < n; ++t) {
        const uint8_t* tok = (const uint8_t*)text.data() + off[t];
        double hf = 0; int64_t hi = 0;
        const int hrc = text_atof(tok, len[t], &hf);
        const int hirc = text_atoi(tok, len[t], &hi) ? 1 : 0;
        const Res& r = res[t];
        const bool ok = r.rc_f == hrc && (hrc != 0 || memcmp(&r.f, &hf, 8) == 0) && r.rc_i == hirc && (!hirc || r.i == hi) &&
                        (r.rc_f_lds == -1 || (r.rc_f_lds == hrc && (hrc != 0 || memcmp(&r.f_lds, &hf, 8) == 0))) &&
                        (r.rc_ff != 0 || (hrc == 0 && memcmp(&r.ff, &hf, 8) == 0)) && (r.rc_fi != 1 || (hirc == 1 && r.fi == hi));
        if (!ok) { ++bad; if (bad < 20) printf("MISMATCH '%.*s': dev atof %d %.17g lds %d %.17g fast %d %.17g | host %d %.17g | atoi dev %d %lld fast %d %lld host %d %lld\n",
                                            len[t], tok, r.rc_f, r.f, r.rc_f_lds, r.f_lds, r.rc_ff, r.ff, hrc, hf, r.rc_i, r.i, r.rc_fi, r.fi, hirc, (long long)hi); }
    }
    printf("%d tokens, %d mismatches\n", n, bad);
    return bad != 0;
}
