// g++ host check of the 32-lane cooperative 512-point FFT of csrc/fft16.h (structures A and B,
// both directions), lane by lane with an explicit LDS array, against a naive double DFT.
#include <cmath>
#include <cstdio>
#include <vector>
#include <complex>
#include "../../pnp_admm_cnc_mri_amd/csrc/fft16.h"
using namespace pnp;
static c32 TW[512];

// structure A: x in t-layout order (natural index n) -> X natural index k
static void coop_a(std::vector<c32>& v, bool inv) {
    std::vector<c32> lds(16 * 34);
    c32 regs[32][16];
    for (int t = 0; t < 32; ++t) {
        c32 a[16];
        for (int j = 0; j < 16; ++j) a[j] = v[t + 32 * j];
        c32 twt[16]; for (int k = 0; k < 16; ++k) twt[k] = TW[(t * k) & 511];      // the lane's row of the W512^(t k) table
        if (inv) fft512_a1<true>(a, twt); else fft512_a1<false>(a, twt);
        for (int k2 = 0; k2 < 16; ++k2) lds[k2 * 34 + t] = a[k2];
    }
    for (int L = 0; L < 32; ++L) {
        const int k2 = L >> 1, h = L & 1;
        c32 a[16];
        for (int i = 0; i < 16; ++i) a[i] = lds[k2 * 34 + 2 * i + h];
        if (inv) fft512_a2<true>(a, h); else fft512_a2<false>(a, h);
        for (int q = 0; q < 16; ++q) regs[L][q] = a[q];
    }
    for (int L = 0; L < 32; ++L) {
        const int k2 = L >> 1, h = L & 1;
        for (int q = 0; q < 16; ++q) v[k2 + 16 * q + 256 * h] = bfly2(regs[L][q], regs[L ^ 1][q], h);
    }
}
// structure B: X in k-layout (natural index k) -> x natural index n
static void coop_b(std::vector<c32>& v, bool inv) {
    std::vector<c32> lds(16 * 34);
    c32 regs[32][16];
    for (int L = 0; L < 32; ++L) {
        const int k2 = L >> 1, h = L & 1;
        for (int q = 0; q < 16; ++q) regs[L][q] = v[k2 + 16 * q + 256 * h];
    }
    for (int L = 0; L < 32; ++L) {
        const int k2 = L >> 1, h = L & 1;
        c32 a[16];
        for (int q = 0; q < 16; ++q) a[q] = bfly2(regs[L][q], regs[L ^ 1][q], h);
        if (inv) fft512_b1<true>(a, h); else fft512_b1<false>(a, h);
        for (int i = 0; i < 16; ++i) lds[k2 * 34 + 2 * i + h] = a[i];
    }
    for (int t = 0; t < 32; ++t) {
        c32 a[16];
        for (int k2 = 0; k2 < 16; ++k2) a[k2] = lds[k2 * 34 + t];
        c32 twt[16]; for (int k = 0; k < 16; ++k) twt[k] = TW[(t * k) & 511];
        if (inv) fft512_b2<true>(a, twt); else fft512_b2<false>(a, twt);
        for (int j = 0; j < 16; ++j) v[t + 32 * j] = a[j];
    }
}

int main() {
    for (int m = 0; m < 512; ++m) { double a = -2.0 * M_PI * m / 512.0; TW[m] = mk((float)cos(a), (float)sin(a)); }
    std::vector<c32> x(512);
    unsigned s = 12345;
    for (auto& e : x) { s = s * 1664525u + 1013904223u; float re = (s >> 8) / 16777216.0f - 0.5f; s = s * 1664525u + 1013904223u; e = mk(re, (s >> 8) / 16777216.0f - 0.5f); }
    double worst = 0;
    for (int structure = 0; structure < 2; ++structure)
        for (int inv = 0; inv < 2; ++inv) {
            std::vector<c32> v = x;
            if (structure == 0) coop_a(v, inv); else coop_b(v, inv);
            double num = 0, den = 0;
            for (int k = 0; k < 512; ++k) {
                std::complex<double> acc = 0;
                for (int n = 0; n < 512; ++n) acc += std::complex<double>(x[n].x, x[n].y) * std::polar(1.0, (inv ? 2.0 : -2.0) * M_PI * n * k / 512.0);
                num += std::norm(acc - std::complex<double>(v[k].x, v[k].y));
                den += std::norm(acc);
            }
            const double err = std::sqrt(num / den);
            printf("structure %c inv %d rel err %.3e\n", structure ? 'B' : 'A', inv, err);
            if (err > worst) worst = err;
        }
    return worst < 1e-6 ? 0 : 1;
}
